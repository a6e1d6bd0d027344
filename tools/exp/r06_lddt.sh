#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ulimit -c 0
RSK_LIB=$PWD/build/var_nofence/librsk.so timeout 300 python -m pytest tests/test_gpu_align.py -x -q 2>&1 | tail -2 | tee gpurun_out/r06_lddt_t.txt
grep -q passed gpurun_out/r06_lddt_t.txt || exit 1
grep -q "failed\|error" gpurun_out/r06_lddt_t.txt && exit 1
for rep in 1 2 3; do for v in head nofence; do
  if [ $v = head ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_$v/librsk.so; fi
  timeout 200 python bench.py --live-only 2>/dev/null | python -c "
import json,sys
for e in json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline_live']:
    if e['kernel'] in ('k_lddt',): print('$v', e['kernel'], round(e['kernel_ms'],4), 'frac', round(e.get('frac',0),3))
"
done; done > gpurun_out/r06_lddt_ab.txt 2>&1
cat gpurun_out/r06_lddt_ab.txt

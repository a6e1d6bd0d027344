#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ulimit -c 0
timeout 300 python -m pytest tests/test_gpu_align.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06_tb_t1.txt
grep -q "failed\|error" gpurun_out/r06_tb_t1.txt && exit 1
grep -q "passed" gpurun_out/r06_tb_t1.txt || exit 1
echo "== prefetch on (0): alignment tests"
RSK_TB_PREFETCH_AFTER=0 timeout 300 python -m pytest tests/test_gpu_align.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06_tb_t2.txt
grep -q "passed" gpurun_out/r06_tb_t2.txt || exit 1
grep -q "failed\|error" gpurun_out/r06_tb_t2.txt && exit 1
RSK_TB_PREFETCH_AFTER=64 timeout 600 python -m pytest tests/test_gpu_db_goldens.py tests/test_gpu_search.py -x -q 2>&1 | tail -3
for rep in 1 2; do for v in -1 192 64 0; do
  RSK_TB_PREFETCH_AFTER=$v timeout 150 python bench.py --live-only 2>/dev/null | python -c "
import json,sys
for e in json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline_live']:
    if e['kernel'] in ('k_traceback',): print('after=$v', e['kernel'], round(e['kernel_ms'],4), 'ns/step', round(e['ns_per_step_of_the_longest_walk'],1))
"
done; done > gpurun_out/r06_tb_ab.txt 2>&1
cat gpurun_out/r06_tb_ab.txt
for rep in 1 2; do for v in -1 192 32; do
  RSK_TB_PREFETCH_AFTER=$v timeout 150 python bench.py --configs-only config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']
for k,x in d.items():
    if isinstance(x,dict): print('after=$v', k.split('_')[0], '%.3f s' % x['seconds'], 'swqp GHz', x.get('swqp_clock_ghz'))
"
done; done > gpurun_out/r06_tb_c4.txt 2>&1
cat gpurun_out/r06_tb_c4.txt

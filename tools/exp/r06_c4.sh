#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_db_goldens.py tests/test_gpu_multidev.py -x -q 2>&1 | tail -2
for rep in 1 2; do for v in head prev; do
  if [ $v = head ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_prev/librsk.so; fi
  timeout 300 python bench.py --configs-only config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']
print('$v', ' '.join('%s %.3f' % (k.split('_')[0], x['seconds']) for k,x in d.items() if isinstance(x,dict)), [round(x['swqp_clock_ghz'],3) for k,x in d.items() if isinstance(x,dict) and x.get('swqp_clock_ghz')])
"
done; done | tee gpurun_out/r06_c4_dense_ab.txt

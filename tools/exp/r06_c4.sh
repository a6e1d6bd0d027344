#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ulimit -c 0
for rep in 1 2 3; do for v in 10 0; do
  RSK_LOADER_NICE=$v timeout 300 python bench.py --configs-only config3,config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']
print('nice=$v', ' '.join('%s %.3f' % (k.split('_')[0], x['seconds']) for k,x in d.items() if isinstance(x,dict)), [round(x['swqp_clock_ghz'],3) for k,x in d.items() if isinstance(x,dict) and x.get('swqp_clock_ghz')], [round(x['loader_seconds'],2) for k,x in d.items() if isinstance(x,dict)])
"
done; done | tee gpurun_out/r06_c4_nice_ab.txt

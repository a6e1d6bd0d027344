#!/bin/bash
# config 3 / 4 shares: the library of the tree vs a variant library (RSK_LIB), alternating runs
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=$PWD/build/var_${1:-oldqp}/librsk.so
for rep in 1 2; do
  for lib in default variant; do
    if [ $lib = variant ]; then export RSK_LIB=$V; else unset RSK_LIB; fi
    python bench.py --configs-only config3,config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']
print('$lib', ' '.join('%s %.3f' % (k.split('_')[0], v['seconds']) for k,v in d.items() if isinstance(v,dict)))
"
  done
done

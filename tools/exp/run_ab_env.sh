#!/bin/bash
# config 3 / 4 shares under environment variants, alternating runs: run_ab_env.sh "A=1" "B=0 C=2" ... ("-" = no variable)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for v in "$@"; do
    ( if [ "$v" != "-" ]; then export $v; fi
      python bench.py --configs-only ${RSK_AB_CONFIGS:-config4} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']
print('$v', ' '.join('%s %.3f' % (k.split('_')[0], x['seconds']) for k,x in d.items() if isinstance(x,dict)))
" )
  done
done

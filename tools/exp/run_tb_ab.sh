#!/bin/bash
# traceback prefetch A/B: stage times of bench.py's structure-based live entries + config-4 share
cd ${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_align.py -x -q -m gpu 2>&1 | grep "passed\|failed"
for v in - RSK_LIB=$PWD/build/var_nopfb/librsk.so - RSK_LIB=$PWD/build/var_nopfb/librsk.so; do
( if [ "$v" != "-" ]; then export $v; fi
  timeout 600 python bench.py --live-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', ' | '.join('%s %.4f ms' % (k['kernel'], k['kernel_ms']) for k in d['roofline_live'] if k['kernel'] in ('k_traceback','k_lddt','k_sw_qp')))
" )
done
RSK_AB_CONFIGS=config4 bash tools/exp/run_ab_env.sh - RSK_LIB=$PWD/build/var_nopfb/librsk.so

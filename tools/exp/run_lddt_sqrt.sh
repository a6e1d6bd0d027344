#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_configs.py tests/test_gpu_db_goldens.py -x -q -m gpu 2>&1 | grep "passed\|failed"
for i in 1 2; do timeout 600 python bench.py --live-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' | '.join('%s %.4f ms' % (k['kernel'], k['kernel_ms']) for k in d['roofline_live'] if k['kernel'] in ('k_traceback','k_lddt','k_sw_qp')))
"; done

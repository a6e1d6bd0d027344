#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_mkf.py -x -q -m gpu 2>&1 | grep "passed\|failed" 
for i in 1 2; do timeout 300 python tools/bench_search.py 0 sensitive 2>/dev/null | grep seconds; done
timeout 600 python tools/exp/aln_len_hist.py 2>&1 | grep -v amdgpu | tail -3

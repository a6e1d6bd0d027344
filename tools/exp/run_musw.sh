#!/bin/bash
# Mu filter iteration loop: parity tests, then the forward pass of the triangle (bench.py --live-only) with the r04 geometry and the new ones
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_mu_sw.py tests/test_gpu_search.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/musw_tests.txt
cat gpurun_out/musw_tests.txt
: > gpurun_out/musw_ab.txt
for round in 1 2; do
  for v in RSK_MUSW2_FIXED_R RSK_MUSW2_POW2 RSK_NONE; do
    unset RSK_MUSW2_FIXED_R RSK_MUSW2_POW2; export $v=1
    echo "== $v" >> gpurun_out/musw_ab.txt
    timeout 600 python tools/exp/live_ms.py 2>&1 | grep "k_mu_sw" >> gpurun_out/musw_ab.txt
  done
done
cat gpurun_out/musw_ab.txt

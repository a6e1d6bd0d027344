#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export RSK_REQUIRE_REF=1
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/all_gpu_tests.txt
cat gpurun_out/all_gpu_tests.txt

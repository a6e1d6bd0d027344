#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_ref_shaped.py tests/test_gpu_multidev.py -x -q -m gpu 2>&1 | grep "passed\|failed"
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -2 gpurun_out/r05_bench.err

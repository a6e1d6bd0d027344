#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RSK_REQUIRE_REF=1
timeout 2400 python -m pytest tests/test_gpu_search.py tests/test_gpu_dist.py tests/test_gpu_rccl.py tests/test_gpu_prefilter.py -x -q > gpurun_out/r06g_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06g_tests.txt; tail -25 gpurun_out/r06g_tests.txt

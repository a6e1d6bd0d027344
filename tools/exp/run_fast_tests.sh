#!/bin/bash
# the tests around the -fast -db path + the config-2 trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_mu_sw.py tests/test_gpu_search.py tests/test_gpu_fast_shards.py tests/test_gpu_ref_shaped.py tests/test_gpu_vs_reference_binary.py -x -q 2>&1 | tail -15 > gpurun_out/fast_tests.txt
cat gpurun_out/fast_tests.txt
bash tools/exp/run_cfg_trace.sh config2 | cut -c1-900
grep "^\[MuPreFilter\]\|^\[PostMuFilter\]\|^\[rsk_rsb\|^\[prefilter" gpurun_out/trace_config2.err | head -16

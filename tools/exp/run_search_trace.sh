#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_search.py tests/test_gpu_configs.py tests/test_gpu_full_golden.py tests/test_gpu_multidev.py tests/test_gpu_vs_reference_binary.py tests/test_gpu_ref_shaped.py -x -q 2>&1 | tail -8 > gpurun_out/search_tests.txt
cat gpurun_out/search_tests.txt
RSK_TRACE=1 timeout 600 python tools/bench_search.py 0 sensitive > gpurun_out/trace_self.json 2> gpurun_out/trace_self.err
grep seconds gpurun_out/trace_self.json
RSK_TRACE=1 timeout 600 python tools/bench_search.py 0 sensitive bca > gpurun_out/trace_bca.json 2> gpurun_out/trace_bca.err
grep seconds gpurun_out/trace_bca.json

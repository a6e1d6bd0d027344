#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${T}_gpu_tests.txt
cat gpurun_out/${T}_gpu_tests.txt
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -2 gpurun_out/${T}_bench.err
timeout 3000 python tools/bench_configs_full.py > gpurun_out/${T}_configs_full.json 2> gpurun_out/${T}_configs_full.err
tail -3 gpurun_out/${T}_configs_full.err
RSK_TRACE=1 python tools/bench_search.py qdb 256 125000 sensitive > gpurun_out/${T}_trace_c3.json 2> gpurun_out/${T}_trace_c3.err
grep seconds gpurun_out/${T}_trace_c3.json

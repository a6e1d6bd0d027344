#!/bin/bash
# round-end measurement set on the GPU box: full GPU test suite, profiles (bench / live / searches), the default bench line,
# the full-size configs, the trace of a config-3-shaped search (loader vs search per batch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r05}
mkdir -p gpurun_out
export RSK_REQUIRE_REF=1
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > gpurun_out/${T}_gpu_tests.txt
cat gpurun_out/${T}_gpu_tests.txt
bash tools/exp/round_profiles.sh $T > gpurun_out/${T}_round_profiles.log 2>&1
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -2 gpurun_out/${T}_bench.err
timeout 3000 python tools/bench_configs_full.py > gpurun_out/${T}_configs_full.json 2> gpurun_out/${T}_configs_full.err
tail -2 gpurun_out/${T}_configs_full.err
RSK_TRACE=1 python tools/bench_search.py qdb 256 125000 sensitive > gpurun_out/${T}_trace_c3.json 2> gpurun_out/${T}_trace_c3.err
grep seconds gpurun_out/${T}_trace_c3.json
ls gpurun_out | head -50

#!/bin/bash
# round-end set on the final kernels: GPU tests, live-kernel PMC (-> profiles/r04_live_pmc.json before the bench line is
# taken), search traces, bench command profile, the default bench line, configs 3 / 4 at their stated size
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04z}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${T}_gpu_tests.txt
cat gpurun_out/${T}_gpu_tests.txt
bash tools/prof_live.sh ${T}_live > gpurun_out/${T}_prof_live.log 2>&1
cp gpurun_out/prof_${T}_live/live_pmc.json profiles/r04_live_pmc.json
bash tools/prof_bench.sh ${T}_bench > gpurun_out/${T}_prof_bench.log 2>&1
cp gpurun_out/prof_${T}_bench/traffic.json profiles/r04_traffic.json
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -2 gpurun_out/${T}_bench.err
bash tools/prof_search.sh ${T}_search_self 0 sensitive > /dev/null 2>&1
bash tools/prof_search.sh ${T}_search_c2 0 fast bca db > /dev/null 2>&1
bash tools/prof_search.sh ${T}_search_c3 qdb 256 125000 sensitive > /dev/null 2>&1
bash tools/prof_search.sh ${T}_search_c4 qdb 1000 87500 verysensitive > /dev/null 2>&1
timeout 3000 python tools/bench_configs_full.py > gpurun_out/${T}_configs_full.json 2> gpurun_out/${T}_configs_full.err
tail -3 gpurun_out/${T}_configs_full.err
tail -c 400 gpurun_out/${T}_bench.json

#!/bin/bash
# the round's profile set on the GPU box: bench command (trace + PMC + traffic), live kernels (trace + PMC), kernel traces
# of the self search and of the three config-shaped searches -> gpurun_out/prof_<tag>/
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
bash tools/prof_bench.sh ${T}_bench > gpurun_out/${T}_prof_bench.log 2>&1
bash tools/prof_live.sh ${T}_live > gpurun_out/${T}_prof_live.log 2>&1
bash tools/prof_search.sh ${T}_search_self 0 sensitive > /dev/null 2>&1
bash tools/prof_search.sh ${T}_search_c2 0 fast bca db > /dev/null 2>&1
bash tools/prof_search.sh ${T}_search_c3 qdb 256 125000 sensitive > /dev/null 2>&1
bash tools/prof_search.sh ${T}_search_c4 qdb 1000 87500 verysensitive > /dev/null 2>&1
for d in gpurun_out/prof_${T}_*; do echo "== $d"; head -14 $d/summary.txt; done
head -c 600 gpurun_out/prof_${T}_live/live_pmc.json

#!/bin/bash
# after the prefilter seed-walk change: prefilter / full-size golden tests, live PMC on the final sources, the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ulimit -c 0
export RSK_REQUIRE_REF=1
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep "passed\|failed\|error" > gpurun_out/r05_final_tests.txt
cat gpurun_out/r05_final_tests.txt
unset RSK_REQUIRE_REF
bash tools/prof_live.sh r05_live > gpurun_out/r05_prof_live.log 2>&1
cp gpurun_out/prof_r05_live/live_pmc.json profiles/r05_live_pmc.json
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -2 gpurun_out/r05_bench.err

#!/bin/bash
# after the side-stream change of the gapless launch: full GPU tests, bench kernel trace + traffic counters, the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ulimit -c 0
export RSK_REQUIRE_REF=1
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep "passed\|failed\|error" > gpurun_out/r05_final_tests.txt
cat gpurun_out/r05_final_tests.txt
unset RSK_REQUIRE_REF
bash tools/prof_bench.sh r05_bench > gpurun_out/r05_prof_bench.log 2>&1
cp gpurun_out/prof_r05_bench/traffic.json profiles/r05_traffic.json
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -2 gpurun_out/r05_bench.err
grep "k_gapless" gpurun_out/prof_r05_bench/summary.txt | grep "SQ_WAVES \|GRBM" | cut -c1-200

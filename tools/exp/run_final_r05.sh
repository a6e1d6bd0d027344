#!/bin/bash
# final pass of round 5 on the GPU box: full GPU test suite, TSan over the host code, live PMC on the final kernels, the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ulimit -c 0
export RSK_REQUIRE_REF=1
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep "passed\|failed\|error" > gpurun_out/r05_final_tests.txt
cat gpurun_out/r05_final_tests.txt
unset RSK_REQUIRE_REF
timeout 900 bash tools/tsan_host.sh > gpurun_out/r05_tsan.log 2>&1
grep -c "WARNING: ThreadSanitizer" gpurun_out/tsan/report.txt; grep "^exit" gpurun_out/tsan/report.txt | sort | uniq -c
bash tools/prof_live.sh r05_live > gpurun_out/r05_prof_live.log 2>&1
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -2 gpurun_out/r05_bench.err

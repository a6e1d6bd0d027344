#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${T}_gpu_tests.txt
cat gpurun_out/${T}_gpu_tests.txt
bash tools/tsan_host.sh > gpurun_out/${T}_tsan.log 2>&1; tail -1 gpurun_out/${T}_tsan.log
bash tools/prof_live.sh ${T}_live > gpurun_out/${T}_prof_live.log 2>&1
bash tools/prof_search.sh ${T}_search_c2 0 fast bca db > /dev/null 2>&1
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -2 gpurun_out/${T}_bench.err

#!/bin/bash
# last pass: GPU tests, live-kernel PMC (the kernel source's sha256 goes into the record), the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04y}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${T}_gpu_tests.txt
cat gpurun_out/${T}_gpu_tests.txt
bash tools/prof_live.sh ${T}_live > gpurun_out/${T}_prof_live.log 2>&1
cp gpurun_out/prof_${T}_live/live_pmc.json profiles/r04_live_pmc.json
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -2 gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.json

#!/bin/bash
# final measurements after the k_sw_qp rewrite: bench command profile (trace + PMC + traffic), the default bench line,
# BASELINE configs 3 / 4 at their stated size (8 == 3 == 1 shards, reference samples)
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04f}
mkdir -p gpurun_out
bash tools/prof_bench.sh ${T}_bench > gpurun_out/${T}_prof_bench.log 2>&1
tail -12 gpurun_out/prof_${T}_bench/traffic.json
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -2 gpurun_out/${T}_bench.err
timeout 3000 python tools/bench_configs_full.py > gpurun_out/${T}_configs_full.json 2> gpurun_out/${T}_configs_full.err
tail -3 gpurun_out/${T}_configs_full.err
head -c 1500 gpurun_out/${T}_configs_full.json

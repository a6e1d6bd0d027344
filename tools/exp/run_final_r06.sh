#!/bin/bash
# round-6 final measurement set on the GPU box: every GPU test, the profile set (bench trace + PMC + traffic, live kernels trace +
# PMC, traces of the self search and the three config-shaped searches), the default bench line as the driver runs it, the
# full-size configs, bench.py --gpus 2 on one device (plumbing)
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=r06
mkdir -p gpurun_out
ulimit -c 0
export RSK_REQUIRE_REF=1
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep "passed\|failed\|error" > gpurun_out/${T}_gpu_tests.txt
cat gpurun_out/${T}_gpu_tests.txt
unset RSK_REQUIRE_REF
bash tools/exp/round_profiles.sh $T > gpurun_out/${T}_round_profiles.log 2>&1
cp gpurun_out/prof_${T}_bench/traffic.json profiles/${T}_traffic.json 2>/dev/null
cp gpurun_out/prof_${T}_live/live_pmc.json profiles/${T}_live_pmc.json 2>/dev/null
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/${T}_bench_detail.json ) > gpurun_out/${T}_bench.out 2> gpurun_out/${T}_bench.err
tail -n 1 gpurun_out/${T}_bench.out | wc -c; tail -n 1 gpurun_out/${T}_bench.out; tail -4 gpurun_out/${T}_bench.err
RSK_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --detail gpurun_out/${T}_bench_2ranks_detail.json > gpurun_out/${T}_bench_2ranks.out 2> gpurun_out/${T}_bench_2ranks.err
tail -n 1 gpurun_out/${T}_bench_2ranks.out
timeout 3000 python tools/bench_configs_full.py > gpurun_out/${T}_configs_full.json 2> gpurun_out/${T}_configs_full.err
tail -2 gpurun_out/${T}_configs_full.err; head -c 1500 gpurun_out/${T}_configs_full.json
ls gpurun_out | head -80

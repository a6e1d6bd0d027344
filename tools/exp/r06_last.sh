#!/bin/bash
# last check of the round on the final tree: every GPU test, smoke(), the bench line as the driver runs it
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; ulimit -c 0
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep "passed\|failed\|error" | tee gpurun_out/r06_last_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06_bench_detail.json ) > gpurun_out/r06_bench.out 2> gpurun_out/r06_bench.err
tail -n 1 gpurun_out/r06_bench.out | wc -c; tail -n 1 gpurun_out/r06_bench.out; tail -4 gpurun_out/r06_bench.err

#!/bin/bash
# round profiles on the GPU box: kernel trace + PMC of the bench command and of the live kernels -> gpurun_out/prof_r05*/
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/prof_bench.sh r05 > gpurun_out/prof_r05.log 2>&1
bash tools/prof_live.sh r05_live > gpurun_out/prof_r05_live.log 2>&1
tail -30 gpurun_out/prof_r05_live.log | cut -c1-300

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_configs.py tests/test_gpu_search.py tests/test_gpu_mkf.py -x -q -m gpu 2>&1 | tail -4
RSK_AB_CONFIGS=config3,config4 bash tools/exp/run_ab_env.sh - RSK_LIB=$PWD/build/var_base/librsk.so

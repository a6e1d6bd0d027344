#!/bin/bash
# k_sw_qp as built + the timing-only variants: alignment parity tests, PMC (clocks), un-profiled A/B, config-4 share A/B against base
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
bash tools/exp/swq_pmc.sh main nobest nostore nobeststore > gpurun_out/swq_whatif.txt 2>&1
bash tools/exp/run_swq_ab.sh main nobest nostore nobeststore
RSK_AB_CONFIGS=config4 bash tools/exp/run_ab_env.sh - RSK_LIB=$PWD/build/var_base/librsk.so

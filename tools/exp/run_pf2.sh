#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefilter.py -x -q 2>&1 | tail -5 > gpurun_out/pf2.txt
export RSK_TRACE=1
for d in 1 0; do echo "=== syn RSK_PF_DEBUG=$d"; RSK_PF_DEBUG=$d timeout 600 python tools/exp/pf_bench.py syn 0 2 2>&1 | grep -v "^\[pool\]\|amdgpu.ids" | tail -5; done >> gpurun_out/pf2.txt 2>&1
for d in 1 0; do echo "=== scop40 RSK_PF_DEBUG=$d"; RSK_PF_DEBUG=$d timeout 600 python tools/exp/pf_bench.py scop40 0 2 2>&1 | grep -v "^\[pool\]\|amdgpu.ids" | tail -5; done >> gpurun_out/pf2.txt 2>&1
cat gpurun_out/pf2.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/swq_prof.txt
for c in 2 4; do
echo "== chunk $c" >> gpurun_out/swq_prof.txt
RSK_SWQ_CHUNK=$c RSK_LIB=$PWD/build/var_prof/librsk.so timeout 600 python tools/exp/swq_bench.py 2 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/swq_prof.txt
RSK_SWQ_CHUNK=$c timeout 600 python tools/exp/swq_bench.py 3 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/swq_prof.txt
RSK_SWQ_CHUNK=$c RSK_LIB=$PWD/build/var_r16/librsk.so timeout 600 python tools/exp/swq_bench.py 3 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/swq_prof.txt
done
cat gpurun_out/swq_prof.txt

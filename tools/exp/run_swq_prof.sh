#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/swq_prof.txt
for c in 0 1 2; do
echo "== chunk $c (groups set; 0 = by work)" >> gpurun_out/swq_prof.txt
if [ $c = 0 ]; then unset RSK_SWQ_CHUNK; else export RSK_SWQ_CHUNK=$c; fi
RSK_LIB=$PWD/build/var_prof/librsk.so timeout 600 python tools/bench_align_groups.py 64 11211 2>&1 | grep "swq_prof\|sw_kernel_ms" | tail -2 >> gpurun_out/swq_prof.txt
timeout 600 python tools/exp/swq_bench.py 3 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/swq_prof.txt
done
echo "== base" >> gpurun_out/swq_prof.txt
RSK_LIB=$PWD/build/var_base/librsk.so timeout 600 python tools/bench_align_groups.py 64 11211 2>&1 | grep "sw_kernel_ms" | tail -1 >> gpurun_out/swq_prof.txt
RSK_LIB=$PWD/build/var_base/librsk.so timeout 600 python tools/exp/swq_bench.py 3 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/swq_prof.txt
cat gpurun_out/swq_prof.txt

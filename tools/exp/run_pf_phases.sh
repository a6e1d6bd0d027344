cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RSK_TRACE=1
for d in 1 2 0; do echo "=== RSK_PF_DEBUG=$d"; RSK_PF_DEBUG=$d timeout 600 python tools/exp/pf_bench.py syn 0 2 2>&1 | grep -v "^\[" | tail -5; done > gpurun_out/pf_phases.txt 2>&1
RSK_PF_DEBUG=0 timeout 600 python tools/exp/pf_bench.py syn 0 1 2>&1 | grep "prefilter\]" | tail -3 >> gpurun_out/pf_phases.txt
echo "=== scop40" >> gpurun_out/pf_phases.txt
timeout 600 python tools/exp/pf_bench.py scop40 0 2 2>&1 | tail -6 >> gpurun_out/pf_phases.txt
cat gpurun_out/pf_phases.txt

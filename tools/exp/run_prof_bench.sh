cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/prof_bench.sh r05 > gpurun_out/prof_r05.log 2>&1
grep -A3 "KERNEL_DISPATCH\|k_gapless_ring" gpurun_out/prof_r05/summary.txt | head -20; cat gpurun_out/prof_r05/traffic.json

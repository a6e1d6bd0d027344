cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gapless.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/gl_tests.txt; cat gpurun_out/gl_tests.txt
python tools/exp/shard_times.py window 2>&1 | grep -v amdgpu.ids > gpurun_out/shard_times.txt; python - <<EE
import json
d=json.load(open("gpurun_out/shard_times.txt"))
print(d["one_gpu_kernel_ms"])
for s in ("window",):
    for k,v in d[s].items():
        print(s,k,v["rank_ms"],v["max_over_mean_ms"],v["efficiency"],v["launch_Tcells_per_s"])
EE

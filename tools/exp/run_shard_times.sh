#!/bin/bash
# gapless parity tests, then the one-GPU launch and the per-rank kernel times of the window scheme (N = 2 / 4 / 8), with the
# per-pair kernel on the side stream (as built) and behind the ring kernels (RSK_GAPLESS_NO_SIDE_STREAM=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gapless.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | grep "passed\|failed"
for v in RSK_GAPLESS_NO_SIDE_STREAM=1 RSK_NONE=1 RSK_GAPLESS_NO_SIDE_STREAM=1 RSK_NONE=1; do
( export $v; timeout 600 python tools/exp/shard_times.py window 2>&1 | grep -v amdgpu.ids > gpurun_out/shard_times.txt; python - <<EE
import json
d=json.load(open("gpurun_out/shard_times.txt"))
print("$v one GPU %.3f ms" % d["one_gpu_kernel_ms"], " ".join("%s %.4f" % (k, v["efficiency"]) for k,v in d["window"].items()), d["window"]["n8"]["rank_ms"])
EE
)
done

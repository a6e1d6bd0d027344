#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for cfg in "1.0 3" "1.0 8" "1.5 8" "0.5 8"; do
set -- $cfg
RSK_WINDOW_RAGGED=$1 RSK_WINDOW_LONGW=$2 timeout 600 python tools/exp/shard_times.py window 2>&1 | grep -v amdgpu.ids > gpurun_out/shard_times.txt; python - <<EE
import json
d=json.load(open("gpurun_out/shard_times.txt"))
for k,v in d["window"].items():
    print("ragged $1 longw $2", k,v["rank_ms"],v["max_over_mean_ms"],v["efficiency"])
EE
done

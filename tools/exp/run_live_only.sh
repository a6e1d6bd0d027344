#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python bench.py --live-only > gpurun_out/live_only.json 2> gpurun_out/live_only.err
tail -3 gpurun_out/live_only.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/live_only.json").read().strip().splitlines()[-1])
for k in d["roofline_live"]:
    print({x:(round(v,4) if isinstance(v,float) else v) for x,v in k.items() if x not in ("pmc","note","what","hbm","lds","pmc_dispatch")})
PY

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/bench_full.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print("value %.3e  ms/step %.2f  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("cpu_baseline", d.get("cpu_baseline",{}).get("value"))
for k in d.get("roofline_live",[]): print(k["kernel"], round(k["kernel_ms"],2), "ms", round(k["cells_per_s"]/1e12,3) if "cells_per_s" in k else "", "frac", round(k.get("frac",0),3), k.get("sustained_clock_ghz"))
ps=d.get("predicted_scaling",{}).get("window",{})
for n,v in ps.items(): print(n, v["rank_ms"], v["efficiency"])
print("search", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.get("search",{}).items() if k in ("seconds","pairs_per_s","hits")})
sb=d.get("search_bca",{}); print("search_bca", {k:sb[k] for k in sb if k in ("seconds","speedup_vs_reference","identical")})
for k,v in d.get("configs",{}).items():
    if isinstance(v,dict): print(k, v.get("seconds"), v.get("pairs_per_s"))
PY

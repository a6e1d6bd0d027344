#!/usr/bin/env python3
"""rocprofv3 PMC databases of tools/prof_live.sh -> per-kernel fractions for bench.py's roofline_live (profiles/r03_live_pmc.json).
Per kernel, the LONGEST dispatch (the timed workload; warm-up dispatches of the same kernel are equal or shorter):
  valu_issue_frac = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * cycles),  cycles = GRBM_GUI_ACTIVE / 8 XCDs
  lds_busy_frac   = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles), lds_conflict_frac = SQ_LDS_BANK_CONFLICT / (256 * cycles)
  hbm bytes       = FETCH_SIZE (KB, doubled on gfx950 as MI355X_MICROARCH.md prescribes) + WRITE_SIZE (KB)."""
import glob
import json
import os
import sqlite3
import sys

root = sys.argv[1]
KERN = {"k_mu_sw": "k_mu_sw", "k_sw_float": "k_sw_float", "k_sw_qp": "k_sw_qp", "k_prefilter": "k_prefilter"}
SRC = {"k_mu_sw": "reseek_amd/csrc/k_mu_sw.hip", "k_sw_float": "reseek_amd/csrc/k_sw_float.hip", "k_sw_qp": "reseek_amd/csrc/k_sw_float.hip",
       "k_prefilter": "reseek_amd/csrc/k_prefilter.hip"}
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
for f in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    try:
        rows = c.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection").fetchall()
    except sqlite3.Error:
        try:
            rows = [(r[0], r[1], r[2], r[3], i) for i, r in enumerate(
                c.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall())]
        except sqlite3.Error:
            continue
    per = {}
    for name, cn, v, dur, did in rows:
        key = next((k for k, pref in KERN.items() if pref in name and "candidates" not in name), None)
        if key is None:
            continue
        per.setdefault((key, did), {"dur": dur, "name": name})
        per[(key, did)][cn] = per[(key, did)].get(cn, 0.0) + v
    for key in KERN:
        cand = [(d["dur"], d) for (k, _), d in per.items() if k == key]
        if not cand:
            continue
        best = max(cand, key=lambda x: x[0])[1]
        vals.setdefault(key, {}).update({k: v for k, v in best.items() if k not in ("dur", "name")})
        vals[key]["dispatch_name"] = best["name"]
        vals[key]["dispatches_of_kernel"] = len(cand)
        vals[key].setdefault("dispatch_ns", []).append(best["dur"])
out = {}
for key, d in vals.items():
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    e = {"source": "rocprofv3 --pmc passes of `bench.py --live-only` (tools/prof_live.sh)", "kernel_source": SRC[key],
         "dispatch": "the LONGEST dispatch whose kernel name contains '%s' (%s; %d dispatches of it in the run)" % (key, str(d.get("dispatch_name"))[:60], int(d.get("dispatches_of_kernel", 0))),
         "counters": {k: v for k, v in d.items() if k not in ("dispatch_ns", "dispatch_name", "dispatches_of_kernel")},
         "dispatch_ms_profiled": [x / 1e6 for x in d.get("dispatch_ns", [])]}
    if cyc > 0:
        e["cycles"] = cyc
        # the clock the kernel sustained while it was profiled (MI355X_MICROARCH.md DVFS: effective clock = GRBM_GUI_ACTIVE / wall):
        # a kernel at its power limit runs below the 2.4 GHz the nominal issue peak assumes
        ns = d.get("dispatch_ns", [])
        if ns:
            e["sustained_clock_ghz"] = cyc / (sum(ns) / len(ns))
        if "SQ_THREAD_CYCLES_VALU" in d and "SQ_INSTS_VALU" in d and d["SQ_INSTS_VALU"] > 0:
            e["active_lanes_per_valu_inst"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_INSTS_VALU"]
        if "SQ_INSTS_VALU" in d:
            e["valu_issue_frac"] = d["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in d:
            e["lds_busy_frac"] = d["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
        if "SQ_LDS_BANK_CONFLICT" in d:
            e["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / (256.0 * cyc)
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        e["hbm_traffic_bytes"] = 2.0 * 1024.0 * d.get("FETCH_SIZE", 0.0) + 1024.0 * d.get("WRITE_SIZE", 0.0)
    out[key] = e
import hashlib
out["kernel_source_sha256"] = {}
for f in sorted(set(SRC.values())):
    with open(os.path.join(REPO, f), "rb") as fh:
        out["kernel_source_sha256"][f] = hashlib.sha256(fh.read()).hexdigest()
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/prof_bench.sh -> the `roofline.traffic` record bench.py reports
(profiles/<round>_traffic.json).  Per dispatch of the dominant kernel (k_gapless_ring<16,16>): HBM bytes = 2 x FETCH_SIZE KB
(gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM / rocprofv3 section) + WRITE_SIZE KB.  The record
carries the sha256 of the kernel's source file: bench.py reports the traffic only while that file is unchanged, so a stale
number cannot ride along silently."""
import glob
import hashlib
import json
import os
import sqlite3
import sys

root = sys.argv[1]
repo = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_gapless_ring<16, 16>"
SRC = "reseek_amd/csrc/k_mu_gapless.hip"


def per_dispatch(counter):
    vals = []
    for f in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*.db"), recursive=True)):
        c = sqlite3.connect(f)
        try:
            rows = c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
        except sqlite3.Error:
            continue
        per = {}
        for name, cn, v, did in rows:
            if cn == counter and KERNEL in name:
                per[did] = per.get(did, 0.0) + v
        vals += list(per.values())
    return vals


fetch, write = per_dispatch("FETCH_SIZE"), per_dispatch("WRITE_SIZE")
if not fetch or not write:
    sys.exit("no FETCH_SIZE / WRITE_SIZE rows for %s under %s" % (KERNEL, root))
fk, wk = sum(fetch) / len(fetch), sum(write) / len(write)
src_sha = hashlib.sha256(open(os.path.join(repo, SRC), "rb").read()).hexdigest()
print(json.dumps({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of tools/prof_bench.sh (bench.py --steps 5 --warmup 1 "
              "--no-cpu-baseline --no-search --no-live --no-predict)",
    "kernel": KERNEL, "dispatches": {"fetch": len(fetch), "write": len(write)},
    "fetch_size_kb_per_dispatch": fk, "write_size_kb_per_dispatch": wk,
    "correction": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads); WRITE_SIZE as reported",
    "traffic_bytes_per_launch": 2.0 * 1024.0 * fk + 1024.0 * wk,
    "kernel_source": SRC, "kernel_source_sha256": src_sha}, indent=1))

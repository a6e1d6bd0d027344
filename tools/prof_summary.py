#!/usr/bin/env python3
"""Condenses rocprofv3 outputs (sqlite .db: kernel stats + PMC) into a short text summary for profiles/."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    print("==", os.path.relpath(f, root))
    try:
        rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    except sqlite3.Error:
        rows = []
    if rows and "trace" in f:
        print("  kernel stats (us): name, calls, total, average, pct")
        for r in rows[:8]:
            print("   %-60s %5d %14.0f %14.0f %6.2f" % (r[0][:60], r[1], r[2], r[3], r[4]))
    try:
        rows = c.execute("select kernel_name,counter_name,sum(value),count(*),avg(value),avg(duration),max(vgpr_count),"
                         "max(lds_block_size),max(grid_size),max(workgroup_size) from counters_collection "
                         "group by kernel_name,counter_name").fetchall()
    except sqlite3.Error:
        rows = []
    for r in rows:
        if "k_" not in r[0]:
            continue
        print("   %-40s %-22s sum %.6g  n=%d  per-dispatch %.6g  (avg dur %.0f ns, vgpr %s lds %s grid %s wg %s)" % (
            r[0][:40], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]))

#!/usr/bin/env python3
"""Condenses rocprofv3 outputs (kernel stats + PMC csv) into a short text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, root))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12:
            print("  ", ",".join(row))
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        rd = csv.DictReader(open(f))
        for row in rd:
            k = row.get("Kernel_Name", "?")[:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
        print("== pmc:", os.path.relpath(f, root))
        for k, cs in agg.items():
            if "gapless" not in k and "k_" not in k:
                continue
            print("  kernel", k)
            for c, v in sorted(cs.items()):
                n = cnt[(k, c)]
                print("     %-24s total %.6g  per-dispatch %.6g  (n=%d)" % (c, v, v / max(1, n), n))

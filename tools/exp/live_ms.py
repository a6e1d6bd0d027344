#!/usr/bin/env python3
"""kernel_ms of the live-path kernels from `bench.py --live-only` (experiments: RSK_LIB selects a variant library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--live-only"], capture_output=True, text=True, cwd=ROOT)
line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
for k in json.loads(line)["roofline_live"]:
    print("%-12s %8.3f ms  %.3f T cells/s" % (k["kernel"], k["kernel_ms"], k["cells_per_s"] / 1e12))

#!/usr/bin/env python3
"""Robustness at a size no reference run covers: an N-chain synthetic .bca searched against itself with one context and with
two contexts on the device (shards); the sorted hit tables must be equal."""
import hashlib
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import bench_search  # noqa: E402
import reseek_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
lens = bench.scop40_lengths()
rng = np.random.default_rng(77)
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
with tempfile.TemporaryDirectory() as td:
    q = os.path.join(td, "q.bca")
    bench_search.write_bca_fast(q, lens[rng.choice(len(lens), n)], rng, "c")
    digs = []
    for dev in (None, "0,0"):
        out = os.path.join(td, "hits.tsv")
        t0 = time.perf_counter()
        nh, st = ctx.search(q, out, mode, **({"devices": dev} if dev else {}))
        dt = time.perf_counter() - t0
        lines = sorted(open(out, "rb").read().splitlines())
        h = hashlib.md5(b"\n".join(lines)).hexdigest()
        digs.append((len(lines), h))
        print("devices", dev, "seconds %.2f" % dt, "pairs", st[0], "hits", nh, h)
    assert digs[0] == digs[1], digs
    assert digs[0][0] > 0
print("identical")

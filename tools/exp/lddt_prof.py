#!/usr/bin/env python3
"""k_lddt work accounting: 300 queries x 3000 chains -verysensitive; prints sum over hits of C (C - 1) / 2 (C = M columns)."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench, bench_search, reseek_amd
lens = bench.scop40_lengths(); rng = np.random.default_rng(11)
ctx = reseek_amd.Ctx(0)
with tempfile.TemporaryDirectory() as td:
    q, db, out = os.path.join(td, "q.bca"), os.path.join(td, "db.bca"), os.path.join(td, "h.tsv")
    bench_search.write_bca_fast(q, lens[rng.choice(len(lens), 300)], rng, "q")
    bench_search.write_bca_fast(db, lens[rng.choice(len(lens), 3000)], rng, "d")
    n, st = ctx.search(q, out, "verysensitive", db=db, columns="ids")
    ids = np.loadtxt(out, dtype=np.int64)
    print("hits", n, "mean M columns", ids.mean(), "pair tests", int((ids * (ids - 1) // 2).sum()), "waves steps", int(np.ceil(ids * (ids - 1) / 2 / 64).sum()))

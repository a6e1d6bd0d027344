#!/usr/bin/env python3
"""alignment-length distribution of a config-4-shaped search (which k_lddt path / how many columns per pair)"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench, bench_search, reseek_amd
lens = bench.scop40_lengths(); rng = np.random.default_rng(11)
ctx = reseek_amd.Ctx(0)
with tempfile.TemporaryDirectory() as td:
    q, db, out = os.path.join(td, "q.bca"), os.path.join(td, "db.bca"), os.path.join(td, "h.tsv")
    bench_search.write_bca_fast(q, lens[rng.choice(len(lens), 200)], rng, "q")
    bench_search.write_bca_fast(db, lens[rng.choice(len(lens), 3000)], rng, "d")
    ctx.search(q, out, "verysensitive", db=db, columns="query+target+qlo+qhi+ql+tlo+thi+tl+cigar")
    L = []
    for ln in open(out):
        f = ln.split("\t")
        L.append(int(f[3]) - int(f[2]) + 1)
    L = np.array(L)
    print("pairs", len(L), "mean aligned query span", L.mean(), "percentiles 10/50/90/99", np.percentile(L, [10, 50, 90, 99]))
    print("share <= 64:", (L <= 64).mean(), "<= 128:", (L <= 128).mean(), "<= 256:", (L <= 256).mean())

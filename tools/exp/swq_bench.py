#!/usr/bin/env python3
"""k_sw_qp alone on bench.py's roofline_live case (64 queries x 11,211 chains, collect=False): kernel ms of the alignment
kernels of the call (HIP events of the library) and T cells/s.  usage: swq_bench.py [reps] [nq]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import reseek_amd  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seqs = bench.synth_mu_chains(0x5EED5EEC, None)
n = len(seqs)
li = np.array([len(s) for s in seqs], np.uint32)
rng = np.random.default_rng(3)
tot = int(li.sum())
prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, int(L))), rng.integers(0, 16, (7, int(L)))]).astype(np.uint8).reshape(-1) for L in li])
xyz = tuple(np.cumsum(rng.normal(0, 2.2, tot)).astype(np.float32) for _ in range(3))
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
dbs = reseek_amd.Db(ctx, li, mu=np.concatenate(seqs), prof=prof, xyz=xyz, selfrev=np.zeros(n, np.float32))
order = np.random.default_rng(4).permutation(n)[:nq].astype(np.uint32)
qa = np.repeat(order, n)
qb = np.tile(np.arange(n, dtype=np.uint32), nq)
ms = []
for _ in range(reps + 1):
    ctx.align_pairs(dbs, dbs, qa, qb, min_fwd_score=0.0, collect=False)
    ms.append(ctx.last_kernel_ms())
p_, cells, tb = ctx.align_last_work()
m = float(np.median(ms[1:]))
print(json.dumps({"kernel_ms": ms, "median_ms": m, "cells": cells, "Tcells_per_s": cells / m / 1e9, "trace_bytes": tb,
                  "query_len_mean": float(li[order].mean())}))

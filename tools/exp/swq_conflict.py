#!/usr/bin/env python3
"""k_sw_qp alone on queries of a chosen length window against all chains of the SCOP40-shaped set (experiments on the
lane-group geometry: RSK_SWQ_MAXR / RSK_LIB variants).  usage: swq_conflict.py Lmin Lmax [nq] [t]   (t: the shared chain on the B side -> k_sw_qp<true>)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import reseek_amd  # noqa: E402

lmin, lmax = int(sys.argv[1]), int(sys.argv[2])
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 64
seqs = bench.synth_mu_chains(0x5EED5EEC)
n = len(seqs)
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(3)
li = np.array([len(s) for s in seqs], np.uint32)
tot = int(li.sum())
prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, int(L))), rng.integers(0, 16, (7, int(L)))]).astype(np.uint8).reshape(-1) for L in li])
xyz = tuple(np.cumsum(rng.normal(0, 2.2, tot)).astype(np.float32) for _ in range(3))
dbs = reseek_amd.Db(ctx, li, mu=np.concatenate(seqs), prof=prof, xyz=xyz, selfrev=np.zeros(n, np.float32))
cand = np.nonzero((li >= lmin) & (li <= lmax))[0]
order = np.random.default_rng(4).permutation(cand)[:nq].astype(np.uint32)
qa = np.repeat(order, n)
qb = np.tile(np.arange(n, dtype=np.uint32), len(order))
if len(sys.argv) > 4 and sys.argv[4] == "t":
    qa, qb = qb, qa
ctx.align_pairs(dbs, dbs, qa, qb, min_fwd_score=0.0, collect=False)
v = []
for _ in range(3):
    ctx.align_pairs(dbs, dbs, qa, qb, min_fwd_score=0.0, collect=False)
    v.append(ctx.last_kernel_ms())
p_, cells, tb = ctx.align_last_work()
ms = float(np.median(v))
print("L %d..%d nq %d lib %s %s MAXR %s: %.3f ms  %.4f T cells/s" % (lmin, lmax, len(order), os.path.basename(os.path.dirname(os.environ.get("RSK_LIB", "default/"))),
      "shared chain = B" if len(sys.argv) > 4 else "shared chain = A", os.environ.get("RSK_SWQ_MAXR"), ms, cells / ms * 1e3 / 1e12))

#!/usr/bin/env python3
"""Is k_traceback's time its longest path?  64 queries x all chains through rsk_align_pairs with / without the 64 self pairs
(a self pair's path is the whole chain).  Run under rocprofv3 --kernel-trace --stats: compare the k_traceback rows."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import reseek_amd  # noqa: E402

noself = len(sys.argv) > 1 and sys.argv[1] == "noself"
seqs = bench.synth_mu_chains(0x5EED5EEC, None)
n = len(seqs)
li = np.array([len(s) for s in seqs], np.uint32)
rng = np.random.default_rng(11)
tot = int(li.sum())
prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, int(L))), rng.integers(0, 16, (7, int(L)))]).astype(np.uint8).reshape(-1) for L in li])
xyz = tuple(np.cumsum(rng.normal(0, 2.2, tot)).astype(np.float32) for _ in range(3))
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
db = reseek_amd.Db(ctx, li, mu=np.concatenate(seqs), prof=prof, xyz=xyz, selfrev=np.zeros(n, np.float32))
order = np.random.default_rng(4).permutation(n)[:64].astype(np.uint32)
qa = np.repeat(order, n)
qb = np.tile(np.arange(n, dtype=np.uint32), 64)
if noself:
    keep = qa != qb
    qa, qb = qa[keep], qb[keep]
print("pairs", len(qa), "longest query", int(li[order].max()))
for _ in range(3):
    ctx.align_pairs(db, db, qa, qb, min_fwd_score=0.0, collect=False)
    print("kernel_ms", ctx.last_kernel_ms())

#!/usr/bin/env python3
"""Float SW stage alone (rsk_align_pairs: k_sw_qp / k_sw_float + traceback + LDDT) on the survivors of the
-sensitive Mu filter of the SCOP40-shaped synthetic set -- the pair list DBSearcher hands to the aligner.
Usage: bench_align.py [reps] ; kernel-selection knobs come from the environment (RSK_SWQ_*)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import reseek_amd  # noqa: E402


def survivors(ctx, seqs):
    n = len(seqs)
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    out8 = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
    cap = 4_000_000
    pq = torch.zeros(cap, dtype=torch.int32, device="cuda")
    pt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    nn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.mu_filter_dev(db, db, True, 12.0, 20.0, out8.data_ptr(), n, pq.data_ptr(), pt.data_ptr(), 0, 0, cap, nn.data_ptr())
    torch.cuda.synchronize()
    ns = int(nn.item())
    ia = pq[:ns].cpu().numpy().astype(np.uint32)
    ib = pt[:ns].cpu().numpy().astype(np.uint32)
    o = np.lexsort((ib, ia))
    return ia[o], ib[o]


def structure_db(ctx, seqs):
    rng = np.random.default_rng(3)
    lens = np.array([len(s) for s in seqs], np.uint32)
    tot = int(lens.sum())
    prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, int(L))), rng.integers(0, 16, (7, int(L)))]).astype(np.uint8).reshape(-1)
                           for L in lens])
    xyz = tuple(np.cumsum(rng.normal(0, 2.2, tot)).astype(np.float32) for _ in range(3))
    return reseek_amd.Db(ctx, lens, mu=np.concatenate(seqs), prof=prof, xyz=xyz, selfrev=np.zeros(len(seqs), np.float32))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    seqs = bench.synth_mu_chains(0x5EED5EEC)
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    ia, ib = survivors(ctx, seqs)
    dba = structure_db(ctx, seqs)
    ctx.align_pairs(dba, dba, ia, ib, min_fwd_score=7.0)      # warm the allocator pool
    ms, tot = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.align_pairs(dba, dba, ia, ib, min_fwd_score=7.0)
        tot.append((time.perf_counter() - t0) * 1e3)
        ms.append(ctx.last_kernel_ms())
    p_, cells, tb = ctx.align_last_work()
    k = float(np.median(ms))
    print(json.dumps({"env": {k_: v for k_, v in os.environ.items() if k_.startswith("RSK_")}, "pairs": p_, "cells": cells,
                      "sw_kernel_ms": k, "Tcells_per_s": cells / (k * 1e-3) / 1e12, "call_ms_incl_python": float(np.median(tot)),
                      "trace_bytes": tb}))


if __name__ == "__main__":
    main()

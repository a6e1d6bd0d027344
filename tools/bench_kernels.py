#!/usr/bin/env python3
"""Secondary kernel timings on the SCOP40-shaped synthetic set (not the driver's bench line)."""
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


def main():
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    seqs = bench.synth_mu_chains(0x5EED5EEC, nch or None)
    n = len(seqs)
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    lens = np.array([len(s) for s in seqs], np.float64)
    suffix = np.cumsum(lens[::-1])[::-1]
    cells = float((lens * suffix).sum())
    pairs = n * (n + 1) // 2
    out8 = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
    res = {}
    for name, rev in (("mu_sw_fwd", False), ("mu_sw_rev", True)):
        ctx.mu_sw_matrix_dev(db, db, True, rev, out8.data_ptr(), n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.mu_sw_matrix_dev(db, db, True, rev, out8.data_ptr(), n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"ms": dt * 1e3, "kernel_ms": ctx.last_kernel_ms(), "Tcells_per_s": cells / dt / 1e12}
    cap = 40_000_000
    pq = torch.zeros(cap, dtype=torch.int32, device="cuda")
    pt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    nn = torch.zeros(1, dtype=torch.int32, device="cuda")
    for preset, (om, of) in (("sensitive", (12.0, 20.0)), ("fast", (22.0, 50.0))):
        t0 = time.perf_counter()
        ctx.mu_filter_dev(db, db, True, om, of, out8.data_ptr(), n, pq.data_ptr(), pt.data_ptr(), 0, 0, cap, nn.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p, c = ctx.mu_filter_last_work()
        res["mu_filter_" + preset] = {"ms": dt * 1e3, "pairs": p, "rev_candidates": c, "survivors": int(nn.item()),
                                      "pairs_per_s": pairs / dt}
    # float SW + traceback + LDDT on synthetic profiles/coordinates, for the pairs that survived the
    # "sensitive" Mu filter above (the list DBSearcher hands to rsk_align_pairs)
    ctx.mu_filter_dev(db, db, True, 12.0, 20.0, out8.data_ptr(), n, pq.data_ptr(), pt.data_ptr(), 0, 0, cap, nn.data_ptr())
    torch.cuda.synchronize()
    ns = int(nn.item())
    ia = pq[:ns].cpu().numpy().astype(np.uint32)
    ib = pt[:ns].cpu().numpy().astype(np.uint32)
    rng = np.random.default_rng(3)
    lens_al = np.array([len(s) for s in seqs], np.uint32)
    tot = int(lens_al.sum())
    prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, int(L))), rng.integers(0, 16, (7, int(L)))]).astype(np.uint8).reshape(-1)
                           for L in lens_al])
    xyz = tuple(np.cumsum(rng.normal(0, 2.2, tot)).astype(np.float32) for _ in range(3))
    dba = reseek_amd.Db(ctx, lens_al, mu=np.concatenate(seqs), prof=prof, xyz=xyz, selfrev=np.zeros(n, np.float32))
    ctx.align_pairs(dba, dba, ia, ib, min_fwd_score=7.0)      # warm the allocator pool
    t0 = time.perf_counter()
    ctx.align_pairs(dba, dba, ia, ib, min_fwd_score=7.0)
    dt = time.perf_counter() - t0
    p_, cells_al, tb = ctx.align_last_work()
    res["align_pairs"] = {"ms_total_incl_host": dt * 1e3, "sw_kernel_ms": ctx.last_kernel_ms(), "pairs": p_, "cells": cells_al,
                          "Tcells_per_s_kernel": cells_al / (ctx.last_kernel_ms() * 1e-3) / 1e12, "trace_bytes": tb}
    # k-mer prefilter (exact k-mers) on the same synthetic Mu set
    cap = 30_000_000
    dq = torch.zeros(cap, dtype=torch.int32, device="cuda")
    dtt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    dsc = torch.zeros(cap, dtype=torch.int32, device="cuda")
    dn = torch.zeros(1, dtype=torch.int32, device="cuda")
    if n <= 65535:
        ctx.mu_prefilter_dev(db, db, dq.data_ptr(), dtt.data_ptr(), dsc.data_ptr(), cap, dn.data_ptr())
        t0 = time.perf_counter()
        ctx.mu_prefilter_dev(db, db, dq.data_ptr(), dtt.data_ptr(), dsc.data_ptr(), cap, dn.data_ptr())
        torch.cuda.synchronize()
        res["mu_prefilter_exact"] = {"ms": (time.perf_counter() - t0) * 1e3, "kernel_ms": ctx.last_kernel_ms(), "triples": int(dn.item())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

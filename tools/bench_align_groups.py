#!/usr/bin/env python3
"""Float SW throughput when every query has MANY targets (the -verysensitive / large-DB regime: BASELINE configs 3-4):
NQ queries x NT targets, all pairs through rsk_align_pairs (query-profile kernel k_sw_qp)."""
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
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 11211
    seqs = bench.synth_mu_chains(0x5EED5EEC, None)
    rng = np.random.default_rng(3)
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    maxq = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # >0: only queries up to this length (the k_sw_qp LDS limit is 300)
    if maxq:
        head = [s for s in seqs if len(s) <= maxq][:nq]
        seqs = head + [s for s in seqs if not any(s is h for h in head)]
    lens = np.array([len(s) for s in seqs], np.uint32)
    tot = int(lens.sum())
    prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, int(L))), rng.integers(0, 16, (7, int(L)))]).astype(np.uint8).reshape(-1)
                           for L in lens])
    xyz = tuple(np.cumsum(rng.normal(0, 2.2, tot)).astype(np.float32) for _ in range(3))
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    db = reseek_amd.Db(ctx, lens, mu=np.concatenate(seqs), prof=prof, xyz=xyz, selfrev=np.zeros(len(seqs), np.float32))
    ia = np.repeat(np.arange(nq, dtype=np.uint32), nt)
    ib = np.tile(np.arange(nt, dtype=np.uint32), nq)
    res = {}
    for rep in range(2):
        t0 = time.perf_counter()
        ctx.align_pairs(db, db, ia, ib, min_fwd_score=0.0)
        dt = time.perf_counter() - t0
        p_, cells, tb = ctx.align_last_work()
        res["run%d" % rep] = {"ms_total_incl_python": dt * 1e3, "sw_kernel_ms": ctx.last_kernel_ms(), "pairs": p_, "cells": cells,
                              "Tcells_per_s_kernel": cells / (ctx.last_kernel_ms() * 1e-3) / 1e12, "trace_bytes": tb}
    print(json.dumps({"queries": nq, "targets": nt, "query_lengths": lens[:nq].tolist()[:16], **res}, indent=1))


if __name__ == "__main__":
    main()

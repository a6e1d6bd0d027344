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
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

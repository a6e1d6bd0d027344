#!/usr/bin/env python3
"""k-mer prefilter stage alone (rsk_mu_prefilter_dev, idxt neighbourhoods = what `-search -fast -db` runs) on
  syn    : the Mu letters of the seeded 11,211-chain synthetic .bca (BASELINE configs[2] input; low-complexity letters)
  scop40 : the real SCOP40 Mu letters (tests/golden/scop40.mu.fa.gz)
Prints seconds / kernel ms / triples and an order-independent digest of the (query, target, score) triples, so that two
builds of the kernel can be compared on the GPU box.  RSK_PF_DEBUG=1|2 stops the scan after the count / scatter phase."""
import hashlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import reseek_amd
import fixtures as fx


def syn_seqs(nch=0):
    import bench, bench_search
    lens = bench.scop40_lengths()
    rng = np.random.default_rng(7)
    if nch:
        lens = lens[rng.choice(len(lens), nch, replace=nch > len(lens))]
    with tempfile.TemporaryDirectory() as td:
        bca, fa = os.path.join(td, "s.bca"), os.path.join(td, "s.mu.fa")
        bench_search.write_bca(bca, lens, rng)
        reseek_amd.capi.bca_to_mu_fasta(bca, fa)
        return fx.read_mu_fasta(fa)[1]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "syn"
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    seqs = syn_seqs(nch) if which == "syn" else fx.read_mu_fasta("scop40.mu.fa.gz")[1]
    if which != "syn" and nch:
        seqs = seqs[:nch]
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    cap = int(min(len(seqs) * len(seqs) + 16, 200_000_000))
    dq, dt, ds = (torch.zeros(cap, dtype=torch.int32, device="cuda") for _ in range(3))
    dn = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(reps):
        t0 = time.perf_counter()
        ctx.mu_prefilter_dev(q, q, dq.data_ptr(), dt.data_ptr(), ds.data_ptr(), cap, dn.data_ptr(), neighbourhood=2)
        torch.cuda.synchronize()
        print("rep", rep, "seconds %.4f" % (time.perf_counter() - t0), "kernel_ms %.2f" % ctx.last_kernel_ms(), "triples", int(dn.item()), flush=True)
    n = int(dn.item())
    # order-independent digest: sum and xor of a 64-bit mix of each triple
    k = (dq[:n].to(torch.int64) << 40) | (dt[:n].to(torch.int64) << 16) | ds[:n].to(torch.int64)
    mix = (k * 0x9E3779B97F4A7C15 % (1 << 63))
    print("set", which, "chains", len(seqs), "triples", n, "digest_sum", int(mix.sum().item()) & ((1 << 63) - 1), "digest_xor",
          int(torch.bitwise_xor(mix[::2][: n // 2], mix[1::2][: n // 2]).sum().item()) & ((1 << 63) - 1), "score_sum", int(ds[:n].to(torch.int64).sum().item()))


if __name__ == "__main__":
    main()

"""GPU parity of the gapped float X-drop extensions of the long-chain path (second half).  The product has
ONE implementation, k_xdrop_wave (a wave per extension); it is checked against
  * the reference's own per-stage outputs on real long-chain pairs (ref_harness xdrophsp fixtures: palms = the reference's
    test chains, taildb = 48 chains of 17 .. 5,000 residues) -- extensions from the reference's start and the whole device
    batch (start, gates, merge, statistics);
  * the CPU oracle (oracle/rsk_oracle.c rsko_xdrop_*, itself pinned to those fixtures and to the reference's -test_xdrop
    vectors: tests/test_oracle_xdrop.py, tests/test_xdrop_kat.py) on starts no fixture holds: chain edges, random starts,
    other gap penalties, and bands wider than the kernel's LDS ring.
Score bits and paths must be identical."""
import ctypes as C
import struct

import numpy as np
import pytest

import fixtures as fx
import oracle_lib as ol
from reseek_amd import capi

pytestmark = pytest.mark.gpu


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def smx(pa, pb):
    pa, pb = np.ascontiguousarray(pa), np.ascontiguousarray(pb)
    S = np.zeros((pa.shape[1], pb.shape[1]), np.float32)
    ol.lib().rsko_set_smx(pa.ctypes.data_as(C.POINTER(C.c_uint8)), pa.shape[1], pb.ctypes.data_as(C.POINTER(C.c_uint8)), pb.shape[1],
                          S.ctypes.data_as(C.POINTER(C.c_float)))
    return S


@pytest.fixture(scope="module")
def ctx():
    import torch
    import reseek_amd
    assert torch.cuda.is_available()
    c = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


@pytest.mark.parametrize("fixture,X,go,ge", [("palms_sensitive.rskdb.gz", 8.0, -0.685533, -0.051881),
                                             ("q100_sensitive.rskdb.gz", 8.0, -0.685533, -0.051881),
                                             ("q100_sensitive.rskdb.gz", 2.5, -3.0, -1.0),
                                             ("taildb_sensitive.rskdb.gz", 8.0, -0.685533, -0.051881)])
def test_xdrop_pairs_match_the_oracle(ctx, fixture, X, go, ge):
    """rsk_xdrop_pairs against the oracle's XDropFwd / XDropBwd on the SetSMx_NoRev matrix, scores and paths bit for bit."""
    import reseek_amd
    chains = fx.read_rskdb(fixture)
    chains = chains[:14] if not fixture.startswith("taildb") else [c for c in chains if c.prof.shape[1] <= 1600][:14]
    db = reseek_amd.Db.from_chains(ctx, chains)
    rng = np.random.default_rng(5)
    ia, ib, la, lb = [], [], [], []
    n = len(chains)
    for a in range(n):
        for b in rng.choice(n, 4, replace=False):
            LA, LB = chains[a].prof.shape[1], chains[b].prof.shape[1]
            if LA < 3 or LB < 3:
                continue
            # a start on a self-like diagonal (long extensions), a random one, and the edges 1 / L - 1 (extents of one row /
            # one column: xdropfwd.cpp:84-92)
            for (x, y) in ((min(LA, LB) // 2, min(LA, LB) // 2), (int(rng.integers(1, LA)), int(rng.integers(1, LB))), (1, 1), (LA - 1, LB - 1),
                           (1, LB - 1)):
                ia.append(a); ib.append(int(b)); la.append(x); lb.append(y)
    res = ctx.xdrop_pairs(db, db, ia, ib, la, lb, X, go, ge)
    nlong = 0
    cache = {}
    for k, (sf, pf, sb, pb) in enumerate(res):
        key = (ia[k], ib[k])
        if key not in cache:
            cache[key] = smx(chains[ia[k]].prof, chains[ib[k]].prof)
        S = cache[key]
        hf, hpf = ol.xdrop_fwd(S, X, go, ge, la[k], lb[k])
        hb, hpb = ol.xdrop_bwd(S, X, go, ge, la[k] - 1, lb[k] - 1)
        assert bits(sf) == bits(hf) and pf == hpf, (k, "fwd", ia[k], ib[k], la[k], lb[k])
        assert bits(sb) == bits(hb) and pb == hpb, (k, "bwd", ia[k], ib[k], la[k], lb[k])
        nlong += len(pf) > 40 or len(pb) > 40
    assert len(res) > 200 and nlong > 10
    db.close()


def test_band_wider_than_the_lds_ring(ctx):
    """Explicit score matrices whose band outgrows the kernel's 512-column LDS ring (the extension is then re-run on HBM
    rows): mildly positive scores keep every column within X of the best, so the band spans the whole row.  The same entry
    points the reference's -test_xdrop vectors go through (tests/test_xdrop_kat.py), against the oracle."""
    rng = np.random.default_rng(11)
    for (LA, LB, X, go, ge) in ((700, 900, 8.0, -0.685533, -0.051881), (1300, 640, 30.0, -1.0, -0.02), (600, 600, 8.0, -3.0, -1.0)):
        S = (rng.random((LA, LB)) * 0.6 - 0.25).astype(np.float32)
        S[np.arange(min(LA, LB)), np.arange(min(LA, LB))] += 0.5
        for (a, b) in ((1, 1), (LA // 2, LB // 2), (LA - 2, 3)):
            gf, gpf = capi.xdrop_fwd(ctx, S, X, go, ge, a, b)
            of, opf = ol.xdrop_fwd(S, X, go, ge, a, b)
            assert bits(gf) == bits(of) and gpf == opf, (LA, LB, a, b, "fwd")
            gb, gpb = capi.xdrop_bwd(ctx, S, X, go, ge, a, b)
            ob, opb = ol.xdrop_bwd(S, X, go, ge, a, b)
            assert bits(gb) == bits(ob) and gpb == opb, (LA, LB, a, b, "bwd")
    # the widest row of the first case must really have exceeded the ring for this test to mean anything
    S = (np.random.default_rng(11).random((700, 900)) * 0.6 - 0.25).astype(np.float32)
    _, p = ol.xdrop_fwd(S, 8.0, -0.685533, -0.051881, 1, 1)
    assert len(p) > 600


def test_xdrop_pairs_rejects_bad_starts(ctx):
    import reseek_amd
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")[:2]
    db = reseek_amd.Db.from_chains(ctx, chains)
    with pytest.raises(RuntimeError):
        ctx.xdrop_pairs(db, db, [0], [1], [0], [1], 8.0, -0.685533, -0.051881)
    with pytest.raises(RuntimeError):
        ctx.xdrop_pairs(db, db, [0], [1], [1], [chains[1].prof.shape[1]], 8.0, -0.685533, -0.051881)
    db.close()


@pytest.mark.parametrize("name,min_gated,min_aln", [("palms", 500, 500), ("taildb", 900, 900)])
def test_long_chain_stages_match_the_reference(ctx, name, min_gated, min_aln):
    """Direct fixture of the reference's long-chain path (oracle/ref_harness xdrophsp on palms.bca / taildb.bca, every long-chain pair,
    each stage from the reference's own GetMegaHSPScore / StaticSubstScore / XDropFwd / XDropBwd / MergeFwdBwd and checked
    against DSSAligner::AlignMKF inside the harness):
      * rsk_xdrop_pairs from the reference's start -> forward / backward score bits and paths;
      * rsk_mkf_align_pairs from the reference's chained HSPs -> the start it derives (through the merged Lo), total score
        bits, merged path, E-value and LDDT bits, and the MinMegaHSPScore / TotalScore gates."""
    import reseek_amd
    chains = fx.read_rskdb(name + "_sensitive.rskdb.gz")
    n, recs = fx.read_xdrophsp("xdrophsp_" + name + "_sensitive.bin.gz")
    assert n == len(chains)
    db = reseek_amd.Db.from_chains(ctx, chains)
    gated = [r for r in recs if r["gate"]]
    assert len(gated) > min_gated
    # 1. the two extensions from the reference's start
    res = ctx.xdrop_pairs(db, db, [r["i"] for r in gated], [r["j"] for r in gated], [r["lo_a"] for r in gated], [r["lo_b"] for r in gated],
                          8.0, -0.685533, -0.051881)
    for r, (sf, pf, sb, pb) in zip(gated, res):
        assert (bits(sf), pf) == r["fwd"], (r["i"], r["j"], "fwd")
        assert (bits(sb), pb) == r["bwd"], (r["i"], r["j"], "bwd")
    # 2. the whole device batch from the chained HSPs
    have = [r for r in recs if len(r["chain"])]
    first = np.concatenate([[0], np.cumsum([len(r["chain"]) for r in have])]).astype(np.uint32)
    hsp = np.concatenate([r["chain"] for r in have]).astype(np.int32)
    out, status = ctx.mkf_align_pairs(db, db, [r["i"] for r in have], [r["j"] for r in have], first, hsp[:, 0], hsp[:, 1], hsp[:, 2],
                                      x2=8.0, min_mega_score=-4.0, min_fwd_score=7.0)
    naln = 0
    for r, (a, path), st in zip(have, out, status):
        key = (r["i"], r["j"])
        if not r["gate"] or r["best_chain_score"] <= 0:
            assert st == 0 or r["best_chain_score"] <= 0, key          # the host drops BestChainScore <= 0 before the batch
            if r["gate"] == 0:
                assert a.path_len == 0, key
            continue
        assert st == 1, key
        assert bits(a.score) == r["total"] and path == r["path"], key
        if r["path"]:
            assert (a.lo_a, a.lo_b) == (r["mlo_a"], r["mlo_b"]), key
            assert bits(a.evalue) == r["evalue"] and bits(a.lddt) == r["lddt"], key
            naln += 1
    assert naln > min_aln
    db.close()

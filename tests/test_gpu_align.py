"""GPU parity: fused score matrix + float affine SW + traceback + LDDT/E-value
through the C-ABI vs the reference's own per-pair outputs (tests/golden)."""
import struct

import numpy as np
import pytest

import fixtures as fx
import oracle_lib as ol

pytestmark = pytest.mark.gpu
FLT_MAX = 3.4028234663852886e38


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


@pytest.fixture(scope="module")
def ctx():
    import torch
    import reseek_amd
    assert torch.cuda.is_available()
    c = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=["per-pair kernel for small groups (default)", "query-profile kernel for every group",
                                      "query-profile kernel, at most 5 rows per lane (an odd row count; passes and LDS segments for every chain above 80 / 160 residues)",
                                      "query-profile kernel, at most 8 rows per lane (two passes per LDS segment)"])
def kernel_choice(request, monkeypatch):
    """rsk_align_pairs sends a group of pairs that share a chain to k_sw_qp only when the group fills ~4 waves
    (RSK_SWQ_MIN_LANES); every test here runs under the default and with the threshold at 1, so both float-SW kernels see
    all the cases -- and k_sw_qp with its rows per lane capped (RSK_SWQ_MAXR: 12 by default; a chain of L residues runs in
    ceil(L / 16R) passes of 16 strips, 4 / ceil(R / 4) passes per LDS profile), which walks the odd-row trace store, the
    boundary rows between passes and the segment hand-over on chains of ordinary length."""
    for v in ("RSK_SWQ_MIN_LANES", "RSK_SWQ_MAXR"):
        monkeypatch.delenv(v, raising=False)
    if request.param.startswith("query-profile"):
        monkeypatch.setenv("RSK_SWQ_MIN_LANES", "1")
    if "at most 5" in request.param:
        monkeypatch.setenv("RSK_SWQ_MAXR", "5")
    if "at most 8" in request.param:
        monkeypatch.setenv("RSK_SWQ_MAXR", "8")


def check_against_records(ctx, chains, recs, min_fwd):
    import reseek_amd
    db = reseek_amd.Db.from_chains(ctx, chains)
    ia = [r["i"] for r in recs]
    ib = [r["j"] for r in recs]
    res = ctx.align_pairs(db, db, ia, ib, min_fwd_score=min_fwd)
    nstats = 0
    for r, (al, path) in zip(recs, res):
        key = (r["i"], r["j"], r["LA"], r["LB"])
        assert bits(al.score) == bits(r["sw"]), key
        assert path == r["path"], key
        if path:
            assert (al.lo_a, al.lo_b) == (r["loA"], r["loB"]), key
        if bits(r["evalue"]) == bits(FLT_MAX):
            assert bits(al.evalue) == bits(FLT_MAX), key
            continue
        nstats += 1
        assert (al.hi_a, al.hi_b, al.ids, al.gaps) == (r["hiA"], r["hiB"], r["ids"], r["gaps"]), key
        for name in ("lddt", "ts", "pvalue", "evalue", "qual"):
            assert bits(getattr(al, name)) == bits(r[name]), (key, name)
    db.close()
    return len(recs), nstats


def test_q100_all_pairs_sensitive(ctx):
    """5,0xx real chain pairs: score bits, lo, CIGAR path, hi/ids/gaps, LDDT, TS, P, E, quality."""
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    _, recs = fx.read_pairs("pairs_q100_sensitive.bin.gz")
    n, nstats = check_against_records(ctx, chains, recs, 7.0)
    assert n > 4900 and nstats > 1000


def test_q32_all_pairs_verysensitive(ctx):
    chains = fx.read_rskdb("q100_verysensitive.rskdb.gz")
    _, recs = fx.read_pairs("pairs_q32_verysensitive.bin.gz")
    n, nstats = check_against_records(ctx, chains, recs, 0.0)
    assert n == 528 and nstats == 528


def test_role_order_matters_and_is_respected(ctx):
    """Aligning (B, A) is not the transpose of (A, B) in general (tie order); both must equal the oracle."""
    import reseek_amd
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")[:24]
    db = reseek_amd.Db.from_chains(ctx, chains)
    ia, ib = np.meshgrid(np.arange(24), np.arange(24), indexing="ij")
    res = ctx.align_pairs(db, db, ia.ravel(), ib.ravel(), min_fwd_score=0.0)
    for a, b, (al, path) in zip(ia.ravel(), ib.ravel(), res):
        s, lo_i, lo_j, opath = ol.align_pair(chains[a].prof, chains[b].prof)
        assert bits(al.score) == bits(s) and path == opath
        if path:
            assert (al.lo_a, al.lo_b) == (lo_i, lo_j)
    db.close()


def test_both_chains_longer_than_one_strip_group(ctx):
    """> 1024 rows on both sides: the kernel runs several 64-strip row groups (boundary via HBM)."""
    import copy
    import reseek_amd
    base = fx.read_rskdb("q100_sensitive.rskdb.gz")[0]
    rng = np.random.default_rng(0)
    cs = []
    for L in (1100, 2300):
        c = copy.copy(base)
        c.mu = rng.integers(0, 36, L).astype(np.uint8)
        c.prof = np.concatenate([rng.integers(0, 20, (1, L)), rng.integers(0, 16, (7, L))]).astype(np.uint8)
        c.prof[:, 100:400] = c.prof[:, 500:800]          # internal repeat -> ties / long paths
        c.x = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        c.y = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        c.z = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        cs.append(c)
    db = reseek_amd.Db.from_chains(ctx, cs)
    res = ctx.align_pairs(db, db, [0, 0, 1, 1], [0, 1, 0, 1], min_fwd_score=0.0)
    for (a, b), (al, path) in zip([(0, 0), (0, 1), (1, 0), (1, 1)], res):
        s, lo_i, lo_j, opath = ol.align_pair(cs[a].prof, cs[b].prof)
        assert bits(al.score) == bits(s) and path == opath and (al.lo_a, al.lo_b) == (lo_i, lo_j), (a, b)
        ok, st = ol.calc_evalue(s, 0.0, opath, lo_i, lo_j, cs[a], cs[b])
        assert bits(al.lddt) == bits(st.lddt) and bits(al.evalue) == bits(st.evalue)
    db.close()


def test_long_alignments_without_a_paths_buffer(ctx):
    """rsk_align_pairs with paths = NULL (how DBSearcher::ComputeSelfRevScores and the test statistic of batches that keep
    no paths call it): alignments of more than 256 / 1024 columns take k_lddt_long over the list k_lddt builds on the
    device -- the statistics kernels run whether or not the caller wants the paths, so the list must be sized either way (r05
    regression: illegal access at config-3 scale); scores equal the call with a paths buffer, whose LDDT / E-value equal the oracle."""
    import copy
    import ctypes as C
    import reseek_amd
    from reseek_amd import capi
    base = fx.read_rskdb("q100_sensitive.rskdb.gz")[0]
    rng = np.random.default_rng(12)
    cs = []
    for L in (300, 700, 1400, 90):
        c = copy.copy(base)
        c.mu = rng.integers(0, 36, L).astype(np.uint8)
        c.prof = np.concatenate([rng.integers(0, 20, (1, L)), rng.integers(0, 16, (7, L))]).astype(np.uint8)
        c.x = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        c.y = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        c.z = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        cs.append(c)
    db = reseek_amd.Db.from_chains(ctx, cs)
    ia = np.array([0, 1, 2, 3, 0, 2], np.uint32)          # self pairs: 300, 700, 1400 and 90 aligned columns; two unrelated pairs
    ib = np.array([0, 1, 2, 3, 1, 3], np.uint32)
    with_paths = ctx.align_pairs(db, db, ia, ib, min_fwd_score=0.0)
    out = (capi.Aln * len(ia))()
    capi._check(capi.lib().rsk_align_pairs(ctx.h, db.h, db.h, capi._p(ia, capi.u32p), capi._p(ib, capi.u32p), len(ia), capi.GAP_OPEN, capi.GAP_EXT,
                                           0.0, out, None, 0))
    for k, (al, path) in enumerate(with_paths):
        assert bits(out[k].score) == bits(al.score) and (out[k].lo_a, out[k].lo_b) == (al.lo_a, al.lo_b), k      # (no paths: the statistics are not reported)
        s, lo_i, lo_j, opath = ol.align_pair(cs[ia[k]].prof, cs[ib[k]].prof)
        ok, st = ol.calc_evalue(s, 0.0, opath, lo_i, lo_j, cs[ia[k]], cs[ib[k]])
        assert path == opath and bits(al.lddt) == bits(st.lddt) and bits(al.evalue) == bits(st.evalue), k
    assert [al.path_len for al, _ in with_paths[:4]] == [300, 700, 1400, 90]
    db.close()


def test_tiny_and_ragged_chains(ctx):
    import reseek_amd
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    import copy
    small = []
    for k, L in enumerate([1, 2, 15, 16, 17, 31, 33]):
        c = copy.copy(chains[k])
        c.mu, c.prof = c.mu[:L].copy(), c.prof[:, :L].copy()
        c.x, c.y, c.z = c.x[:L].copy(), c.y[:L].copy(), c.z[:L].copy()
        c.seq = c.seq[:L]
        small.append(c)
    db = reseek_amd.Db.from_chains(ctx, small + chains[:5])
    n = len(small) + 5
    ia, ib = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    res = ctx.align_pairs(db, db, ia.ravel(), ib.ravel(), min_fwd_score=0.0)
    allc = small + chains[:5]
    for a, b, (al, path) in zip(ia.ravel(), ib.ravel(), res):
        s, lo_i, lo_j, opath = ol.align_pair(allc[a].prof, allc[b].prof)
        assert bits(al.score) == bits(s) and path == opath, (a, b)
        ok, st = ol.calc_evalue(s, 0.0, opath, lo_i, lo_j, allc[a], allc[b])
        if opath:
            assert bits(al.evalue) == bits(st.evalue) and bits(al.lddt) == bits(st.lddt)
    db.close()


def test_long_strip_chains_in_groups(ctx):
    """Strip chains with many partners at the edges of k_sw_qp's geometry (16 lanes per pair, R = ceil(L / 16P) rows per lane,
    R <= 12): 63..65 (R = 4 with idle lanes / R = 5), 97 (R = 7: an odd row count, the last quad of a lane half used),
    191..193 (one pass of R = 12 / two passes of R = 7), 256 / 257 (two passes in one LDS profile / R = 9: a second
    segment), 300.., 384 / 385 (two passes of R = 12 / three passes), 599, 601, 913 (passes and segments whose rows meet
    through HBM).  Both orientations (long chain first / second), a repeat for ties."""
    import copy
    import reseek_amd
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    rng = np.random.default_rng(7)
    longs = []
    for L in (63, 64, 65, 97, 191, 192, 193, 256, 257, 300, 301, 312, 384, 385, 599, 601, 913):
        c = copy.copy(chains[0])
        c.mu = rng.integers(0, 36, L).astype(np.uint8)
        c.prof = np.concatenate([rng.integers(0, 20, (1, L)), rng.integers(0, 16, (7, L))]).astype(np.uint8)
        src = chains[len(longs) % 12 + 1].prof
        reps = (L + src.shape[1] - 1) // src.shape[1]
        c.prof[:, :] = np.tile(src, (1, reps))[:, :L]      # real profile, repeated: long alignments across segments
        c.x = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        c.y = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        c.z = np.cumsum(rng.normal(0, 2.2, L)).astype(np.float32)
        c.seq = (c.seq * (L // max(1, len(c.seq)) + 1))[:L]
        longs.append(c)
    others = chains[1:13]
    allc = longs + others
    db = reseek_amd.Db.from_chains(ctx, allc)
    nl, n = len(longs), len(allc)
    pairs = [(a, b) for a in range(nl) for b in range(n)] + [(b, a) for a in range(nl) for b in range(nl, n)]
    res = ctx.align_pairs(db, db, [p[0] for p in pairs], [p[1] for p in pairs], min_fwd_score=0.0)
    nlong = 0
    for (a, b), (al, path) in zip(pairs, res):
        s, lo_i, lo_j, opath = ol.align_pair(allc[a].prof, allc[b].prof)
        assert bits(al.score) == bits(s) and path == opath, (a, b)
        if opath:
            assert (al.lo_a, al.lo_b) == (lo_i, lo_j), (a, b)
            ok, st = ol.calc_evalue(s, 0.0, opath, lo_i, lo_j, allc[a], allc[b])
            assert bits(al.lddt) == bits(st.lddt) and bits(al.evalue) == bits(st.evalue), (a, b)
            nlong += len(opath) > 300
    assert nlong >= 4
    db.close()

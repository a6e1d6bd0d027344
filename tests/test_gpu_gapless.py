"""GPU parity: gapless integer Mu kernel through the C-ABI vs the CPU oracle
and vs the reference's own outputs (tests/golden)."""
import numpy as np
import pytest

import fixtures as fx
import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    import reseek_amd
    assert torch.cuda.is_available()
    c = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def run_matrix(ctx, seqs_q, seqs_t=None, tri=False):
    import torch
    import reseek_amd
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs_q)
    t = q if seqs_t is None else reseek_amd.Db.from_mu_seqs(ctx, seqs_t)
    nq, nt = len(seqs_q), (len(seqs_q) if seqs_t is None else len(seqs_t))
    out = torch.full((nq, nt), -1, dtype=torch.int16, device="cuda")
    ctx.mu_gapless_matrix_dev(q, t, tri, out.data_ptr(), nt)
    torch.cuda.synchronize()
    res = out.cpu().numpy().view(np.uint16).astype(np.int32)
    q.close()
    if t is not q:
        t.close()
    return res


def test_q100_matches_reference_goldens(ctx):
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    _, recs = fx.read_pairs("pairs_q100_sensitive.bin.gz")
    seqs = [c.mu for c in chains]
    got = run_matrix(ctx, seqs, tri=True)
    for r in recs:
        assert got[r["i"], r["j"]] == r["gapless_fwd"], (r["i"], r["j"])
    # reversed queries = the "rev" half of AlignMu_Int (dssaligner.cpp:1069-1086)
    rev = run_matrix(ctx, [s[::-1].copy() for s in seqs], seqs)
    for r in recs:
        assert rev[r["i"], r["j"]] == r["gapless_rev"]


def test_pairs_api_positions_match_reference(ctx):
    import reseek_amd
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    _, recs = fx.read_pairs("pairs_q100_sensitive.bin.gz")
    db = reseek_amd.Db.from_mu_seqs(ctx, [c.mu for c in chains])
    ia = np.array([r["i"] for r in recs], np.uint32)
    ib = np.array([r["j"] for r in recs], np.uint32)
    sc, bi, bj = ctx.mu_gapless_pairs(db, db, ia, ib, positions=True)
    assert np.array_equal(sc, np.array([r["gli"] for r in recs]))
    assert np.array_equal(bi, np.array([r["gli_besti"] for r in recs], np.uint32))
    assert np.array_equal(bj, np.array([r["gli_bestj"] for r in recs], np.uint32))
    db.close()


def test_scop40_real_sequences_full_matrix(ctx):
    seqs, tab = fx.read_mukat("mukat_scop40_160.bin.gz")
    got = run_matrix(ctx, seqs, seqs)
    assert np.array_equal(got, tab[:, :, 2])


def test_random_lengths_vs_oracle_including_long_and_tiny(ctx):
    rng = np.random.default_rng(11)
    lens = list(rng.integers(1, 60, 40)) + list(rng.integers(100, 700, 60)) + [1, 2, 7, 8, 9, 1015, 1016, 1023, 1024, 1419, 2500]
    seqs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in lens]
    # plant homologs so that scores are not all tiny
    for k in range(0, 60, 3):
        src = seqs[40 + k]
        m = src.copy()
        idx = rng.random(len(m)) < 0.3
        m[idx] = rng.integers(0, 36, int(idx.sum()))
        seqs[40 + k + 1] = m
    got = run_matrix(ctx, seqs, tri=True)
    ia, ib = np.triu_indices(len(seqs))
    want = ol.mu_gapless_pairs(seqs, ia, ib)
    assert np.array_equal(got[ia, ib], want)
    assert want.max() > 300


def test_rectangular_query_vs_db(ctx):
    rng = np.random.default_rng(5)
    qs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(20, 300, 17)]
    ts = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(5, 500, 301)]
    got = run_matrix(ctx, qs, ts)
    ia, ib = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    want = ol.mu_gapless_pairs(qs + ts, ia.ravel(), ib.ravel() + len(qs)).reshape(len(qs), len(ts))
    assert np.array_equal(got, want)


def test_low_complexity_high_scores(ctx):
    # identical poly-letter chains: score = 4 * L for letters whose self score is 4 (max of the matrix)
    seqs = [np.full(L, 1, np.uint8) for L in (50, 333, 1000, 1023)]
    got = run_matrix(ctx, seqs, tri=True)
    ia, ib = np.triu_indices(len(seqs))
    assert np.array_equal(got[ia, ib], ol.mu_gapless_pairs(seqs, ia, ib))
    assert got[3, 3] == 4 * 1023


def test_scop40_scale_properties(ctx):
    """BASELINE config[1] shape (11,211 chains, SCOP40 lengths): size-independent properties --
    symmetry of the score under swapping roles, self score == sum of diagonal self scores, and a
    seeded sample of pairs bit-exact vs the oracle."""
    import torch
    import reseek_amd
    lens = fx.scop40_lengths()
    rng = np.random.default_rng(2024)
    order = np.argsort(lens, kind="stable")
    seqs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in lens[order]]
    n = len(seqs)
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    out = torch.zeros((n, n), dtype=torch.int16, device="cuda")
    ctx.mu_gapless_matrix_dev(db, db, False, out.data_ptr(), n)
    torch.cuda.synchronize()
    assert bool((out == out.T).all())            # IntScoreMx_Mu is symmetric => score(A,B) == score(B,A)
    got = out.cpu().numpy().view(np.uint16).astype(np.int32)
    ia = rng.integers(0, n, 3000)
    ib = rng.integers(0, n, 3000)
    assert np.array_equal(got[ia, ib], ol.mu_gapless_pairs(seqs, ia, ib))
    pairs, cells, slots = ctx.mu_gapless_last_work()
    assert pairs == n * n and cells == int(lens.sum()) ** 2 and slots >= cells
    db.close()


def test_scop40_scale_triangle_mode_vs_oracle(ctx):
    """The configuration bench.py times (BASELINE configs[1]): 11,211 chains sorted by length, self_triangle = True.  A seeded
    sample of 3,000 pairs i <= j of the triangle bit-exact vs the oracle, the work counters of the triangle, and equality
    with the rectangular pass on the upper triangle (the score is symmetric)."""
    import torch
    import reseek_amd
    lens = fx.scop40_lengths()
    rng = np.random.default_rng(2025)
    order = np.argsort(lens, kind="stable")
    seqs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in lens[order]]
    n = len(seqs)
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    out = torch.zeros((n, n), dtype=torch.int16, device="cuda")
    ctx.mu_gapless_matrix_dev(db, db, True, out.data_ptr(), n)
    torch.cuda.synchronize()
    pairs, cells, slots = ctx.mu_gapless_last_work()
    L = lens[order].astype(np.int64)
    assert pairs == n * (n + 1) // 2 and cells == int((L * np.cumsum(L[::-1])[::-1]).sum()) and slots >= cells
    a = rng.integers(0, n, 3000)
    b = rng.integers(0, n, 3000)
    ia, ib = np.minimum(a, b), np.maximum(a, b)
    got = out[torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda()].cpu().numpy().view(np.uint16).astype(np.int32)
    assert np.array_equal(got, ol.mu_gapless_pairs(seqs, ia, ib))
    rect = torch.zeros((n, n), dtype=torch.int16, device="cuda")
    ctx.mu_gapless_matrix_dev(db, db, False, rect.data_ptr(), n)
    torch.cuda.synchronize()
    iu = torch.triu_indices(n, n, device="cuda")
    assert bool((out[iu[0], iu[1]] == rect[iu[0], iu[1]]).all())
    db.close()


def test_uint16_range_contract(ctx):
    """rsk_mu_gapless_matrix_dev writes uint16: a pair of chains longer than 16383 could exceed 65535 -> RSK_E_RANGE (the header's
    contract); one such chain against short ones is fine (score <= 4 * min(LA, LB))."""
    import torch
    import reseek_amd
    from reseek_amd import capi
    rng = np.random.default_rng(3)
    big = [rng.integers(0, 36, 16400).astype(np.uint8), rng.integers(0, 36, 50).astype(np.uint8)]
    small = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in (30, 200, 999)]
    dbig, dsmall = reseek_amd.Db.from_mu_seqs(ctx, big), reseek_amd.Db.from_mu_seqs(ctx, small)
    out = torch.zeros((2, 2), dtype=torch.int16, device="cuda")
    with pytest.raises(capi.RskError):
        ctx.mu_gapless_matrix_dev(dbig, dbig, True, out.data_ptr(), 2)
    got = run_matrix(ctx, big, small)
    ia, ib = np.meshgrid(np.arange(2), np.arange(3), indexing="ij")
    want = ol.mu_gapless_pairs(big + small, ia.ravel(), ib.ravel() + 2).reshape(2, 3)
    assert np.array_equal(got, want)
    dbig.close(); dsmall.close()

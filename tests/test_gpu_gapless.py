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


def test_ring_geometry_edges(ctx):
    """The ring kernel's layout cases: sub-rings that hit the 64-member cap (tiny chains), a small (D = 8) ring, a ring whose
    second sub-ring is empty, query blocks of every size modulo 4 next to chains that fill a whole sub-ring (1019 .. 1023
    residues), and targets of every length modulo 8 (whole 8-letter words, then the tail two letters at a time)."""
    rng = np.random.default_rng(23)
    # 150 chains of 1 .. 6 residues: blocks of 4 or 8 slots, more than 64 members in 1024 slots
    tiny = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(1, 7, 150)]
    got = run_matrix(ctx, tiny, tri=True)
    ia, ib = np.triu_indices(len(tiny))
    assert np.array_equal(got[ia, ib], ol.mu_gapless_pairs(tiny, ia, ib))
    # one chain per sub-ring, an odd number of sub-rings, all block remainders
    lens = [1019, 1020, 1021, 1022, 1023, 3, 4, 5, 6, 7, 8, 130, 131, 132, 133, 509, 510, 511, 512]
    big = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in lens]
    big[6] = big[11][40:44].copy()                       # a 4-residue chain that matches inside a longer one
    got = run_matrix(ctx, big, tri=True)
    ia, ib = np.triu_indices(len(big))
    assert np.array_equal(got[ia, ib], ol.mu_gapless_pairs(big, ia, ib))
    # every target length 1 .. 40 against a few queries (rectangular mode, small target blocks)
    qs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in (37, 200, 64, 9)]
    ts = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in range(1, 41)]
    ts += [q[: 1 + k].copy() for k, q in enumerate(qs * 3)] + [np.concatenate([qs[1], qs[1]])[:333]]
    got = run_matrix(ctx, qs, ts)
    iq, it = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    want = ol.mu_gapless_pairs(qs + ts, iq.ravel(), it.ravel() + len(qs)).reshape(len(qs), len(ts))
    assert np.array_equal(got, want)


def test_many_target_blocks_few_queries(ctx):
    """A handful of queries against several thousand targets: the launch picks small target blocks (many work items of few
    rings); scores against the oracle on a sample, the whole matrix against the pair-list kernel."""
    import reseek_amd
    rng = np.random.default_rng(29)
    qs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(30, 400, 9)]
    ts = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(1, 300, 5000)]
    for k in range(0, 5000, 50):
        ts[k] = qs[k % 9][: max(1, len(qs[k % 9]) - k % 7)].copy()
    got = run_matrix(ctx, qs, ts)
    sel_q = rng.integers(0, 9, 3000)
    sel_t = rng.integers(0, 5000, 3000)
    want = ol.mu_gapless_pairs(qs + ts, sel_q, sel_t + len(qs))
    assert np.array_equal(got[sel_q, sel_t], want)
    q = reseek_amd.Db.from_mu_seqs(ctx, qs)
    t = reseek_amd.Db.from_mu_seqs(ctx, ts)
    iq, it = np.meshgrid(np.arange(9, dtype=np.uint32), np.arange(5000, dtype=np.uint32), indexing="ij")
    sc, _, _ = ctx.mu_gapless_pairs(q, t, iq.ravel(), it.ravel(), positions=True)
    assert np.array_equal(got.ravel(), np.minimum(sc, 65535))
    q.close(); t.close()


def test_rectangular_query_vs_db(ctx):
    rng = np.random.default_rng(5)
    qs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(20, 300, 17)]
    ts = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(5, 500, 301)]
    got = run_matrix(ctx, qs, ts)
    ia, ib = np.meshgrid(np.arange(len(qs)), np.arange(len(ts)), indexing="ij")
    want = ol.mu_gapless_pairs(qs + ts, ia.ravel(), ib.ravel() + len(qs)).reshape(len(qs), len(ts))
    assert np.array_equal(got, want)


def test_thin_target_set_of_long_chains(ctx):
    """Many queries against a thin set of long targets (a rank's shard of the longest chains, reseek_amd/shardplan.py): dense
    matrix and hit records with index bases against the oracle; includes a query and a target beyond a ring (per-pair
    kernel) and a one-residue chain on each side."""
    import torch
    import reseek_amd
    rng = np.random.default_rng(31)
    qs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in np.concatenate([rng.integers(20, 260, 598), [1, 1100]])]
    ts = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in np.concatenate([rng.integers(400, 1000, 58), [1, 1300]])]
    for k in range(0, 58, 5):
        ts[k][:len(qs[k])] = qs[k]                               # planted copies: real hits
    nq, nt = len(qs), len(ts)
    ia, ib = np.meshgrid(np.arange(nq), np.arange(nt), indexing="ij")
    want = ol.mu_gapless_pairs(qs + ts, ia.ravel(), ib.ravel() + nq).reshape(nq, nt)
    q, t = reseek_amd.Db.from_mu_seqs(ctx, qs), reseek_amd.Db.from_mu_seqs(ctx, ts)
    cap = 1 << 16
    thr = 80
    for _ in range(1):
        dense = torch.full((nq, nt), -1, dtype=torch.int16, device="cuda")
        rec = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        ctx.mu_gapless_hits_dev(q, t, False, thr, rec.data_ptr(), cap, cnt.data_ptr(), d_scores_ptr=dense.data_ptr(), ldo=nt, q_base=1000, t_base=5000)
        torch.cuda.synchronize()
        D = dense.cpu().numpy().view(np.uint16).astype(np.int32)
        assert np.array_equal(D, want)
        got = sorted(map(tuple, rec[:int(cnt.item())].cpu().numpy().tolist()))
        qi, ti = np.nonzero(want >= thr)
        assert got == sorted(zip((qi + 1000).tolist(), (ti + 5000).tolist(), want[qi, ti].tolist())) and len(got) >= 12
    assert ctx.mu_gapless_last_work()[0] == nq * nt
    q.close(); t.close()


def test_low_complexity_high_scores(ctx):
    # identical poly-letter chains: score = 4 * L for letters whose self score is 4 (max of the matrix)
    seqs = [np.full(L, 1, np.uint8) for L in (50, 333, 1000, 1023)]
    got = run_matrix(ctx, seqs, tri=True)
    ia, ib = np.triu_indices(len(seqs))
    assert np.array_equal(got[ia, ib], ol.mu_gapless_pairs(seqs, ia, ib))
    assert got[3, 3] == 4 * 1023


def test_scores_around_the_half_float_ceiling(ctx):
    """The ring kernel keeps scores as n / 2048 in half floats whose clamp tops out at 2048; a pair that reads 2048 is scored
    again in integers by the wave that found it.  Chains whose scores straddle the ceiling (2044, 2048, 2052, 3000+),
    triangle and rectangle, dense matrix and hit records."""
    import torch
    import reseek_amd
    rng = np.random.default_rng(8)
    seqs = [np.full(L, 1, np.uint8) for L in (511, 512, 513, 760)]           # self score 4 per residue
    mixed = rng.integers(0, 36, 900).astype(np.uint8)
    seqs += [mixed, np.concatenate([mixed[:700], rng.integers(0, 36, 150).astype(np.uint8)])]      # a long near-copy
    seqs += [rng.integers(0, 36, int(L)).astype(np.uint8) for L in (40, 300, 1010)]
    got = run_matrix(ctx, seqs, tri=True)
    ia, ib = np.triu_indices(len(seqs))
    want = ol.mu_gapless_pairs(seqs, ia, ib)
    assert np.array_equal(got[ia, ib], want)
    assert (want == 2044).any() and (want == 2048).any() and (want == 2052).any() and (want > 2500).sum() >= 2
    rect = run_matrix(ctx, seqs[:5], seqs[3:])
    qa, qb = np.meshgrid(np.arange(5), np.arange(len(seqs) - 3), indexing="ij")
    assert np.array_equal(rect, ol.mu_gapless_pairs(seqs, qa.ravel(), qb.ravel() + 3).reshape(rect.shape))
    c2 = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    db = reseek_amd.Db.from_mu_seqs(c2, seqs)
    rec = torch.zeros((256, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    c2.mu_gapless_hits_dev(db, db, True, 2000, rec.data_ptr(), 256, cnt.data_ptr())
    torch.cuda.synchronize()
    k = int(cnt.item())
    keep = want >= 2000
    assert sorted(map(tuple, rec[:k].cpu().numpy().tolist())) == sorted(zip(ia[keep].tolist(), ib[keep].tolist(), want[keep].tolist()))
    db.close(); c2.close()


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


def test_device_hit_records_equal_the_thresholded_matrix():
    """rsk_mu_gapless_hits_dev: the kernel appends {q, t, score} for every pair reaching min_score (what a search keeps of
    the pair space) -- must be exactly the entries >= min_score of the dense matrix, each unordered pair once in triangle
    mode (pairs of two members of one ring are scored twice there), also without a dense matrix, also with index bases
    (shards), also when the record buffer is too small (count stays exact)."""
    import torch
    import reseek_amd
    rng = np.random.default_rng(23)
    lens = np.concatenate([rng.integers(5, 400, 700), [1030, 1500, 9, 1]])          # two chains beyond a ring: per-pair kernel
    seqs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in lens]
    for k in range(0, 200, 7):                                                       # planted near-copies: real hits
        src = seqs[k]
        seqs[k + 1] = src[:len(seqs[k + 1])].copy() if len(src) >= len(seqs[k + 1]) else np.concatenate([src, seqs[k + 1][len(src):]])
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    n = len(seqs)
    dense = torch.zeros((n, n), dtype=torch.int32, device="cuda").to(torch.uint16)
    ctx.mu_gapless_matrix_dev(db, db, True, dense.data_ptr(), n)
    torch.cuda.synchronize()
    D = dense.cpu().numpy().astype(np.int64)
    thr = 60
    ii, jj = np.triu_indices(n)
    keep = D[ii, jj] >= thr
    want = sorted(zip(ii[keep].tolist(), jj[keep].tolist(), D[ii, jj][keep].tolist()))
    assert 50 < len(want) < 60000
    cap = 1 << 17
    rec = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for with_dense in (False, True):
        d2 = torch.zeros((n, n), dtype=torch.int32, device="cuda").to(torch.uint16) if with_dense else None
        ctx.mu_gapless_hits_dev(db, db, True, thr, rec.data_ptr(), cap, cnt.data_ptr(), d_scores_ptr=d2.data_ptr() if with_dense else 0, ldo=n)
        torch.cuda.synchronize()
        m = int(cnt.item())
        got = sorted(map(tuple, rec[:m].cpu().numpy().tolist()))
        assert got == want, (m, len(want))
        if with_dense:
            assert np.array_equal(d2.cpu().numpy().astype(np.int64)[ii, jj], D[ii, jj])
    # rectangular block with bases, as a shard of a larger set emits them
    qa = reseek_amd.Db.from_mu_seqs(ctx, seqs[:300])
    tb = reseek_amd.Db.from_mu_seqs(ctx, seqs[300:])
    ctx.mu_gapless_hits_dev(qa, tb, False, thr, rec.data_ptr(), cap, cnt.data_ptr(), q_base=0, t_base=300)
    torch.cuda.synchronize()
    m = int(cnt.item())
    got = sorted(map(tuple, rec[:m].cpu().numpy().tolist()))
    assert got == [w for w in want if w[0] < 300 <= w[1]]
    # a buffer that is too small: the count is still exact, the stored records are a subset
    ctx.mu_gapless_hits_dev(db, db, True, thr, rec.data_ptr(), 10, cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(cnt.item()) == len(want)
    assert set(map(tuple, rec[:10].cpu().numpy().tolist())) <= set(want)
    db.close(); qa.close(); tb.close(); ctx.close()


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_shard_windows_of_the_triangle_tile_it_exactly(shards):
    """rsk_mu_gapless_shard_window / rsk_mu_gapless_hits_window_dev (multi-GPU share of the self search): the windows of all
    shards are contiguous and cover the set's positions; the union of the shards' hit records equals the records of the whole
    triangle, no pair twice; the union of the cells the shards write into the dense n x n matrix equals the whole triangle's
    matrix (a cell two shards write -- two members of one ring -- gets the same value); pairs / cells of the shards add up to the triangle's; chains beyond a
    ring (per-pair kernel) and an empty window included."""
    import torch
    import reseek_amd
    rng = np.random.default_rng(41)
    lens = np.concatenate([rng.integers(5, 400, 900), [1030, 1500, 9, 1]])
    rng.shuffle(lens)
    seqs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in lens]
    for k in range(0, 300, 7):
        src = seqs[k]
        seqs[k + 1] = src[:len(seqs[k + 1])].copy() if len(src) >= len(seqs[k + 1]) else np.concatenate([src, seqs[k + 1][len(src):]])
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    n = len(seqs)
    thr, cap = 60, 1 << 17
    rec = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    full = torch.full((n, n), -1, dtype=torch.int16, device="cuda")
    ctx.mu_gapless_hits_dev(db, db, True, thr, rec.data_ptr(), cap, cnt.data_ptr(), d_scores_ptr=full.data_ptr(), ldo=n, q_base=7, t_base=7)
    torch.cuda.synchronize()
    want = sorted(map(tuple, rec[:int(cnt.item())].cpu().numpy().tolist()))
    F = full.cpu().numpy()
    work_full = ctx.mu_gapless_last_work()
    assert len(want) > 40
    ii, jj = np.triu_indices(n)
    assert (F[ii, jj] != -1).all()
    wins = [ctx.mu_gapless_shard_window(db, k, shards) for k in range(shards)]
    assert wins[0][0] == 0 and wins[-1][1] == n and all(a[1] == b[0] for a, b in zip(wins, wins[1:]))
    got, pairs, cells = [], 0, 0
    U = np.full((n, n), -1, np.int16)
    for lo, hi in wins + [(5, 5)]:
        part = torch.full((n, n), -1, dtype=torch.int16, device="cuda")
        ctx.mu_gapless_hits_window_dev(db, lo, hi, thr, rec.data_ptr(), cap, cnt.data_ptr(), d_scores_ptr=part.data_ptr(), ldo=n, base=7)
        torch.cuda.synchronize()
        got += list(map(tuple, rec[:int(cnt.item())].cpu().numpy().tolist()))
        w = ctx.mu_gapless_last_work()
        pairs, cells = pairs + w[0], cells + w[1]
        P = part.cpu().numpy()
        wrote = P != -1
        both = wrote & (U != -1)                                      # two members of one ring: scored from either side, the same value
        assert np.array_equal(P[both], U[both])
        U[wrote] = P[wrote]
        if lo == hi:
            assert not wrote.any() and w[0] == 0
    assert sorted(got) == want and len(set((a, b) for a, b, _ in got)) == len(got)
    assert np.array_equal(U[ii, jj], F[ii, jj])
    assert (pairs, cells) == (work_full[0], work_full[1]) and pairs == n * (n + 1) // 2
    db.close()
    ctx.close()

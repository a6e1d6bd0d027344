"""GPU parity: Mu-letter affine SW score + the Mu filter through the C-ABI
vs the reference's own outputs (tests/golden) and the CPU oracle."""
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


def run_sw_matrix(ctx, seqs_q, seqs_t=None, tri=False, reverse=False):
    import torch
    import reseek_amd
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs_q)
    t = q if seqs_t is None else reseek_amd.Db.from_mu_seqs(ctx, seqs_t)
    nq, nt = len(seqs_q), (len(seqs_q) if seqs_t is None else len(seqs_t))
    out = torch.full((nq, nt), 77, dtype=torch.uint8, device="cuda")
    ctx.mu_sw_matrix_dev(q, t, tri, reverse, out.data_ptr(), nt)
    torch.cuda.synchronize()
    res = out.cpu().numpy().astype(np.int32)
    q.close()
    if t is not q:
        t.close()
    return res


def run_filter(ctx, seqs, omega, omega_fwd, tri=True, seqs_t=None):
    import torch
    import reseek_amd
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    t = q if seqs_t is None else reseek_amd.Db.from_mu_seqs(ctx, seqs_t)
    nq, nt = q.n, t.n
    fwd = torch.zeros((nq, nt), dtype=torch.uint8, device="cuda")
    cap = nq * nt
    pq = torch.zeros(cap, dtype=torch.int32, device="cuda")
    pt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    pf = torch.zeros(cap, dtype=torch.int32, device="cuda")
    pr = torch.zeros(cap, dtype=torch.int32, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.mu_filter_dev(q, t, tri, omega, omega_fwd, fwd.data_ptr(), nt, pq.data_ptr(), pt.data_ptr(), pf.data_ptr(),
                      pr.data_ptr(), cap, n.data_ptr())
    torch.cuda.synchronize()
    k = int(n.item())
    res = {(int(a), int(b)): (int(f), int(r)) for a, b, f, r in zip(pq[:k].cpu(), pt[:k].cpu(), pf[:k].cpu(), pr[:k].cpu())}
    assert len(res) == k
    work = ctx.mu_filter_last_work()
    q.close()
    if t is not q:
        t.close()
    return res, fwd.cpu().numpy().astype(np.int32), work


def test_q100_raw_scores_match_reference(ctx):
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    _, recs = fx.read_pairs("pairs_q100_sensitive.bin.gz")
    seqs = [c.mu for c in chains]
    fwd = run_sw_matrix(ctx, seqs, tri=True)
    rev = run_sw_matrix(ctx, seqs, tri=True, reverse=True)
    nsat = 0
    for r in recs:
        assert fwd[r["i"], r["j"]] == r["para_fwd"], (r["i"], r["j"], r["LA"], r["LB"])
        assert rev[r["i"], r["j"]] == r["para_rev"]
        nsat += r["para_fwd_sat"]
    assert nsat >= 100


def test_q100_filter_matches_reference_sensitive_and_fast(ctx):
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    _, recs = fx.read_pairs("pairs_q100_sensitive.bin.gz")
    seqs = [c.mu for c in chains]
    # the fixture's mufilter value was produced with OmegaFwd = 20 (Sensitive); pass iff >= Omega = 12
    res, fwd, work = run_filter(ctx, seqs, 12.0, 20.0)
    want = {(r["i"], r["j"]) for r in recs if r["mufilter"] >= 12.0}
    assert set(res) == want
    for r in recs:
        key = (r["i"], r["j"])
        if key in res:
            f, rv = res[key]
            assert f == (777 if r["para_fwd_sat"] else r["para_fwd"])
            assert rv == r["para_rev"]             # 255 when saturated
            assert float(f - rv) == r["mufilter"]
    assert work[0] == 5050 and work[1] == sum(1 for r in recs if (777 if r["para_fwd_sat"] else r["para_fwd"]) >= 20)
    # Fast preset (22 / 50) from the raw scores
    res2, _, _ = run_filter(ctx, seqs, 22.0, 50.0)
    want2 = set()
    for r in recs:
        f = 777 if r["para_fwd_sat"] else r["para_fwd"]
        if f >= 50 and f - r["para_rev"] >= 22:
            want2.add((r["i"], r["j"]))
    assert set(res2) == want2


def test_scop40_real_sequences_full_matrix(ctx):
    seqs, tab = fx.read_mukat("mukat_scop40_160.bin.gz")
    got = run_sw_matrix(ctx, seqs, seqs)
    assert np.array_equal(got, tab[:, :, 0])
    assert (tab[:, :, 1] == (tab[:, :, 0] == 255)).all()


def test_random_and_adversarial_pairs_vs_reference_kat(ctx):
    kat = fx.read_randkat("randkat_3000.bin.gz")[:600]
    qs = [k[0] for k in kat]
    ts = [k[1] for k in kat]
    got = run_sw_matrix(ctx, qs, ts)
    for i, k in enumerate(kat):
        assert got[i, i] == k[2], (i, len(k[0]), len(k[1]))


def test_lengths_edge_cases_vs_oracle(ctx):
    rng = np.random.default_rng(3)
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 224, 225, 416, 417, 700, 1056, 1057, 1500, 2048, 2049, 2300]
    seqs = [rng.integers(0, 36, L).astype(np.uint8) for L in lens]
    seqs += [rng.integers(0, 4, L).astype(np.uint8) for L in (40, 300, 1200)]     # low complexity -> saturation
    got = run_sw_matrix(ctx, seqs, tri=True)
    rev = run_sw_matrix(ctx, seqs, tri=True, reverse=True)
    n = len(seqs)
    for i in range(n):
        for j in range(i, n):
            assert got[i, j] == ol.mu_sw(seqs[i], seqs[j])[0], (lens[i] if i < len(lens) else -1, j)
            assert rev[i, j] == ol.mu_sw(seqs[i][::-1].copy(), seqs[j])[0]


def test_rectangular_and_unsorted_targets(ctx):
    rng = np.random.default_rng(9)
    qs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(10, 500, 23)]
    ts = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(3, 900, 157)]
    got = run_sw_matrix(ctx, qs, ts)
    f, _, _ = ol.mu_filter_pairs(qs + ts, np.repeat(np.arange(23), 157), np.tile(np.arange(157), 23) + 23, 1e9)
    f = np.where(f == 777, 255, f).reshape(23, 157)
    assert np.array_equal(got, f)
    res, fwd, _ = run_filter(ctx, qs, 3.0, 10.0, tri=False, seqs_t=ts)
    ff, rr, ss = ol.mu_filter_pairs(qs + ts, np.repeat(np.arange(23), 157), np.tile(np.arange(157), 23) + 23, 10.0)
    want = {(int(a), int(b)) for a, b, f1, s1 in zip(np.repeat(np.arange(23), 157), np.tile(np.arange(157), 23), ff, ss)
            if f1 >= 10 and s1 >= 3.0}
    assert set(res) == want and len(want) > 5


def test_pair_list_filter_matches_oracle(ctx):
    """rsk_mu_filter_pairs (AlignMuParaBags per prefilter candidate, chainbag.cpp:68-74): arbitrary pair
    lists with repeats, both presets, vs the per-pair oracle and the reference's q100 filter values."""
    import reseek_amd
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    seqs = [c.mu for c in chains]
    rng = np.random.default_rng(11)
    n = 3000
    iq = rng.integers(0, len(seqs), n).astype(np.uint32)
    it = rng.integers(0, len(seqs), n).astype(np.uint32)
    iq[:50] = it[:50]                       # self pairs (saturate)
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    for omega, omega_fwd in ((12.0, 20.0), (22.0, 50.0)):
        ok, fwd, rev = ctx.mu_filter_pairs(q, q, iq, it, omega, omega_fwd)
        of, orv, osc = ol.mu_filter_pairs(seqs, iq, it, omega_fwd)
        want_fwd = of                       # the oracle reports 777 for saturated forward scores
        assert np.array_equal(fwd, want_fwd) and (fwd == 777).sum() >= 50
        cand = ~(want_fwd.astype(np.float32) < omega_fwd)
        assert np.array_equal(rev[cand], orv[cand]) and not rev[~cand].any()
        assert np.array_equal(ok.astype(bool), ~(osc < omega))
        assert ok.any() and not ok.all()
    q.close()
    # empty list
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs[:3])
    ok, fwd, rev = ctx.mu_filter_pairs(q, q, np.zeros(0, np.uint32), np.zeros(0, np.uint32), 12.0, 20.0)
    assert len(ok) == 0
    q.close()


def test_pairs_sort_dev_orders_the_survivor_list(ctx):
    """rsk_pairs_sort_dev: the filter appends survivors in no particular order; the host layer wants them by (A-side chain,
    B-side chain), the order the reference walks its pairs in (runself.cpp:72-99).  Duplicates, a single pair, an empty
    list, indices that need 17 and 32 bits."""
    import torch
    rng = np.random.default_rng(17)
    for n, bound in ((0, 0), (1, 5), (1000, 7), (300_000, 70_000), (1_500_000, 0)):
        hi = bound if bound else 2 ** 32 - 1
        a = rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 2 ** 32 - 1, n, dtype=np.uint64).astype(np.uint32)
        if n > 10:
            a[5:10] = a[4]                     # runs of one major index
            b[7] = b[6]                        # a duplicate pair
        da = torch.from_numpy(a.view(np.int32)).cuda()
        db = torch.from_numpy(b.view(np.int32)).cuda()
        ctx.pairs_sort_dev(da.data_ptr(), db.data_ptr(), n, bound)
        torch.cuda.synchronize()
        ga, gb = da.cpu().numpy().view(np.uint32), db.cpu().numpy().view(np.uint32)
        order = np.lexsort((b, a))
        assert np.array_equal(ga, a[order]) and np.array_equal(gb, b[order]), (n, bound)


def test_two_queries_per_register_equals_one(ctx, monkeypatch):
    """k_mu_sw2 (dense forward pass: rows of two queries of neighbouring length share a packed register) against k_mu_sw
    (RSK_MUSW_QUERY_PAIRS=0) and the oracle: odd query counts (a lone last query), lengths across every class boundary of
    the pair kernel (208 / 512 / 1024 residues) and beyond it (k_mu_sw takes those), self triangle, rectangle with unsorted
    targets, reversed queries, saturated and empty-ish chains."""
    rng = np.random.default_rng(31)
    lens = [1, 2, 15, 16, 17, 100, 207, 208, 209, 300, 511, 512, 513, 800, 1023, 1024, 1025, 1400, 2048]
    seqs = [rng.integers(0, 36, L).astype(np.uint8) for L in lens]
    seqs += [rng.integers(0, 3, L).astype(np.uint8) for L in (33, 260, 900)]          # low complexity: saturation
    seqs = [seqs[i] for i in rng.permutation(len(seqs))]
    assert len(seqs) % 2 == 0
    seqs.append(rng.integers(0, 36, 77).astype(np.uint8))                             # odd count
    ts = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(3, 700, 41)]
    two = (run_sw_matrix(ctx, seqs, tri=True), run_sw_matrix(ctx, seqs, tri=True, reverse=True), run_sw_matrix(ctx, seqs, ts),
           run_sw_matrix(ctx, seqs, ts, reverse=True))
    monkeypatch.setenv("RSK_MUSW_QUERY_PAIRS", "0")
    one = (run_sw_matrix(ctx, seqs, tri=True), run_sw_matrix(ctx, seqs, tri=True, reverse=True), run_sw_matrix(ctx, seqs, ts),
           run_sw_matrix(ctx, seqs, ts, reverse=True))
    monkeypatch.delenv("RSK_MUSW_QUERY_PAIRS")
    n = len(seqs)
    iu = np.triu_indices(n)
    for k in range(2):
        assert np.array_equal(two[k][iu], one[k][iu])
    for k in (2, 3):
        assert np.array_equal(two[k], one[k])
    for i in range(n):
        for j in range(0, len(ts), 5):
            assert two[2][i, j] == ol.mu_sw(seqs[i], ts[j])[0]
        for j in range(i, n, 3):
            assert two[0][i, j] == ol.mu_sw(seqs[i], seqs[j])[0]


def test_filter_windows_tile_the_triangle(ctx):
    """rsk_mu_filter_window_dev (r06: one shard of the self-search triangle = a window of the set's length order): the survivors of
    the windows of N = 1, 2, 3, 8 shards (rsk_shard_range kind 2) are disjoint and their union is the whole triangle's survivor
    set with the same (fwd, rev); every survivor's longer member (rsk_len_rank: the later one of equal lengths) stands in the
    window that reported it; ragged windows (empty, one position, beyond a 1,100-residue chain of the slow class)."""
    import torch
    import reseek_amd
    from reseek_amd import capi
    rng = np.random.default_rng(33)
    lens = np.concatenate([rng.integers(5, 400, 180), [1100, 1500, 7, 7, 7, 1024, 1025]])
    rng.shuffle(lens)
    base = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in lens]
    for k in range(0, 60, 2):                               # planted look-alikes so that pairs survive
        src = base[k]
        L = len(base[k + 1])
        base[k + 1] = np.resize(src, L).copy()
        base[k + 1][rng.integers(0, L, max(1, L // 5))] = rng.integers(0, 36, max(1, L // 5))
    want, _, _ = run_filter(ctx, base, 12.0, 20.0, tri=True)
    assert len(want) > 40
    db = reseek_amd.Db.from_mu_seqs(ctx, base)
    n = db.n
    rank = ctx.len_rank(db)
    assert sorted(rank) == list(range(n)) and all(lens[a] < lens[b] or (lens[a] == lens[b] and a < b) for a, b in zip(np.argsort(rank)[:-1], np.argsort(rank)[1:]))
    cap = n * n
    fwd = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
    pq, pt, pf, pr = (torch.zeros(cap, dtype=torch.int32, device="cuda") for _ in range(4))
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")

    def window(lo, hi):
        ctx.mu_filter_window_dev(db, lo, hi, 12.0, 20.0, fwd.data_ptr(), n, pq.data_ptr(), pt.data_ptr(), pf.data_ptr(), pr.data_ptr(), cap, cnt.data_ptr())
        torch.cuda.synchronize()
        k = int(cnt.item())
        res = {(int(a), int(b)): (int(f), int(r)) for a, b, f, r in zip(pq[:k].cpu(), pt[:k].cpu(), pf[:k].cpu(), pr[:k].cpu())}
        assert len(res) == k
        for a, b in res:
            assert a <= b and lo <= max(rank[a], rank[b]) < hi, (a, b, lo, hi)
        assert ctx.mu_filter_last_work()[0] == (hi * (hi + 1) - lo * (lo + 1)) // 2
        return res

    for count in (1, 2, 3, 8):
        got = {}
        for r in range(count):
            lo, hi = capi.shard_range(2, lens.astype(np.uint32), r, count)
            w = window(lo, hi)
            assert not (set(w) & set(got))
            got.update(w)
        assert got == want, count
    assert window(5, 5) == {} and window(n, n) == {}
    one = window(n - 1, n)                                  # the longest chain against everything
    assert one == {k: v for k, v in want.items() if max(rank[k[0]], rank[k[1]]) == n - 1}
    with pytest.raises(capi.RskError):
        window(3, n + 1)
    db.close()

"""GPU parity: Mu k-mer prefilter (exact k-mers) through the C-ABI vs
`reseek -prefilter_mu` outputs of the reference binary and the CPU oracle."""
import gzip
import hashlib
import os
import tempfile

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


def run_prefilter(ctx, qseqs, tseqs=None, cap=20_000_000, mode=0):
    import torch
    import reseek_amd
    q = reseek_amd.Db.from_mu_seqs(ctx, qseqs)
    t = q if tseqs is None else reseek_amd.Db.from_mu_seqs(ctx, tseqs)
    dq = torch.zeros(cap, dtype=torch.int32, device="cuda")
    dt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    ds = torch.zeros(cap, dtype=torch.int32, device="cuda")
    dn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.mu_prefilter_dev(q, t, dq.data_ptr(), dt.data_ptr(), ds.data_ptr(), cap, dn.data_ptr(), neighbourhood=mode)
    torch.cuda.synchronize()
    n = int(dn.item())
    assert n <= cap
    ms = ctx.last_kernel_ms()
    q.close()
    if t is not q:
        t.close()
    return dq[:n].cpu().numpy().astype(np.uint32), dt[:n].cpu().numpy().astype(np.uint32), ds[:n].cpu().numpy().astype(np.uint32), ms


def as_set(q, t, s):
    return set(zip(q.tolist(), t.tolist(), s.tolist()))


def scores_text(labels, q, t, s):
    lines = ["%s\t%s\t%d" % (labels[a], labels[b], c) for a, b, c in zip(q.tolist(), t.tolist(), s.tolist())]
    lines.sort()
    return "\n".join(lines) + "\n"


def test_sub1000_triples_and_bags_match_reference(ctx):
    import reseek_amd
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz", limit=1000)
    q, t, s, _ = run_prefilter(ctx, seqs)
    oq, ot, os_ = ol.prefilter(seqs, seqs)
    assert as_set(q, t, s) == as_set(oq, ot, os_)
    for B, tag in ((1500, ""), (50, "_b50")):
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "tmp.tsv")
            rq, rt, rs = reseek_amd.capi.rsb_select(q, t, s, len(seqs), B, tmp_tsv_path=tmp)
            want = gzip.open(os.path.join(fx.GOLDEN, "prefilter_sub1000%s_scores.tsv.gz" % tag)).read().decode()
            assert scores_text(labels, rq, rt, rs) == want
            want_tmp = gzip.open(os.path.join(fx.GOLDEN, "prefilter_sub1000%s_tmp.tsv.gz" % tag)).read().decode()
            assert open(tmp).read() == want_tmp


def test_scop40_full_checksums(ctx):
    """BASELINE configs[2] shape: all 11,211 SCOP40 Mu sequences against themselves (125.7 M ordered pairs)."""
    import reseek_amd
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz")
    q, t, s, ms = run_prefilter(ctx, seqs)
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "tmp.tsv")
        rq, rt, rs = reseek_amd.capi.rsb_select(q, t, s, len(seqs), 1500, tmp_tsv_path=tmp)
        want = dict(zip(*[iter(open(os.path.join(fx.GOLDEN, "prefilter_scop40_full.md5.txt")).read().split())] * 2))
        assert len(rq) == int(want["lines"])
        assert hashlib.md5(scores_text(labels, rq, rt, rs).encode()).hexdigest() == want["sorted_scores_md5"]
        assert hashlib.md5(open(tmp, "rb").read()).hexdigest() == want["tmp_tsv_md5"]
    print("prefilter kernel ms", ms)


def test_rectangular_edge_cases_vs_oracle(ctx):
    rng = np.random.default_rng(4)
    base = rng.integers(0, 36, 600).astype(np.uint8)
    qs = [base[:L].copy() for L in (6, 7, 8, 50, 300, 600)] + [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(7, 400, 40)]
    ts = [base[100:500].copy(), base[::-1].copy(), np.full(300, 17, np.uint8), base[:6].copy(), base[:7].copy()]
    ts += [np.concatenate([base[50:250], rng.integers(0, 36, 100).astype(np.uint8), base[50:250]])]      # repeats -> many two-hit diagonals
    ts += [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(7, 700, 30)]
    q, t, s, _ = run_prefilter(ctx, qs, ts)
    oq, ot, os_ = ol.prefilter(qs, ts)
    assert as_set(q, t, s) == as_set(oq, ot, os_) and len(oq) >= 4


def test_many_hits_forces_lds_chunking(ctx):
    """Low-complexity chains: tens of thousands of postings hits per target -> several query-range chunks."""
    rng = np.random.default_rng(6)
    motif = rng.integers(0, 36, 40).astype(np.uint8)
    qs = [np.tile(motif, 6)[: int(L)] for L in rng.integers(100, 240, 300)]
    ts = [np.tile(motif, 8)[: int(L)] for L in rng.integers(150, 320, 6)]
    q, t, s, _ = run_prefilter(ctx, qs, ts)
    oq, ot, os_ = ol.prefilter(qs, ts)
    assert as_set(q, t, s) == as_set(oq, ot, os_) and len(oq) == 300 * 6


def test_long_targets_and_overflowing_buckets_vs_oracle(ctx):
    """Paths of k_prefilter that ordinary chains do not reach: targets longer than the LDS letter staging (read in place),
    a 64-query bucket whose keys exceed the hash set (cut into query runs), and a single query that exceeds it alone
    (bitmap path); exact and neighbourhood indexes."""
    rng = np.random.default_rng(8)
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz")
    long_t = np.concatenate([seqs[i] for i in range(0, 120)])[:16000]                   # 16,000 letters > staging
    motif = rng.integers(0, 36, 30).astype(np.uint8)
    lowc_t = np.tile(motif, 400)[:9000]                                                  # low complexity, long
    qs = [seqs[i] for i in range(0, 40)]                                                 # real chains (parts of long_t)
    qs += [np.tile(motif, 12)[: int(L)] for L in rng.integers(200, 360, 70)]             # same bucket as each other: overflow
    qs += [np.tile(motif, 60)[:1700]]                                                    # one query with > 4096 keys vs lowc_t
    ts = [long_t, lowc_t, seqs[500], np.tile(motif, 10)[:250]]
    for mode in (0, 2):
        q, t, s, _ = run_prefilter(ctx, qs, ts, mode=mode)
        oq, ot, os_ = ol.prefilter(qs, ts, mode=mode)
        assert as_set(q, t, s) == as_set(oq, ot, os_) and len(oq) > 100, mode


def test_neighbourhood_modes_match_muprefilter(ctx):
    """`-search -fast -db` prefilter (MuPreFilter muprefilter.cpp:70): idxq and idxt neighbourhood modes,
    80 queries x 1000 targets, against the reference's score list and hand-off TSV."""
    import reseek_amd
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz", limit=1000)
    qs = seqs[:80]
    for mode, tag in ((1, "h80"), (2, "h80t")):
        q, t, s, _ = run_prefilter(ctx, qs, seqs, mode=mode)
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "tmp.tsv")
            rq, rt, rs = reseek_amd.capi.rsb_select(q, t, s, len(qs), 1500, tmp_tsv_path=tmp)
            want = gzip.open(os.path.join(fx.GOLDEN, "prefilter_hood_%s_scores.tsv.gz" % tag)).read().decode()
            assert scores_text(labels, rq, rt, rs) == want
            want_tmp = gzip.open(os.path.join(fx.GOLDEN, "prefilter_hood_%s_tmp.tsv.gz" % tag)).read().decode()
            assert open(tmp).read() == want_tmp
    # the reference's automatic choice: <= 100 queries -> idxq
    q2, t2, s2, _ = run_prefilter(ctx, qs, seqs[:200], mode=-1)
    oq, ot, os_ = ol.prefilter(qs, seqs[:200], mode=1)
    assert as_set(q2, t2, s2) == as_set(oq, ot, os_)


def test_neighbourhood_self_1000(ctx):
    """1000 x 1000 (> 100 queries -> idxt): device vs the reference's scores."""
    import reseek_amd
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz", limit=1000)
    q, t, s, ms = run_prefilter(ctx, seqs, mode=-1)
    rq, rt, rs = reseek_amd.capi.rsb_select(q, t, s, len(seqs), 1500)
    want = gzip.open(os.path.join(fx.GOLDEN, "prefilter_hood_h1000_scores.tsv.gz")).read().decode()
    assert scores_text(labels, rq, rt, rs) == want


def test_scop40_full_neighbourhood_checksums(ctx):
    """All 11,211 SCOP40 Mu sequences against themselves WITH k-mer neighbourhoods (> 100 queries -> idxt), as
    MuPreFilter runs inside `-search -fast -db`: 2.9e9 seed items, 65 M (query, target) two-hit pairs, top-1500 per
    query -> md5 of the reference's score list and hand-off file (ref_harness prefhood, ~15 min on one CPU thread)."""
    import reseek_amd
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz")
    q, t, s, ms = run_prefilter(ctx, seqs, cap=80_000_000, mode=-1)
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "tmp.tsv")
        rq, rt, rs = reseek_amd.capi.rsb_select(q, t, s, len(seqs), 1500, tmp_tsv_path=tmp)
        want = dict(zip(*[iter(open(os.path.join(fx.GOLDEN, "prefilter_hood_scop40_full.md5.txt")).read().split())] * 2))
        assert len(rq) == int(want["lines"])
        assert hashlib.md5(scores_text(labels, rq, rt, rs).encode()).hexdigest() == want["sorted_scores_md5"]
        assert hashlib.md5(open(tmp, "rb").read()).hexdigest() == want["tmp_tsv_md5"]


def test_device_sorted_keys_and_key_replay(ctx, tmp_path):
    """rsk_triples_sort_dev: the kernel's unordered triples -> query << 48 | target << 16 | score, ascending (numpy is the
    check); rsk_rsb_select_keys on them == rsk_rsb_select on the triples (bags and hand-off file), with a bag size that
    overflows.  Also a dense low-complexity set (every pair has many seeds: the bitmap spans and the >4096-diagonal rounds of
    the scan) against the oracle."""
    import torch
    import reseek_amd
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz", limit=600)
    q = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    cap = 600 * 600
    dq, dt, ds = (torch.zeros(cap, dtype=torch.int32, device="cuda") for _ in range(3))
    dn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.mu_prefilter_dev(q, q, dq.data_ptr(), dt.data_ptr(), ds.data_ptr(), cap, dn.data_ptr(), neighbourhood=2)
    torch.cuda.synchronize()
    n = int(dn.item())
    dk = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
    ctx.triples_sort_dev(dq.data_ptr(), dt.data_ptr(), ds.data_ptr(), n, dk.data_ptr())
    hq, ht, hs = (x[:n].cpu().numpy().astype(np.uint64) for x in (dq, dt, ds))
    want = np.sort((hq << np.uint64(48)) | (ht << np.uint64(16)) | hs)
    got = dk[:n].cpu().numpy().view(np.uint64)
    assert n > 1000 and np.array_equal(got, want)
    fa, fb = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    a = reseek_amd.capi.rsb_select(hq, ht, hs, len(seqs), 40, tmp_tsv_path=fa)
    b = reseek_amd.capi.rsb_select_keys(got, len(seqs), 40, tmp_tsv_path=fb)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and open(fa, "rb").read() == open(fb, "rb").read()
    q.close()
    # dense: 40 low-complexity chains (three letters in long runs), neighbourhood index
    rng = np.random.default_rng(3)
    dense = []
    for L in rng.integers(60, 700, 40):
        runs = rng.integers(3, 40, int(L))
        lets = rng.choice(np.array([17, 34, 35, 33, 14], np.uint8), int(L))
        dense.append(np.repeat(lets, runs)[: int(L)].astype(np.uint8))
    for mode in (0, 2):
        gq, gt, gs, _ = run_prefilter(ctx, dense, cap=40 * 40 + 16, mode=mode)
        oq, ot, os_ = ol.prefilter(dense, dense, mode=mode)
        assert as_set(gq, gt, gs) == as_set(np.asarray(oq), np.asarray(ot), np.asarray(os_)), "dense set, mode %d" % mode


def test_very_long_chains_and_the_diagonal_cut(ctx):
    """Chains beyond 8,192 residues: QL + TL - 1 exceeds the 16,384 diagonals a query's bitmap holds, seeds on diagonals
    > 16383 are dropped (prefiltermu.cpp:254) and the diagonal index wraps at 16 bits before that test (a reference quirk the
    oracle restates).  Both roles, exact k-mers and the neighbourhood index, against the oracle."""
    rng = np.random.default_rng(11)

    def lowc(L):
        runs = rng.integers(2, 30, L)
        lets = rng.choice(np.array([17, 34, 35, 33, 14, 3], np.uint8), L)
        return np.repeat(lets, runs)[:L].astype(np.uint8)

    seqs = [lowc(17000), lowc(300), lowc(9000), lowc(40), lowc(700), lowc(66000)[:65000]]
    for mode in (0, 2):
        gq, gt, gs, _ = run_prefilter(ctx, seqs, cap=64, mode=mode)
        oq, ot, os_ = ol.prefilter(seqs, seqs, mode=mode)
        assert as_set(gq, gt, gs) == as_set(np.asarray(oq), np.asarray(ot), np.asarray(os_)), "mode %d" % mode

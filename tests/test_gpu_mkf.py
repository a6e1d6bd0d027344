"""GPU parity: seeding stage of the long-chain MKF path (rsk_mkf_seed_pairs) vs the HSP lists
MuKmerFilter::Align of the reference kept (tests/golden/mkfkat_*, oracle/ref_harness mkfkat) and the oracle."""
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


def test_palms_all_pairs_match_reference(ctx):
    import reseek_amd
    chains = fx.read_rskdb("palms_sensitive.rskdb.gz")
    n, kat = fx.read_mkfkat("mkfkat_palms_sensitive.bin.gz")
    db = reseek_amd.Db.from_mu_seqs(ctx, [c.mu for c in chains])
    iq, it = np.divmod(np.arange(n * n, dtype=np.uint32), n)
    found, recs = ctx.mkf_seed_pairs(db, db, iq, it, cap=32)
    nfound = 0
    for p in range(n * n):
        kept, _, _ = kat[(int(iq[p]), int(it[p]))]
        assert bool(found[p]) == (len(kept) > 0)
        if found[p]:
            nk, got = recs[p]
            assert nk == len(kept) and np.array_equal(got, kept)
            nfound += 1
        else:
            assert p not in recs
    assert nfound > 100
    # a table budget smaller than the number of distinct queries (one 373 KB 3-mer table each): the call runs in chunks
    # that reuse one table block -- interleaved query order so that queries recur across chunks
    import os
    perm = np.random.default_rng(5).permutation(n * n)
    os.environ["RSK_MKF_MAX_TABLES"] = "3"
    try:
        found3, recs3 = ctx.mkf_seed_pairs(db, db, iq[perm], it[perm], cap=32)
    finally:
        del os.environ["RSK_MKF_MAX_TABLES"]
    assert np.array_equal(found3, found[perm])
    for k, p in enumerate(perm):
        if found[p]:
            assert recs3[k][0] == recs[p][0] and np.array_equal(recs3[k][1], recs[p][1])
    # truncation: cap 1 keeps the first HSP; the count is exact when it fits and an upper bound (> cap) otherwise
    found1, recs1 = ctx.mkf_seed_pairs(db, db, iq, it, cap=1)
    assert np.array_equal(found1, found)
    for p, (nk, got) in recs1.items():
        assert np.array_equal(got, recs[p][1][:1])
        assert nk == 1 if recs[p][0] == 1 else nk >= recs[p][0] > 1
    db.close()


def test_other_thresholds_and_short_chains_vs_oracle(ctx):
    import reseek_amd
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")
    seqs = [c.mu for c in chains[:40]] + [np.array([1, 2], np.uint8), np.array([3, 3, 3], np.uint8), np.zeros(300, np.uint8)]
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    n = len(seqs)
    iq, it = np.divmod(np.arange(n * n, dtype=np.uint32), n)
    for x1, ms in ((8, 50), (4, 20), (99999, 0)):
        found, recs = ctx.mkf_seed_pairs(db, db, iq, it, x1=x1, min_hsp_score=ms, cap=32)
        for p in range(n * n):
            f, nk, kept = ol.mkf_seed(seqs[iq[p]], seqs[it[p]], x1, ms, 32)
            assert bool(found[p]) == f
            if f:
                assert recs[p][0] == nk and np.array_equal(recs[p][1], kept)
    db.close()


def test_device_chaining_equals_the_reference_chains(ctx):
    """rsk_mkf_chain_align_pairs (k_mkf_chain = Chainer::Chain on the device, then the long-chain batch) against the same batch
    fed the chains the REFERENCE built from the same seed HSPs (mkfkat fixture: kept HSPs + chain of every palms pair,
    MuKmerFilter::Align + ChainHSPs of the reference): every pair that is not flagged as qsort-dependent must come out
    identical -- gates, score bits, path, E-value bits."""
    import struct
    import reseek_amd
    chains = fx.read_rskdb("palms_sensitive.rskdb.gz")
    n, kat = fx.read_mkfkat("mkfkat_palms_sensitive.bin.gz")
    db = reseek_amd.Db.from_chains(ctx, chains)
    pairs = [(i, j) for i in range(n) for j in range(n) if len(kat[(i, j)][0])]
    assert len(pairs) > 100
    ia, ib = [p[0] for p in pairs], [p[1] for p in pairs]
    kept = [kat[p][0] for p in pairs]
    first_u = np.concatenate([[0], np.cumsum([len(k) for k in kept])]).astype(np.uint32)
    ku = np.concatenate(kept).astype(np.int32)
    out_dev, st_dev = ctx.mkf_align_pairs(db, db, ia, ib, first_u, ku[:, 0], ku[:, 1], ku[:, 2], hsp_score=ku[:, 3])
    # reference chains; a pair whose best chain score is <= 0 has no alignment (dssaligner.cpp:1397): empty list
    ch = [kat[p][2] if kat[p][1] > 0 else kat[p][2][:0] for p in pairs]
    first_c = np.concatenate([[0], np.cumsum([len(c) for c in ch])]).astype(np.uint32)
    kc = np.concatenate([c for c in ch if len(c)] or [np.zeros((0, 3), np.int32)]).astype(np.int32)
    out_ref, st_ref = ctx.mkf_align_pairs(db, db, ia, ib, first_c, kc[:, 0], kc[:, 1], kc[:, 2])
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    ntie = naln = 0
    for k, p in enumerate(pairs):
        if st_dev[k] == 3:
            ntie += 1
            continue
        (a, pa), (b, pb) = out_dev[k], out_ref[k]
        assert st_dev[k] == st_ref[k], p
        assert pa == pb and bits(a.score) == bits(b.score), p
        if pa:
            assert (a.lo_a, a.lo_b) == (b.lo_a, b.lo_b) and bits(a.evalue) == bits(b.evalue) and bits(a.lddt) == bits(b.lddt), p
            naln += 1
    assert naln > 100 and ntie < len(pairs) // 10
    db.close()


def test_chains_that_depend_on_qsort_order_are_flagged(ctx):
    """Two seed HSPs ending at one query position with equal chain scores: the reference keeps whichever libc qsort leaves in
    front (chainer.cpp:11-29, 121-124), so the device reports status 3 and aligns nothing for the pair; lists without such a
    tie (also with equal scores at different ends, or equal ends with different scores) are chained."""
    import reseek_amd
    chains = fx.read_rskdb("palms_sensitive.rskdb.gz")[:2]
    db = reseek_amd.Db.from_chains(ctx, chains)

    def status(hsps):
        h = np.array(hsps, np.int32)
        _, st = ctx.mkf_align_pairs(db, db, [0], [1], np.array([0, len(h)], np.uint32), h[:, 0], h[:, 1], h[:, 2], hsp_score=h[:, 3])
        return int(st[0])

    assert status([(10, 10, 31, 60), (20, 25, 21, 60)]) == 3            # both end at query position 40, scores equal
    assert status([(10, 10, 31, 60), (20, 25, 21, 61)]) != 3            # same end, different scores
    assert status([(10, 10, 31, 60), (20, 25, 22, 60)]) != 3            # equal scores, different ends
    assert status([(10, 10, 20, 50), (40, 42, 11, 30), (25, 30, 26, 80)]) == 3   # [10,29]+[40,50] = 50 + 30 ties with [25,50] = 80 (overlaps the first) at end 50
    db.close()

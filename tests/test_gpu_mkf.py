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

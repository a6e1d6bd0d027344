"""GPU parity for the remaining dead-but-named reference kernels of SURVEY 8a row D1
(SWFastPinop, SWFastGaplessProfb, SWFastGapless on the float S matrix), pair-list C-ABI entry points,
vs values produced by the reference functions themselves (tests/golden, oracle/ref_harness) and the oracle."""
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


def make_db(ctx, chains):
    import reseek_amd
    lens = np.array([c.L for c in chains], np.uint32)
    mu = np.concatenate([c.mu for c in chains])
    prof = np.concatenate([c.prof.reshape(-1) for c in chains])
    return reseek_amd.Db(ctx, lens, mu=mu, prof=prof)


def test_profb_and_float_gapless_match_reference(ctx):
    chains = fx.read_rskdb("q100_sensitive.rskdb.gz")[:40]
    n, pb, g, bij = fx.read_d1pairs("d1pairs_q40_sensitive.bin.gz")
    db = make_db(ctx, chains)
    ia, ib = np.divmod(np.arange(n * n, dtype=np.uint32), n)
    got = ctx.mu_gapless_profb_pairs(db, db, ia, ib)
    assert got.tobytes() == pb.reshape(-1).astype(np.float32).tobytes()
    sc, bi, bj = ctx.gapless_float_pairs(db, db, ia, ib)
    assert sc.tobytes() == g.reshape(-1).astype(np.float32).tobytes()
    assert np.array_equal(bi, bij[..., 0].reshape(-1)) and np.array_equal(bj, bij[..., 1].reshape(-1))
    db.close()


def test_pinop_matches_reference_kats(ctx):
    import reseek_amd
    # real SCOP40 Mu sequences, all ordered pairs of the first 60 (reference SWFastPinop values)
    seqs, tab = fx.read_mukat("mukat_scop40_160.bin.gz")
    m = 60
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs[:m])
    ia, ib = np.divmod(np.arange(m * m, dtype=np.uint32), m)
    got = ctx.mu_pinop_pairs(db, db, ia, ib)
    assert np.array_equal(got, tab[:m, :m, 3].reshape(-1))
    db.close()
    # random / adversarial pairs
    kat = fx.read_randkat("randkat_3000.bin.gz")[:600]
    seqs2 = []
    for A, B, *_ in kat:
        seqs2 += [A if len(A) else np.zeros(1, np.uint8), B if len(B) else np.zeros(1, np.uint8)]
    keep = [k for k, (A, B, *_r) in enumerate(kat) if len(A) and len(B)]
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs2)
    ia = np.array([2 * k for k in keep], np.uint32)
    ib = ia + 1
    got = ctx.mu_pinop_pairs(db, db, ia, ib)
    assert np.array_equal(got, np.array([kat[k][5] for k in keep], np.int32))
    # other gap costs vs the oracle
    got = ctx.mu_pinop_pairs(db, db, ia[:100], ib[:100], open_=-5, ext=-2)
    want = [ol.mu_pinop(seqs2[a], seqs2[b], -5, -2) for a, b in zip(ia[:100], ib[:100])]
    assert got.tolist() == want
    db.close()


@pytest.mark.gpu
def test_db_create_rejects_bad_chain_sets(ctx):
    """rsk_db_create error behaviour: zero-length / over-long chains, Mu or profile letters outside their alphabets
    (reported for the first offending chain, the packing runs on several threads); nothing leaks into the next call."""
    import reseek_amd
    from reseek_amd import capi
    rng = np.random.default_rng(3)
    lengths = np.array([50, 70, 30, 90], np.uint32)
    tot = int(lengths.sum())
    mu = rng.integers(0, 36, tot).astype(np.uint8)
    prof = np.concatenate([np.concatenate([rng.integers(0, 20, (1, L)), rng.integers(0, 16, (7, L))]).astype(np.uint8).ravel() for L in lengths])
    with pytest.raises(capi.RskError, match="length"):
        reseek_amd.Db(ctx, np.array([50, 0, 30], np.uint32), mu=mu[:80])
    with pytest.raises(capi.RskError, match="length"):
        reseek_amd.Db(ctx, np.array([65535], np.uint32), mu=np.zeros(65535, np.uint8))
    bad = mu.copy(); bad[50 + 10] = 36; bad[50 + 70 + 5] = 99          # chains 1 and 2: chain 1 is reported
    with pytest.raises(capi.RskError, match="Mu letter 36 out of range in chain 1"):
        reseek_amd.Db(ctx, lengths, mu=bad)
    badp = prof.copy(); badp[8 * (50 + 70) + 3 * 30 + 7] = 16          # chain 2, feature 3
    with pytest.raises(capi.RskError, match="chain 2 feature 3"):
        reseek_amd.Db(ctx, lengths, mu=mu, prof=badp)
    badp = prof.copy(); badp[8 * 50 + 4] = 20                          # chain 1, feature 0 (amino acids: 20 letters)
    with pytest.raises(capi.RskError, match="chain 1 feature 0"):
        reseek_amd.Db(ctx, lengths, mu=mu, prof=badp)
    db = reseek_amd.Db(ctx, lengths, mu=mu, prof=prof)
    assert capi.lib().rsk_db_nchains(db.h) == 4

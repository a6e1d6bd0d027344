"""Host featurisation: the DSS mirror and the .bca reader against the per-chain bytes the
reference computed (tests/golden/*.rskdb.gz, dumped by oracle/ref_harness from the reference's own DSS).
The .bca files are the reference's own test data (test_data/q10.bca, q100.bca, palms.bca).  No GPU needed."""
import gzip
import os
import shutil
import tempfile

import numpy as np
import pytest

import fixtures as fx


@pytest.fixture(scope="module")
def bca_dir():
    d = tempfile.mkdtemp(prefix="rsk_bca_")
    for name in ("q10", "q100", "palms"):
        with gzip.open(os.path.join(fx.GOLDEN, name + ".bca.gz"), "rb") as f, open(os.path.join(d, name + ".bca"), "wb") as g:
            g.write(f.read())
    yield d
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("name,fixture", [("q10", "q10_sensitive.rskdb.gz"), ("q100", "q100_sensitive.rskdb.gz"),
                                          ("palms", "palms_sensitive.rskdb.gz")])
def test_bca_reader_and_dss_bytes(bca_dir, name, fixture):
    from reseek_amd import capi
    chains = fx.read_rskdb(fixture)
    path = os.path.join(bca_dir, name + ".bca")
    n, nres, maxlen, _ = capi.bca_info(path)
    assert n == len(chains) and nres == sum(c.L for c in chains) and maxlen == max(c.L for c in chains)
    bad = []
    for i, c in enumerate(chains):
        label, seq, x, y, z = capi.bca_read_chain(path, i)
        assert label == c.label and seq == c.seq
        assert x.tobytes() == c.x.tobytes() and y.tobytes() == c.y.tobytes() and z.tobytes() == c.z.tobytes()
        prof, mu = capi.dss_featurize(seq, x, y, z)
        if not (np.array_equal(prof, c.prof) and np.array_equal(mu, c.mu)):
            bad.append((i, [int((prof[f] != c.prof[f]).sum()) for f in range(8)], int((mu != c.mu).sum())))
    assert not bad, bad[:10]


def test_reversed_chain_profile_shortcut(bca_dir):
    """rsk_dss_featurize_reversed (exp() table mirrored from the un-reversed chain, used for the self-rev scores) gives
    the bytes of a full featurisation of the reversed arrays; fixtures + random walks incl. very short and > 2*window chains."""
    from reseek_amd import capi
    cases = []
    for name in ("q10", "q100", "palms"):
        path = os.path.join(bca_dir, name + ".bca")
        for i in range(capi.bca_info(path)[0]):
            cases.append(capi.bca_read_chain(path, i)[1:])
    rng = np.random.default_rng(17)
    for L in (1, 2, 3, 7, 13, 14, 27, 51, 52, 101, 102, 203, 640, 1500):
        seq = "".join("ACDEFGHIKLMNPQRSTVWY"[k] for k in rng.integers(0, 20, L))
        xyz = np.cumsum(rng.normal(0, 2.2, (3, L)), axis=1).astype(np.float32)
        cases.append((seq, xyz[0], xyz[1], xyz[2]))
    for seq, x, y, z in cases:
        want = capi.dss_featurize(seq[::-1], x[::-1], y[::-1], z[::-1])[0]
        got = capi.dss_featurize_reversed(seq, x, y, z)
        assert np.array_equal(got, want), (len(seq), [int((got[f] != want[f]).sum()) for f in range(8)])


def test_bca_errors(bca_dir):
    from reseek_amd import capi
    with pytest.raises(capi.RskError):
        capi.bca_info(os.path.join(bca_dir, "missing.bca"))
    junk = os.path.join(bca_dir, "junk.bca")
    open(junk, "wb").write(b"not a bca file at all........................")
    with pytest.raises(capi.RskError):
        capi.bca_info(junk)
    with pytest.raises(capi.RskError):
        capi.bca_read_chain(os.path.join(bca_dir, "q10.bca"), 10)


def test_bca_writer_and_mu_fasta_match_convert(bca_dir):
    """`reseek -convert`: the BCAData writer reproduces the reference's own files byte for byte, and the Mu FASTA
    equals `-convert q100.bca -feature_fasta` of the reference binary."""
    from reseek_amd import capi
    for name in ("q10", "q100", "palms"):
        src = os.path.join(bca_dir, name + ".bca")
        dst = os.path.join(bca_dir, name + "_copy.bca")
        capi.bca_copy(src, dst)
        assert open(dst, "rb").read() == open(src, "rb").read()
    fa = os.path.join(bca_dir, "q100.mu.fa")
    capi.bca_to_mu_fasta(os.path.join(bca_dir, "q100.bca"), fa)
    want = gzip.open(os.path.join(fx.GOLDEN, "q100_convert.mu.fa.gz")).read()
    assert open(fa, "rb").read() == want

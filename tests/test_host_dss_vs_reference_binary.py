"""Host featurisation against the reference itself on chains no fixture has seen: 1500 synthetic CA traces go through
oracle/_ref/ref_harness (DSS::GetProfile / GetMuLetters of the unmodified reference objects) and through
rsk_dss_featurize; profiles and Mu letters must be byte-identical.  (This comparison, at 3000 chains, found the one
residue in ~500,000 where a double-precision GetDist flips a nearest-neighbour tie.)  CPU only; skipped when the
harness was not built."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import fixtures as fx
from reseek_amd import capi

ROOT = os.path.dirname(os.path.dirname(fx.GOLDEN))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness was not built (no /root/reference at build time)")
def test_profiles_and_mu_letters_of_synthetic_chains():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import bench_search
    lens = bench.scop40_lengths()
    rng = np.random.default_rng(123)
    n = 1500
    with tempfile.TemporaryDirectory() as td:
        bca, ref = os.path.join(td, "syn.bca"), os.path.join(td, "ref.rskdb")
        bench_search.write_bca(bca, lens[rng.choice(len(lens), n)], rng)
        subprocess.run([HARNESS, "dbq", bca, ref, "--", "-fast"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        chains = fx.read_rskdb(ref)
        assert len(chains) == n
        nres = 0
        for i, c in enumerate(chains):
            _, seq, x, y, z = capi.bca_read_chain(bca, i)
            prof, mu = capi.dss_featurize(seq, x, y, z)
            assert np.array_equal(prof, c.prof) and np.array_equal(mu, c.mu), (i, len(seq))
            nres += len(seq)
        assert nres > 200000

"""End to end against the reference BINARY itself (oracle/_ref/reseek, built from /root/reference by oracle/Makefile.ref;
it travels with the snapshot like the built .so): fresh synthetic structure sets that no fixture has seen go through
`reseek -search` (-threads 1, see tests/compare_with_reference.py for why) and through rsk_search; the sorted hit
tables must be identical -- every mode, self and -db, prefilter path and long-chain (MKF / X-drop) pairs included.
Each case also has a committed golden of that binary's one-thread table (tests/golden/refbin_goldens.json, made by
tests/golden/make_refbin_goldens.py: row count + md5 of the sorted table + md5 of the seeded input files): where the binary did
not travel the cases run against the goldens instead of skipping, where it did they check both."""
import os
import sys

import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(fx.GOLDEN))
REF = os.path.join(ROOT, "oracle", "_ref", "reseek")
# The reference binary is built in the build container and travels to the GPU box with the snapshot.  Where it did not, the
# tests skip -- unless RSK_REQUIRE_REF=1 (set by tools/exp/run_all_gpu_tests.sh), which turns the silent skip of the strongest
# parity tests into a failure.
REQUIRE_REF = os.environ.get("RSK_REQUIRE_REF", "") not in ("", "0")
HAVE_GOLDENS = os.path.exists(os.path.join(fx.GOLDEN, "refbin_goldens.json"))
need_ref = pytest.mark.skipif(not os.path.exists(REF) and not REQUIRE_REF and not HAVE_GOLDENS,
                              reason="neither oracle/_ref/reseek nor tests/golden/refbin_goldens.json is present")


def test_reference_binary_present_when_required():
    if REQUIRE_REF:
        assert os.path.exists(REF), "RSK_REQUIRE_REF=1 but oracle/_ref/reseek did not travel: the reference-binary parity tests cannot run"


@need_ref
@pytest.mark.parametrize("n,mode,ndb,seed", [(260, "sensitive", 0, 3), (110, "verysensitive", 0, 4), (400, "fast", 0, 5),
                                            (60, "sensitive", 500, 6), (60, "fast", 500, 7)])
def test_hit_table_equals_the_reference_binary(n, mode, ndb, seed):
    import compare_with_reference as cwr
    res = cwr.compare(n, mode, ndb, threads=1, seed=seed, long_chains=4 if mode != "verysensitive" else 0)
    assert res["identical"], res
    assert res["reference_rows"] > 0
    assert "golden" in res and all(res["golden"].values()), res        # the committed one-thread table of the reference binary
    if os.path.exists(REF):
        # the golden route on its own (what a box without the binary runs)
        assert cwr.compare(n, mode, ndb, threads=1, seed=seed, long_chains=4 if mode != "verysensitive" else 0, use_binary=False)["identical"]
    if mode != "verysensitive":
        assert res["long_chain_pairs"] > 0        # chains >= 600: MKF seeding + GPU X-drop extensions took part


@need_ref
@pytest.mark.parametrize("n,mode,ndb,seed", [(24, "verysensitive", 160, 11), (48, "sensitive", 400, 12)])
def test_db_search_with_a_pdb_like_length_tail(n, mode, ndb, seed):
    """BASELINE configs[3] / configs[4] at test size: query batch vs a DB whose lengths are lognormal with a tail to 5,000
    (-verysensitive: every pair through SW + traceback, chains > 1024 in row groups / transposed; -sensitive: Mu filter
    fallback > 2048 and the long-chain path), reference binary vs rsk_search on data no fixture has seen."""
    import compare_with_reference as cwr
    res = cwr.compare(n, mode, ndb, threads=1, seed=seed, tail=True)
    assert res["identical"], res
    assert res["reference_rows"] > 0
    assert "golden" in res and all(res["golden"].values()), res
    if mode == "sensitive":
        assert res["long_chain_pairs"] > 0

"""BASELINE's acceptance wording as a test: "identical hit tables on SCOP40 all-vs-all".  The seeded 11,211-chain synthetic
.bca (SCOP40 lengths; the set bench.py's search_bca leg uses) is regenerated here, rsk_search runs the whole
`-search -sensitive` / `-search -fast` call on it, and the sorted hit table must have the row count and md5 that
oracle/_ref/reseek produced with ONE thread in the build container (tests/golden/make_full_golden.py; hours of CPU, so only
the digests are committed):
  full11211_<mode>.md5.txt          the complete table, when the reference run has finished;
  full11211_<mode>_prefix.md5.txt   the rows of every pair whose smaller chain index is below a cut -- what an interrupted
                                    1-thread run has written completely (it walks the pairs row-major, runself.cpp:72-99).
The .bca's own md5 is part of each golden: a drift of the generator (numpy / scipy) is reported as such, not as a search
difference."""
import hashlib
import json
import os
import sys
import tempfile

import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
sys.path.insert(0, fx.GOLDEN)


def golden(name):
    p = os.path.join(fx.GOLDEN, name)
    return json.load(open(p)) if os.path.exists(p) else None


@pytest.fixture(scope="module")
def bca():
    import make_full_golden as mfg
    d = tempfile.mkdtemp(prefix="rsk_full_")
    p = os.path.join(d, "syn11211.bca")
    assert mfg.synth_bca(p) == 11211
    yield p, mfg.file_md5(p), d
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)


@pytest.mark.parametrize("mode", ["sensitive", "fast"])
def test_full_size_hit_table_equals_the_reference(bca, mode):
    import make_full_golden as mfg
    import reseek_amd
    path, md5, d = bca
    full, prefix = golden("full11211_%s.md5.txt" % mode), golden("full11211_%s_prefix.md5.txt" % mode)
    assert full or prefix, "no full-size golden for -%s" % mode
    for g in (full, prefix):
        if g:
            assert g["bca_md5"] == md5, "the synthetic .bca differs from the one the golden was made from (generator drift, not a search difference)"
    ctx = reseek_amd.Ctx(0)
    out = os.path.join(d, "hits_%s.tsv" % mode)
    nhits, st = ctx.search(path, out, mode)
    ctx.close()
    assert st[0] == 11211 * 11212 // 2
    lines = open(out, "rb").read().splitlines()
    assert len(lines) == nhits
    if full:
        lines.sort()
        h = hashlib.md5()
        for ln in lines:
            h.update(ln + b"\n")
        assert (len(lines), h.hexdigest()) == (full["rows"], full["sorted_table_md5"]), "full table: %d rows vs %d" % (len(lines), full["rows"])
    if prefix:
        got_md5, got_rows = mfg.prefix_md5(lines, prefix["rows_with_min_index_below"])
        assert (got_rows, got_md5) == (prefix["rows"], prefix["sorted_rows_md5"]), "rows below chain %d: %d vs %d" % (
            prefix["rows_with_min_index_below"], got_rows, prefix["rows"])


def test_full_size_fast_db_equals_the_reference(bca):
    """BASELINE configs[2] at full size: `-search Q.bca -db Q.bca -fast` (search.cpp:76-111: MuPreFilter over 11,211 x 11,211,
    the per-query top-1500 bags, PostMuFilter of the 16.5 M candidates) must reproduce BOTH the reference's hand-off file
    (-keeptmp: md5 of the bytes) and its sorted hit table (tests/golden/full11211_fastdb.md5.txt; the golden comes from
    one-thread processes of the reference over target ranges, a route make_full_golden.py --fastdb --validate checks against
    the literal one-thread command on samples)."""
    import make_full_golden as mfg
    import reseek_amd
    g = golden("full11211_fastdb.md5.txt")
    assert g, "no full-size golden for -fast -db"
    path, md5, d = bca
    assert g["bca_md5"] == md5, "the synthetic .bca differs from the one the golden was made from (generator drift, not a search difference)"
    ctx = reseek_amd.Ctx(0)
    out = os.path.join(d, "hits_fastdb.tsv")
    nhits, st = ctx.search(path, out, "fast", db=path, keeptmp=1)
    ctx.close()
    tmp = out + ".prefilter.tmp"
    assert st[7] == 1 and st[0] == g["candidates"], "prefilter candidates: %d vs %d" % (st[0], g["candidates"])
    assert (os.path.getsize(tmp), mfg.file_md5(tmp)) == (g["handoff_bytes"], g["handoff_md5"]), "hand-off file differs from the reference's"
    if g.get("sorted_table_md5"):
        got_md5, got_rows = mfg.table_md5(out)
        assert (got_rows, got_md5) == (g["rows"], g["sorted_table_md5"]), "hit table: %d rows vs %d" % (got_rows, g["rows"])
        assert nhits == g["rows"]

"""Complete reference tables for the `-db` shapes of BASELINE configs[3] / configs[4] (runquery.cpp:82-125): the seeded
inputs of tests/golden/make_db_goldens.py are regenerated here, rsk_search runs the whole `-search Q -db DB` call, and the
sorted hit table must have the row count and md5 of ONE one-thread run of oracle/_ref/reseek in the build container
(tests/golden/db_<name>.md5.txt; only digests are committed -- the tables are 32-40 MB):
  c3db  256 SCOP40-length queries x 20,000 PDB-like chains (lognormal lengths, planted tail up to 5,000), -sensitive
  c4db  100 queries x 5,000 chains of the same shape, -verysensitive (every pair a hit row: SW + traceback + LDDT on all)
The md5 of each regenerated .bca is part of the golden: generator drift is reported as such, not as a search difference.
The sharded form of the same call (3 shards in sequence, the union of the tables) must give the same digest."""
import hashlib
import json
import os
import shutil
import sys
import tempfile

import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
sys.path.insert(0, fx.GOLDEN)


def sorted_md5(lines):
    lines = sorted(lines)
    h = hashlib.md5()
    for ln in lines:
        h.update(ln + b"\n")
    return len(lines), h.hexdigest(), (lines[0].decode() if lines else ""), (lines[-1].decode() if lines else "")


@pytest.mark.parametrize("name", ["c3db", "c4db"])
def test_db_search_equals_the_complete_reference_table(name):
    import make_db_goldens as mdg
    import reseek_amd
    p = os.path.join(fx.GOLDEN, "db_%s.md5.txt" % name)
    assert os.path.exists(p), "no golden for %s (tests/golden/make_db_goldens.py %s)" % (name, name)
    g = json.load(open(p))
    d = tempfile.mkdtemp(prefix="rsk_dbg_")
    try:
        q, db = mdg.gen_inputs(name, d)
        assert (mdg.file_md5(q), mdg.file_md5(db)) == (g["q_md5"], g["db_md5"]), \
            "the synthetic .bca files differ from the ones the golden was made from (generator drift, not a search difference)"
        ctx = reseek_amd.Ctx(0)
        out = os.path.join(d, "hits.tsv")
        nhits, st = ctx.search(q, out, g["mode"], db=db)
        assert st[0] == g["queries"] * g["db_chains"]
        lines = open(out, "rb").read().splitlines()
        assert len(lines) == nhits
        rows, md5, first, last = sorted_md5(lines)
        assert (rows, first, last) == (g["rows"], g["first_sorted_row"], g["last_sorted_row"])
        assert md5 == g["sorted_table_md5"], "%d rows, first/last equal, md5 differs" % rows
        # the same call as 3 shards of the DB (what 3 GPUs would run side by side): the union is the same table
        union = []
        for k in range(3):
            part = os.path.join(d, "part%d.tsv" % k)
            ctx.search(q, part, g["mode"], db=db, shard_index=k, shard_count=3)
            union += open(part, "rb").read().splitlines()
        ctx.close()
        assert sorted_md5(union)[:2] == (g["rows"], g["sorted_table_md5"])
    finally:
        shutil.rmtree(d, ignore_errors=True)

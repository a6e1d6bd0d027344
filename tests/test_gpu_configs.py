"""BASELINE configs[3] (query batch x large DB, -sensitive, pairs sharded) and configs[4] (query batch x PDB-scale DB,
-verysensitive: affine-gap SW + traceback on every pair) exercised at fixture size against the reference binary's
goldens (tests/golden/make_golden.sh section 12), and at one GPU's share of configs[3] through size-independent
properties (shard union == unsharded, counters add up)."""
import gzip
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
COLS = "query+target+qlo+qhi+ql+tlo+thi+tl+pctid+pvalue+evalue+cigar+dpscore+lddt+newts+ids+gaps+aq"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import torch
    import reseek_amd
    assert torch.cuda.is_available()
    c = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


@pytest.fixture(scope="module")
def work():
    d = tempfile.mkdtemp(prefix="rsk_cfg_")
    for name in ("q100.bca", "tailq.bca", "taildb.bca"):
        with gzip.open(os.path.join(fx.GOLDEN, name + ".gz"), "rb") as f, open(os.path.join(d, name), "wb") as g:
            g.write(f.read())
    sys.path.insert(0, fx.GOLDEN)
    import make_tail_bca
    make_tail_bca.subset_bca(os.path.join(d, "q100.bca"), os.path.join(d, "q32.bca"), 32)
    yield d
    shutil.rmtree(d, ignore_errors=True)


def check(ctx, work, q, db, mode, golden, **kw):
    out = os.path.join(work, "out.tsv")
    n, st = ctx.search(os.path.join(work, q), out, mode, db=os.path.join(work, db) if db else None, columns=COLS, **kw)
    got = sorted(open(out).read().splitlines())
    want = ["\t".join(r) for r in fx.read_tsv(golden)]
    assert n == len(got)
    assert got == want, "%d vs %d rows" % (len(got), len(want))
    return st


def test_config0_literal_32_chain_subset_sensitive(ctx, work):
    """BASELINE configs[0] as written: `reseek -search <32-chain .bca subset> -sensitive`, all-vs-all, bit-exact hit table
    (all columns and the default columns) against the reference binary run on the CPU."""
    st = check(ctx, work, "q32.bca", None, "sensitive", "hits_q32_sensitive.tsv.gz")
    assert st[0] == 32 * 33 // 2
    out = os.path.join(work, "out_std.tsv")
    n, _ = ctx.search(os.path.join(work, "q32.bca"), out, "sensitive")
    assert sorted(open(out).read().splitlines()) == ["\t".join(r) for r in fx.read_tsv("hits_q32_sensitive_std.tsv.gz")] and n == 60


def test_config4_shape_verysensitive_db_real_chains(ctx, work):
    st = check(ctx, work, "q32.bca", "q100.bca", "verysensitive", "hits_q32_db_q100_verysensitive.tsv.gz")
    assert st[0] == 3200 and st[5] == 3200 and st[4] == 0       # every pair through SW + traceback, no filter, no MKF


def test_dense_rectangle_batches_equal_the_pair_list_route(ctx, work, monkeypatch):
    """A -verysensitive -db pass aligns every pair of (database batch) x (queries): r06 cuts its alignment batches from length
    prefix sums (binary searches over the cells of the first k pairs) and lets every stage write its own slice of (i, j) instead of
    filling two index arrays per database batch.  Same tables as the pair-list route (RSK_DENSE_PAIR_LISTS=1) and as the reference,
    with batches of a few hundred pairs / few million cells so that cuts fall inside rows, at row ends and on single pairs."""
    want32 = ["\t".join(r) for r in fx.read_tsv("hits_q32_db_q100_verysensitive.tsv.gz")]
    wantt = ["\t".join(r) for r in fx.read_tsv("hits_tail_db_verysensitive.tsv.gz")]
    for pairs, cells in (("777", None), ("100", "3000000"), ("1", None), (None, "200000")):
        for dense in (None, "1"):
            for k, v in (("RSK_BATCH_PAIRS", pairs), ("RSK_BATCH_CELLS", cells), ("RSK_DENSE_PAIR_LISTS", dense)):
                if v is None:
                    monkeypatch.delenv(k, raising=False)
                else:
                    monkeypatch.setenv(k, v)
            if pairs != "1":
                out = os.path.join(work, "dense32.tsv")
                n, st = ctx.search(os.path.join(work, "q32.bca"), out, "verysensitive", db=os.path.join(work, "q100.bca"), columns=COLS)
                assert sorted(open(out).read().splitlines()) == want32 and st[0] == 3200 and st[5] == 3200, (pairs, cells, dense)
            out = os.path.join(work, "denset.tsv")
            n, st = ctx.search(os.path.join(work, "tailq.bca"), out, "verysensitive", db=os.path.join(work, "taildb.bca"), columns=COLS)
            assert sorted(open(out).read().splitlines()) == wantt and st[0] == 12 * 48, (pairs, cells, dense)


def test_config4_shape_length_tail_to_5000(ctx, work):
    st = check(ctx, work, "tailq.bca", "taildb.bca", "verysensitive", "hits_tail_db_verysensitive.tsv.gz")
    assert st[0] == 12 * 48 and st[5] == 12 * 48
    check(ctx, work, "taildb.bca", None, "verysensitive", "hits_taildb_self_verysensitive.tsv.gz")
    # sharded over 3 "GPUs" (run in sequence): union == the same table
    lines = []
    for k in range(3):
        out = os.path.join(work, "tail_v_%d.tsv" % k)
        ctx.search(os.path.join(work, "tailq.bca"), out, "verysensitive", db=os.path.join(work, "taildb.bca"), columns=COLS, shard_index=k, shard_count=3)
        lines += open(out).read().splitlines()
    assert sorted(lines) == ["\t".join(r) for r in fx.read_tsv("hits_tail_db_verysensitive.tsv.gz")]


def test_config3_shape_length_tail_sensitive(ctx, work):
    st = check(ctx, work, "tailq.bca", "taildb.bca", "sensitive", "hits_tail_db_sensitive.tsv.gz")
    assert st[4] > 100                                          # long-chain pairs (either chain >= 600)
    assert st[2] + st[4] == st[0]                               # filter input + MKF pairs = all pairs


def test_config3_one_gpu_share_properties(ctx, work):
    """256 queries x 12,000 synthetic DB chains, -sensitive (one GPU's share of configs[3] is 256 x 125,000; the same code
    path: streamed DB batches, Mu filter, float SW, long-chain path).  Properties: the union of 3 target shards is the
    unsharded table; the counters add up; every hit row names a query and a DB chain; hits == rows written."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_search
    lens = fx.scop40_lengths()
    rng = np.random.default_rng(33)
    q, db = os.path.join(work, "p_q.bca"), os.path.join(work, "p_db.bca")
    bench_search.write_bca_records(q, bench_search.gen_bca_chains(lens[rng.choice(len(lens), 256)], rng), labels=["Q%03d" % k for k in range(256)])
    bench_search.write_bca(db, lens[rng.choice(len(lens), 12000)], rng)
    out = os.path.join(work, "p_all.tsv")
    n, st = ctx.search(q, out, "sensitive", db=db)
    rows = open(out).read().splitlines()
    assert n == len(rows) and n > 1000
    assert st[0] == 256 * 12000 and st[2] + st[4] == st[0] and st[3] <= st[2] and st[5] == st[2] - st[3]
    for r in rows[:2000]:
        f = r.split("\t")
        assert f[0].startswith("Q") and f[1].startswith("syn")  # runquery.cpp:73 BaseOnAln(DA, Up = false): the query column is the query file's chain
    lines, pairs = [], 0
    for k in range(3):
        o = os.path.join(work, "p_%d.tsv" % k)
        nk, sk = ctx.search(q, o, "sensitive", db=db, shard_index=k, shard_count=3)
        lines += open(o).read().splitlines()
        pairs += sk[0]
    assert pairs == st[0]
    assert sorted(lines) == sorted(rows)
    # smaller streamed batches: the same table again
    os.environ["RSK_STREAM_CHAINS"] = "2500"
    try:
        o = os.path.join(work, "p_stream.tsv")
        ctx.search(q, o, "sensitive", db=db)
        assert sorted(open(o).read().splitlines()) == sorted(rows)
    finally:
        del os.environ["RSK_STREAM_CHAINS"]


def test_full_size_tool_machinery_at_small_scale():
    """tools/bench_configs_full.py (BASELINE configs[3] / configs[4] at their stated size as 8 sequential shards; its full run is
    committed as profiles/r04_configs_full.json) at 1/500 of the size: the 8-shard union equals the 3- and 1-shard unions --
    config4 through rsk_search_opts.hits_digest, the order-independent digest that replaces a 30 GB table --, pairs and
    hits add up, and the reference binary agrees on the sample."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs_full.py"), "--scale", "0.002"], capture_output=True, text=True,
                       cwd=ROOT, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(r.stdout)
    for key in ("config3", "config4"):
        e = d[key]
        assert e["union_8_equals_3_equals_1"] and e["pairs_and_hits_add_up"], key
        assert len(e["shards_8"]["shards"]) == 8 and e["shards_8"]["peak_host_rss_gb"] > 0
        assert e["vs_reference_on_sample"]["identical"] in (True, None), e["vs_reference_on_sample"]
    assert d["config4"]["hit_lines"].startswith("digest") and d["config4"]["shards_1"]["union_digest"][0] == d["config4"]["shards_1"]["hits"]

"""End-to-end hit tables: `reseek -search` through DBSearcher/DSSAligner mirrors + GPU kernels vs the
reference binary's own output (tests/golden/hits_*.tsv*, generated with -threads 1, compared sorted)."""
import gzip
import os
import shutil
import tempfile

import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
COLS = "query+target+qlo+qhi+ql+tlo+thi+tl+pctid+pvalue+evalue+cigar+dpscore+lddt+newts+ids+gaps+aq"


@pytest.fixture(scope="module")
def ctx():
    import torch
    import reseek_amd
    assert torch.cuda.is_available()
    c = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


@pytest.fixture(scope="module")
def tmpdir():
    d = tempfile.mkdtemp(prefix="rsk_search_")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def unpack(name, tmpdir):
    dst = os.path.join(tmpdir, name[:-3])
    if not os.path.exists(dst):
        with gzip.open(os.path.join(fx.GOLDEN, name), "rb") as f, open(dst, "wb") as g:
            g.write(f.read())
    return dst


def run(ctx, tmpdir, dbname, mode, columns, golden, db2=None, **kw):
    q = unpack(dbname, tmpdir)
    out = os.path.join(tmpdir, "out_%s_%s.tsv" % (dbname, mode))
    nhits, stats = ctx.search_rskdb(q, out, mode, db=unpack(db2, tmpdir) if db2 else None, columns=columns, **kw)
    got = sorted(open(out).read().splitlines())
    want = ["\t".join(r) for r in fx.read_tsv(golden)]
    assert nhits == len(got)
    if got != want:
        gs, ws = set(got), set(want)
        missing = sorted(ws - gs)[:5]
        extra = sorted(gs - ws)[:5]
        raise AssertionError("hit tables differ: %d vs %d rows\nmissing: %s\nextra: %s" % (len(got), len(want), missing, extra))
    return stats


def test_q100_verysensitive_all_columns(ctx, tmpdir):
    st = run(ctx, tmpdir, "q100_verysensitive.rskdb.gz", "verysensitive", COLS, "hits_q100_verysensitive.tsv.gz")
    assert st[0] == 5050 and st[4] == 0


def test_q100_sensitive_all_columns(ctx, tmpdir):
    st = run(ctx, tmpdir, "q100_sensitive.rskdb.gz", "sensitive", COLS, "hits_q100_sensitive.tsv.gz")
    assert st[0] == 5050 and st[4] == 490          # 490 pairs take the MKF path (+5 self-rev calls = the 495 of SURVEY 3.4)
    assert st[2] == 4560                           # m_MuFilterInputCount


def test_q100_fast_all_columns(ctx, tmpdir):
    run(ctx, tmpdir, "q100_fast.rskdb.gz", "fast", COLS, "hits_q100_fast.tsv.gz")


def test_q100_default_columns(ctx, tmpdir):
    run(ctx, tmpdir, "q100_sensitive.rskdb.gz", "sensitive", None, "hits_q100_sensitive_std.tsv.gz")
    run(ctx, tmpdir, "q100_verysensitive.rskdb.gz", "verysensitive", None, "hits_q100_verysensitive_std.tsv.gz")


def test_q10_sensitive(ctx, tmpdir):
    run(ctx, tmpdir, "q10_sensitive.rskdb.gz", "sensitive", COLS, "hits_q10_sensitive.tsv")


def test_palms_sensitive_long_chains_mkf(ctx, tmpdir):
    st = run(ctx, tmpdir, "palms_sensitive.rskdb.gz", "sensitive", COLS, "hits_palms_sensitive.tsv.gz")
    assert st[4] > 300      # lengths 418..2,099: most pairs take the MKF path


def test_palms_with_truncated_seed_lists(ctx, tmpdir, monkeypatch):
    """RSK_MKF_CAP=1: the device returns one seed HSP per pair, every pair with more takes the path that re-seeds
    with MuKmerFilter::Align on the host before its extensions go to the GPU -- same hit table."""
    monkeypatch.setenv("RSK_MKF_CAP", "1")
    run(ctx, tmpdir, "palms_sensitive.rskdb.gz", "sensitive", COLS, "hits_palms_sensitive.tsv.gz")


@pytest.mark.parametrize("own_streams", ["0", "1"])
def test_many_small_batches_pipeline(ctx, tmpdir, monkeypatch, own_streams):
    """RSK_BATCH_PAIRS=300: the alignment job becomes a pipeline of many batches (two GPU stages in flight on two
    contexts, hit replay of the batch before them on the host threads) with the long-chain job beside it; with
    RSK_OWN_STREAMS=1 the secondary contexts launch on non-blocking streams of their own.  Same hit tables."""
    monkeypatch.setenv("RSK_BATCH_PAIRS", "300")
    monkeypatch.setenv("RSK_OWN_STREAMS", own_streams)
    st = run(ctx, tmpdir, "q100_verysensitive.rskdb.gz", "verysensitive", COLS, "hits_q100_verysensitive.tsv.gz")
    assert st[0] == 5050
    run(ctx, tmpdir, "q100_sensitive.rskdb.gz", "sensitive", COLS, "hits_q100_sensitive.tsv.gz")
    run(ctx, tmpdir, "palms_sensitive.rskdb.gz", "sensitive", COLS, "hits_palms_sensitive.tsv.gz")
    run(ctx, tmpdir, "q100_sensitive.rskdb.gz", "sensitive", COLS, "hits_q100_db_q100_sensitive.tsv.gz", db2="q100_sensitive_dbq.rskdb.gz")
    # PostMuFilter (the -fast -db path) consumes its candidates through the same pipeline
    run(ctx, tmpdir, "q100_sensitive_dbq.rskdb.gz", "fast", COLS, "hits_q100_db_q100_fast.tsv.gz", db2="q100_sensitive_dbq.rskdb.gz")
    run_bca(ctx, tmpdir, "q100.bca", "sensitive", COLS, "hits_q100_sensitive.tsv.gz")


def test_q100_vs_db_q100_sensitive(ctx, tmpdir):
    """`reseek -search Q -db DB -sensitive` (Search_NoMuFilter search.cpp:39): A = streamed DB chain whose
    self-rev score is computed under the search params (runquery.cpp:43-44), B = query chain."""
    st = run(ctx, tmpdir, "q100_sensitive.rskdb.gz", "sensitive", COLS, "hits_q100_db_q100_sensitive.tsv.gz",
             db2="q100_sensitive_dbq.rskdb.gz")
    assert st[0] == 10000


def test_q100_vs_db_q100_fast_prefilter_path(ctx, tmpdir, monkeypatch):
    """`reseek -search Q -db DB -fast` (search.cpp:76-111): MuPreFilter (k-mer neighbourhoods, idxq for 100
    queries) -> hand-off TSV -> PostMuFilter (AlignBags under the sensitive preset).  Both the hand-off
    file (-keeptmp) and the hit table must equal the reference's."""
    monkeypatch.setenv("RSK_KEEPTMP", "1")
    st = run(ctx, tmpdir, "q100_sensitive_dbq.rskdb.gz", "fast", COLS, "hits_q100_db_q100_fast.tsv.gz",
             db2="q100_sensitive_dbq.rskdb.gz")
    out = os.path.join(tmpdir, "out_q100_sensitive_dbq.rskdb.gz_fast.tsv")
    want_tmp = gzip.open(os.path.join(fx.GOLDEN, "prefilter_q100_db_q100_fast_tmp.tsv.gz")).read().decode()
    assert open(out + ".prefilter.tmp").read() == want_tmp
    assert st[7] == 1 and st[0] == sum(int(ln.split("\t")[1]) for ln in want_tmp.splitlines()[1:])
    run(ctx, tmpdir, "q100_sensitive_dbq.rskdb.gz", "fast", None, "hits_q100_db_q100_fast_std.tsv.gz",
        db2="q100_sensitive_dbq.rskdb.gz")


def unpack_bca(name, tmpdir):
    dst = os.path.join(tmpdir, name)
    if not os.path.exists(dst):
        with gzip.open(os.path.join(fx.GOLDEN, name + ".gz"), "rb") as f, open(dst, "wb") as g:
            g.write(f.read())
    return dst


def run_bca(ctx, tmpdir, q, mode, columns, golden, db=None):
    out = os.path.join(tmpdir, "outbca_%s_%s_%s.tsv" % (q, db, mode))
    nhits, stats = ctx.search_rskdb(unpack_bca(q, tmpdir), out, mode, db=unpack_bca(db, tmpdir) if db else None, columns=columns)
    got = sorted(open(out).read().splitlines())
    want = ["\t".join(r) for r in fx.read_tsv(golden)]
    if got != want:
        gs, ws = set(got), set(want)
        raise AssertionError("hit tables differ: %d vs %d rows\nmissing: %s\nextra: %s" % (len(got), len(want), sorted(ws - gs)[:5], sorted(gs - ws)[:5]))
    return stats


def test_bca_input_all_modes(ctx, tmpdir):
    """The whole path from the reference's own .bca test files: BCAData reader, DSS featurisation on the host,
    self-rev scores (P8) on the GPU, then the searches above -- against the same golden hit tables."""
    run_bca(ctx, tmpdir, "q10.bca", "sensitive", COLS, "hits_q10_sensitive.tsv")
    run_bca(ctx, tmpdir, "q100.bca", "sensitive", COLS, "hits_q100_sensitive.tsv.gz")
    run_bca(ctx, tmpdir, "q100.bca", "fast", COLS, "hits_q100_fast.tsv.gz")
    run_bca(ctx, tmpdir, "q100.bca", "verysensitive", None, "hits_q100_verysensitive_std.tsv.gz")


def test_bca_input_long_chains(ctx, tmpdir):
    run_bca(ctx, tmpdir, "palms.bca", "sensitive", COLS, "hits_palms_sensitive.tsv.gz")


def test_bca_input_db_modes(ctx, tmpdir):
    run_bca(ctx, tmpdir, "q100.bca", "sensitive", COLS, "hits_q100_db_q100_sensitive.tsv.gz", db="q100.bca")
    run_bca(ctx, tmpdir, "q100.bca", "fast", COLS, "hits_q100_db_q100_fast.tsv.gz", db="q100.bca")


def test_bca_fast_db_with_dbmu_and_options(ctx, tmpdir):
    """-dbmu (prefilter targets from the Mu FASTA that `-convert -feature_fasta` wrote; both sides then see the
    exchanged letters 10/11, so candidates and hits differ from the run without -dbmu -- as in the reference),
    plus -keeptmp and -rsb_size through the options struct."""
    from reseek_amd import capi
    q = unpack_bca("q100.bca", tmpdir)
    fa = os.path.join(tmpdir, "q100_dbmu.mu.fa")
    capi.bca_to_mu_fasta(q, fa)
    out = os.path.join(tmpdir, "out_dbmu.tsv")
    nhits, st = ctx.search(q, out, "fast", db=q, columns=COLS, dbmu=fa, keeptmp=1)
    want = ["\t".join(r) for r in fx.read_tsv("hits_q100_db_q100_fast_dbmu.tsv.gz")]
    assert sorted(open(out).read().splitlines()) == want and nhits == len(want)
    want_tmp = gzip.open(os.path.join(fx.GOLDEN, "prefilter_q100_db_q100_fast_dbmu_tmp.tsv.gz")).read().decode()
    assert open(out + ".prefilter.tmp").read() == want_tmp
    # a tiny bag keeps fewer candidates
    out2 = os.path.join(tmpdir, "out_rsb5.tsv")
    n2, st2 = ctx.search(q, out2, "fast", db=q, rsb_size=5)
    assert st2[0] <= 5 * 100 and 0 < n2 <= nhits
    # -evalue through the struct == through the positional API
    out3 = os.path.join(tmpdir, "out_e.tsv")
    n3, _ = ctx.search(q, out3, "sensitive", evalue=1e-3, columns=COLS)
    out4 = os.path.join(tmpdir, "out_e2.tsv")
    n4, _ = ctx.search_rskdb(q, out4, "sensitive", columns=COLS, evalue=1e-3)
    assert n3 == n4 and sorted(open(out3).read().splitlines()) == sorted(open(out4).read().splitlines())


def test_sharded_search_equals_unsharded(ctx, tmpdir):
    """the union of the shards' hit tables is the unsharded table (self search: windows of the set's length order +
    the long-chain list in contiguous pieces, r06; -db mode: DB chains cut by residues).  The shards run one after another on this
    single GPU."""
    q = unpack_bca("q100.bca", tmpdir)
    for db, golden in ((None, "hits_q100_sensitive.tsv.gz"), (q, "hits_q100_db_q100_sensitive.tsv.gz")):
        for count in (2, 3, 7, 8):
            lines, pairs = [], 0
            for idx in range(count):
                out = os.path.join(tmpdir, "shard_%s_%d_%d.tsv" % ("db" if db else "self", count, idx))
                n, st = ctx.search(q, out, "sensitive", db=db, columns=COLS, shard_index=idx, shard_count=count)
                got = open(out).read().splitlines()
                assert n == len(got)
                lines += got
                pairs += st[0]
            want = ["\t".join(r) for r in fx.read_tsv(golden)]
            assert sorted(lines) == want
            assert pairs == (10000 if db else 5050)
    # palms: shards with MKF pairs (every chain is long: the long-chain list IS the pair space, cut into contiguous pieces), N = 4 and 8;
    # the counters of the shards add up to the unsharded call's
    p = unpack_bca("palms.bca", tmpdir)
    n1, st1 = ctx.search(p, os.path.join(tmpdir, "palms_one.tsv"), "sensitive", columns=COLS)
    for count in (4, 8):
        lines, st = [], np.zeros(8, np.int64)
        for idx in range(count):
            out = os.path.join(tmpdir, "shard_palms_%d.tsv" % idx)
            _, s8 = ctx.search(p, out, "sensitive", columns=COLS, shard_index=idx, shard_count=count)
            lines += open(out).read().splitlines()
            st += np.array(s8, np.int64)
        assert sorted(lines) == ["\t".join(r) for r in fx.read_tsv("hits_palms_sensitive.tsv.gz")]
        assert list(st[:7]) == list(np.array(st1, np.int64)[:7]), (count, st, st1)
    # the tail set (short chains + a few long ones: both kinds of pairs in every window), -noself, and the r01-r05 cut (target
    # ranges of the chain order: still the route without a Mu filter and beyond one filter pass) through RSK_SELF_SHARD_RANGES
    t = unpack_bca("taildb.bca", tmpdir)
    for src, kws in ((t, ({}, {"noself": 1})), (q, ({"noself": 1},))):
        for kw in kws:
            one = os.path.join(tmpdir, "one.tsv")
            n1, st1 = ctx.search(src, one, "sensitive", columns=COLS, **kw)
            want = sorted(open(one).read().splitlines())
            for env in (None, "1"):
                if env:
                    os.environ["RSK_SELF_SHARD_RANGES"] = env
                try:
                    lines, st = [], np.zeros(8, np.int64)
                    for idx in range(8):
                        out = os.path.join(tmpdir, "shard8_%d.tsv" % idx)
                        _, s8 = ctx.search(src, out, "sensitive", columns=COLS, shard_index=idx, shard_count=8, **kw)
                        lines += open(out).read().splitlines()
                        st += np.array(s8, np.int64)
                finally:
                    os.environ.pop("RSK_SELF_SHARD_RANGES", None)
                assert sorted(lines) == want, (src, kw, env)
                assert list(st[:7]) == list(np.array(st1, np.int64)[:7]), (src, kw, env, st, st1)
    from reseek_amd import capi
    with pytest.raises(capi.RskError):
        ctx.search(q, os.path.join(tmpdir, "x.tsv"), "fast", db=q, shard_index=0, shard_count=2)


def test_edge_case_chains(ctx, tmpdir):
    """Ragged input (tests/golden/make_edge_bca.py): chains of 1, 2, 3, 5, 7, 8, 12, 31..33, 63..65 residues (below the
    k-mer / DSS window sizes), a 2099-residue chain (MKF + row groups), an identical chain under another label and
    a repeated label -- every mode, -db with and without -noself, and the two-stage -fast -db path."""
    e = unpack_bca("edge.bca", tmpdir)
    run_bca(ctx, tmpdir, "edge.bca", "sensitive", COLS, "hits_edge_sensitive.tsv.gz")
    run_bca(ctx, tmpdir, "edge.bca", "fast", COLS, "hits_edge_fast.tsv.gz")
    run_bca(ctx, tmpdir, "edge.bca", "verysensitive", COLS, "hits_edge_verysensitive.tsv.gz")
    run_bca(ctx, tmpdir, "edge.bca", "sensitive", COLS, "hits_edge_db.tsv.gz", db="edge.bca")
    out = os.path.join(tmpdir, "edge_noself.tsv")
    ctx.search(e, out, "sensitive", db=e, columns=COLS, noself=1)
    assert sorted(open(out).read().splitlines()) == ["\t".join(r) for r in fx.read_tsv("hits_edge_db_noself.tsv.gz")]
    out = os.path.join(tmpdir, "edge_fastdb.tsv")
    ctx.search(e, out, "fast", db=e, columns=COLS, keeptmp=1)
    assert sorted(open(out).read().splitlines()) == ["\t".join(r) for r in fx.read_tsv("hits_edge_fastdb.tsv.gz")]
    assert open(out + ".prefilter.tmp").read() == gzip.open(os.path.join(fx.GOLDEN, "prefilter_edge_fastdb_tmp.tsv.gz")).read().decode()


def test_tiled_pair_space_equals_one_pass(ctx, tmpdir, monkeypatch):
    """Pair spaces larger than one Mu-filter pass run in target blocks (self search: triangle + rectangle per block, as the
    multi-GPU shards) or row blocks of the streamed set (-db); the survivor lists start small and are re-run on overflow.
    With the tile forced down to 1500 pairs every fixture takes that path and must reproduce the same tables."""
    monkeypatch.setenv("RSK_FILTER_TILE_PAIRS", "1500")
    run_bca(ctx, tmpdir, "q100.bca", "sensitive", COLS, "hits_q100_sensitive.tsv.gz")
    run_bca(ctx, tmpdir, "q100.bca", "fast", COLS, "hits_q100_fast.tsv.gz")
    run_bca(ctx, tmpdir, "palms.bca", "sensitive", COLS, "hits_palms_sensitive.tsv.gz")
    run_bca(ctx, tmpdir, "q100.bca", "sensitive", COLS, "hits_q100_db_q100_sensitive.tsv.gz", db="q100.bca")
    st = run(ctx, tmpdir, "q100_sensitive.rskdb.gz", "sensitive", COLS, "hits_q100_sensitive.tsv.gz")
    assert st[0] == 5050 and st[4] == 490 and st[2] == 4560          # the counters add up over the blocks
    q = unpack_bca("q100.bca", tmpdir)
    lines = []
    for idx in range(3):
        out = os.path.join(tmpdir, "tiled_shard_%d.tsv" % idx)
        ctx.search(q, out, "sensitive", columns=COLS, shard_index=idx, shard_count=3)
        lines += open(out).read().splitlines()
    assert sorted(lines) == ["\t".join(r) for r in fx.read_tsv("hits_q100_sensitive.tsv.gz")]


def test_streamed_db_batches(ctx, tmpdir, monkeypatch):
    """RunQuery(ChainReader2 &): the -db file streams in batches of RSK_STREAM_CHAINS chains (loader thread featurises
    batch k + 1 while batch k is searched); 7 chains per batch = 15 batches of q100."""
    monkeypatch.setenv("RSK_STREAM_CHAINS", "7")
    q = unpack_bca("q100.bca", tmpdir)
    out = os.path.join(tmpdir, "streamed.tsv")
    n, st = ctx.search(q, out, "sensitive", db=q, columns=COLS)
    assert sorted(open(out).read().splitlines()) == ["\t".join(r) for r in fx.read_tsv("hits_q100_db_q100_sensitive.tsv.gz")]
    assert st[0] == 10000
    p = unpack_bca("palms.bca", tmpdir)
    out1, out2 = os.path.join(tmpdir, "palms_db_stream.tsv"), os.path.join(tmpdir, "palms_db_whole.tsv")
    ctx.search(p, out1, "sensitive", db=p, columns=COLS)
    monkeypatch.delenv("RSK_STREAM_CHAINS")
    ctx.search(p, out2, "sensitive", db=p, columns=COLS)
    assert sorted(open(out1).read().splitlines()) == sorted(open(out2).read().splitlines())


def test_long_chain_device_batch_from_bca(ctx, tmpdir, monkeypatch):
    """The long-chain pairs go through ONE device batch after the chaining (rsk_mkf_align_pairs: mega-HSP scores, 8-mer
    start, both X-drop extensions, MergeFwdBwd, LDDT / E-value), and so do the self-rev alignments of the long chains
    (chain against its reversed copy, alignpair.cpp:7) when the search starts from a .bca: the library has no host X-drop.
    RSK_MKF_CAP=2 truncates the device seed lists so that MuKmerFilter::Align re-seeds those pairs on the host before they
    join the batch.  Golden tables either way."""
    run_bca(ctx, tmpdir, "palms.bca", "sensitive", COLS, "hits_palms_sensitive.tsv.gz")
    run_bca(ctx, tmpdir, "edge.bca", "sensitive", COLS, "hits_edge_sensitive.tsv.gz")
    monkeypatch.setenv("RSK_MKF_CAP", "2")
    run_bca(ctx, tmpdir, "palms.bca", "sensitive", COLS, "hits_palms_sensitive.tsv.gz")


def test_hits_digest_is_the_digest_of_the_table(ctx, tmp_path):
    """rsk_search_opts.hits_digest: the one-line digest (lines, bytes, sum and xor of a 64-bit hash per line) equals the same
    quantities computed here from the table the same search writes; shard digests combine to it."""
    import gzip
    from reseek_amd import capi
    bca = str(tmp_path / "q100.bca")
    with gzip.open(os.path.join(fx.GOLDEN, "q100.bca.gz"), "rb") as f, open(bca, "wb") as g:
        g.write(f.read())
    M = (1 << 64) - 1

    def h64(b):
        n = len(b)
        h = 0x9E3779B97F4A7C15 ^ ((n * 0xD6E8FEB86659FD93) & M)
        for i in range(0, n, 8):
            v = int.from_bytes(b[i:i + 8], "little")
            h = ((h ^ v) * 0xFF51AFD7ED558CCD) & M
            h ^= h >> 32
        h = (h * 0xC4CEB9FE1A85EC53) & M
        return h ^ (h >> 29)

    for mode in ("sensitive", "verysensitive"):
        tab, dig = str(tmp_path / "t.tsv"), str(tmp_path / "d.tsv")
        ctx.search(bca, tab, mode, db=bca)
        ctx.search(bca, dig, mode, db=bca, hits_digest=1)
        lines = open(tab, "rb").read().splitlines()
        s = x = 0
        for ln in lines:
            v = h64(ln)
            s = (s + v) & M
            x ^= v
        assert capi.read_hits_digest(dig) == (len(lines), os.path.getsize(tab), s, x), mode
        parts = []
        for k in range(3):
            ctx.search(bca, dig, mode, db=bca, hits_digest=1, shard_index=k, shard_count=3)
            parts.append(capi.read_hits_digest(dig))
        assert capi.combine_hits_digests(parts) == (len(lines), os.path.getsize(tab), s, x), mode


@pytest.mark.parametrize("ranges", ["3", "7"])
def test_fast_db_scanned_in_target_ranges(ctx, tmpdir, monkeypatch, ranges):
    """The prefilter of `-search -fast -db` scans the DB in contiguous target ranges and replays the bags of range k while
    the device scans range k + 1 (a set the size of q100 takes one range; RSK_PF_RANGES forces several).  The hand-off file
    and the hit table must stay the reference's -- also with bags of 5 that overflow in every range (-rsb_size 5: the
    truncation sequence of a bag runs across the ranges)."""
    monkeypatch.setenv("RSK_PF_RANGES", ranges)
    q = unpack_bca("q100.bca", tmpdir)
    out = os.path.join(tmpdir, "out_ranges.tsv")
    ctx.search(q, out, "fast", db=q, columns=COLS, keeptmp=1)
    assert sorted(open(out).read().splitlines()) == ["\t".join(r) for r in fx.read_tsv("hits_q100_db_q100_fast.tsv.gz")]
    assert open(out + ".prefilter.tmp").read() == gzip.open(os.path.join(fx.GOLDEN, "prefilter_q100_db_q100_fast_tmp.tsv.gz")).read().decode()
    ctx.search(q, out, "fast", db=q, columns=COLS, keeptmp=1, rsb_size=5)
    assert sorted(open(out).read().splitlines()) == ["\t".join(r) for r in fx.read_tsv("hits_q100_db_q100_fast_rsb5.tsv.gz")]
    assert open(out + ".prefilter.tmp").read() == gzip.open(os.path.join(fx.GOLDEN, "prefilter_q100_db_q100_fast_rsb5_tmp.tsv.gz")).read().decode()


def test_malformed_rskdb_containers_are_errors(ctx, tmpdir):
    """LoadDB reads the container in one piece and indexes the records before any chain is built: a file cut anywhere
    (header, inside a record, inside the stored k-mers), a wrong magic, a wrong feature count and stored Mu k-mers that
    disagree with the letters are errors of the call (an RskError with a message), never a crash or a silent short read;
    the intact file still searches afterwards."""
    import struct
    import reseek_amd
    src = unpack("q10_sensitive.rskdb.gz", tmpdir) if os.path.exists(os.path.join(fx.GOLDEN, "q10_sensitive.rskdb.gz")) else unpack("q100_sensitive.rskdb.gz", tmpdir)
    buf = open(src, "rb").read()
    out = os.path.join(tmpdir, "malformed.tsv")
    L0, ll0 = struct.unpack_from("<II", buf, 16)
    first_len = 8 + ll0 + L0 * (2 + 8) + 12 * L0 + 4
    (nk0,) = struct.unpack_from("<I", buf, 16 + first_len)
    cases = {
        "header only": buf[:12],
        "cut inside the first record": buf[:16 + first_len // 2],
        "cut inside the first record's k-mers": buf[:16 + first_len + 4 + 2 * nk0],
        "cut before the last byte": buf[:-1],
        "wrong magic": b"RSKDB2\0\0" + buf[8:],
        "wrong feature count": buf[:12] + struct.pack("<I", 7) + buf[16:],
        "k-mers disagree": buf[:16 + first_len + 4] + struct.pack("<I", 46655 - struct.unpack_from("<I", buf, 16 + first_len + 4)[0]) + buf[16 + first_len + 8:],
        # a chain count the file cannot hold must fail before anything is sized by it (~34 GB vector)
        "hostile chain count": buf[:8] + struct.pack("<I", 0xFFFFFFF0) + buf[12:],
        # letters index device tables: a feature letter beyond its alphabet (16 in a 16-letter feature) and a Mu letter >= 36
        "feature letter out of range": buf[:16 + 8 + ll0 + 2 * L0 + L0] + b"\x10" + buf[16 + 8 + ll0 + 2 * L0 + L0 + 1:],
        "amino-acid feature letter out of range": buf[:16 + 8 + ll0 + 2 * L0] + b"\x14" + buf[16 + 8 + ll0 + 2 * L0 + 1:],
        "Mu letter out of range": buf[:16 + 8 + ll0 + L0] + b"\x24" + buf[16 + 8 + ll0 + L0 + 1:],
    }
    for what, data in cases.items():
        bad = os.path.join(tmpdir, "bad.rskdb")
        with open(bad, "wb") as f:
            f.write(data)
        with pytest.raises(reseek_amd.RskError) as e:
            ctx.search_rskdb(bad, out, "sensitive")
        assert "LoadDB" in str(e.value), (what, str(e.value))
    n, _ = ctx.search_rskdb(src, out, "sensitive")
    assert n > 0


def test_bca_to_rskdb_slices_and_prepared_search(ctx, tmpdir):
    """rsk_bca_to_rskdb (r06): the featurised form of a .bca file -- what LoadDB computes per chain -- as the RSKDB1 container
    rsk_search reads without featurising.  (a) a search from the prepared container writes the table of the search from the
    .bca file (the reference's goldens), self search in every mode; (b) the containers of 1, 3 and 8 slices of the chains,
    concatenated (reseek_amd.dist.merge_rskdb: what the ranks of a multi-GPU run all-gather), are byte-identical to the
    one-slice container -- slices with long chains (palms: self-rev through the long-chain batch) included."""
    from reseek_amd import dist as rdist
    for name, golds in (("q100.bca", {"sensitive": "hits_q100_sensitive.tsv.gz", "fast": "hits_q100_fast.tsv.gz", "verysensitive": "hits_q100_verysensitive.tsv.gz"}),
                        ("palms.bca", {"sensitive": "hits_palms_sensitive.tsv.gz"})):
        src = unpack_bca(name, tmpdir)
        for mode, gold in golds.items():
            if not os.path.exists(os.path.join(fx.GOLDEN, gold)):
                continue
            whole = os.path.join(tmpdir, "whole_%s.rskdb" % mode)
            n = ctx.bca_to_rskdb(src, whole, mode)
            assert n > 0
            out = os.path.join(tmpdir, "prepared.tsv")
            nh, st = ctx.search(whole, out, mode, columns=COLS)
            assert sorted(open(out).read().splitlines()) == ["\t".join(r) for r in fx.read_tsv(gold)], (name, mode)
            for count in (3, 8):
                parts = []
                for k in range(count):
                    part = os.path.join(tmpdir, "part%d.rskdb" % k)
                    ctx.bca_to_rskdb(src, part, mode, shard_index=k, shard_count=count)
                    parts.append(open(part, "rb").read())
                assert rdist.merge_rskdb(parts) == open(whole, "rb").read(), (name, mode, count)
    from reseek_amd import capi
    with pytest.raises(capi.RskError):
        ctx.bca_to_rskdb(os.path.join(tmpdir, "missing.bca"), os.path.join(tmpdir, "x.rskdb"), "sensitive")
    with pytest.raises(capi.RskError):
        ctx.bca_to_rskdb(src, os.path.join(tmpdir, "x.rskdb"), "sensitive", shard_index=3, shard_count=3)

"""SURVEY 8e from ONE process: a DBSearcher that drives several device contexts (DBSearcher::m_Devices / RSK_DEVICES /
rsk_search_opts.devices) -- one context, one host thread and one shard of the pair space per list entry, hit lines into one
file.  The lists come from fixtures.device_list(k): k entries over the DISTINCT devices of the box where it has several, device 0
repeated ("0,0", "0,0,0") on the one-GPU box, where the shards then run CONCURRENTLY on
separate contexts and streams of that device, which is the part a single GPU can check (ranges, replication, concurrent
writers, counters); nothing here depends on the entries being distinct devices.  The tables must equal the reference
binary's goldens.  The C++ form of the same (tests/ref_shaped/search_main.cpp -devices, and the reference's own search.cpp
under RSK_DEVICES) is in the second half."""
import gzip
import os
import shutil
import subprocess
import tempfile

import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = "query+target+qlo+qhi+ql+tlo+thi+tl+pctid+pvalue+evalue+cigar+dpscore+lddt+newts+ids+gaps+aq"


@pytest.fixture(scope="module")
def ctx():
    import reseek_amd
    c = reseek_amd.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def work():
    d = tempfile.mkdtemp(prefix="rsk_multidev_")
    for name in ("q100.bca", "palms.bca", "tailq.bca", "taildb.bca"):
        with gzip.open(os.path.join(fx.GOLDEN, name + ".gz"), "rb") as f, open(os.path.join(d, name), "wb") as g:
            g.write(f.read())
    yield d
    shutil.rmtree(d, ignore_errors=True)


def table(path):
    return sorted(open(path).read().splitlines())


def golden(name):
    return ["\t".join(r) for r in fx.read_tsv(name)]


@pytest.mark.parametrize("entries", [2, 3])
def test_self_search_on_several_contexts(ctx, work, entries):
    devices = fx.device_list(entries)
    out = os.path.join(work, "self.tsv")
    n, st = ctx.search(os.path.join(work, "q100.bca"), out, "sensitive", columns=COLS, devices=devices)
    assert table(out) == golden("hits_q100_sensitive.tsv.gz") and n == len(golden("hits_q100_sensitive.tsv.gz"))
    assert st[0] == 5050 and st[4] == 490                 # the counters of the shards add up to the single-context run's
    n, st = ctx.search(os.path.join(work, "palms.bca"), out, "sensitive", columns=COLS, devices=devices)       # long-chain pairs in every shard
    assert table(out) == golden("hits_palms_sensitive.tsv.gz")
    n, st = ctx.search(os.path.join(work, "q100.bca"), out, "verysensitive", columns=COLS, devices=devices)
    assert table(out) == golden("hits_q100_verysensitive.tsv.gz") and st[0] == 5050


@pytest.mark.parametrize("entries", [2, 3])
def test_db_search_on_several_contexts(ctx, work, entries, monkeypatch):
    devices = fx.device_list(entries)
    out = os.path.join(work, "db.tsv")
    q = os.path.join(work, "q100.bca")
    n, st = ctx.search(q, out, "sensitive", db=q, columns=COLS, devices=devices)       # every context streams its own range of the -db file
    assert table(out) == golden("hits_q100_db_q100_sensitive.tsv.gz") and st[0] == 10000
    monkeypatch.setenv("RSK_STREAM_CHAINS", "7")                                        # several streamed batches per context
    n, st = ctx.search(os.path.join(work, "tailq.bca"), out, "verysensitive", db=os.path.join(work, "taildb.bca"), columns=COLS, devices=devices)
    assert table(out) == golden("hits_tail_db_verysensitive.tsv.gz")
    n, st = ctx.search(os.path.join(work, "tailq.bca"), out, "sensitive", db=os.path.join(work, "taildb.bca"), columns=COLS, devices=devices)
    assert table(out) == golden("hits_tail_db_sensitive.tsv.gz")


def test_fast_db_two_stage_on_several_contexts(ctx, work):
    """-fast -db: target shards per context, the per-query top-B exchange in host memory, PostMuFilter per shard; hit table
    and merged hand-off file equal the reference's."""
    out = os.path.join(work, "fastdb.tsv")
    q = os.path.join(work, "q100.bca")
    for devices in (fx.device_list(2), fx.device_list(3)):
        n, st = ctx.search(q, out, "fast", db=q, columns=COLS, devices=devices, keeptmp=1)
        want = golden("hits_q100_db_q100_fast.tsv.gz")
        assert table(out) == want and n == len(want)
        with gzip.open(os.path.join(fx.GOLDEN, "prefilter_q100_db_q100_fast_tmp.tsv.gz"), "rt") as f:
            assert open(out + ".prefilter.tmp").read() == f.read()


def test_fast_db_top_b_cut_is_the_single_device_one(ctx, work):
    """A bag of 5 (-rsb_size) makes every query overflow its bag, so the kept candidates depend on the reference's
    truncation sequence and quicksort tie order (rankedscoresbag.cpp:34-51).  Within one process the shards' complete triple
    lists go through the same replay as the single-device path: identical hit table and hand-off file for any device list."""
    q = os.path.join(work, "q100.bca")
    outs = []
    for devices in (None, fx.device_list(2), fx.device_list(3)):
        out = os.path.join(work, "rsb5_%s.tsv" % (devices or "one").replace(",", "_"))
        kw = {"devices": devices} if devices else {}
        n, st = ctx.search(q, out, "fast", db=q, columns=COLS, rsb_size=5, keeptmp=1, **kw)
        outs.append((table(out), open(out + ".prefilter.tmp").read(), n))
    assert outs[0][0] and outs[0] == outs[1] == outs[2]
    want = sorted("\t".join(r) for r in fx.read_tsv("hits_q100_db_q100_fast_rsb5.tsv.gz"))     # the reference binary, -rsb_size 5
    with gzip.open(os.path.join(fx.GOLDEN, "prefilter_q100_db_q100_fast_rsb5_tmp.tsv.gz"), "rt") as f:
        assert outs[2][1] == f.read()
    assert sorted(outs[2][0]) == want


def test_environment_list_and_bad_lists(ctx, work, monkeypatch):
    out = os.path.join(work, "env.tsv")
    monkeypatch.setenv("RSK_DEVICES", "0, 0")
    ctx.search(os.path.join(work, "q100.bca"), out, "sensitive", columns=COLS)
    assert table(out) == golden("hits_q100_sensitive.tsv.gz")
    monkeypatch.delenv("RSK_DEVICES")
    from reseek_amd import capi
    with pytest.raises(capi.RskError):
        ctx.search(os.path.join(work, "q100.bca"), out, "sensitive", devices="0,x")


# ---- the same from C++ ---------------------------------------------------------------------------------------------------
def _compile(src, exe):
    cxx = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "g++"
    cmd = [cxx] + (["-x", "c++"] if cxx.endswith("hipcc") else []) + [
        "-std=c++17", "-O1", "-I", os.path.join(ROOT, "reseek_amd", "csrc", "host"), src, "-L", os.path.join(ROOT, "reseek_amd"), "-lrsk",
        "-Wl,-rpath," + os.path.join(ROOT, "reseek_amd"), "-pthread", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def _run(exe, work, args, gold, env=None):
    out = os.path.join(work, "cpp_hits.tsv")
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([exe] + args + ["-output", out, "-columns", COLS], capture_output=True, text=True, cwd=work, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    assert table(out) == golden(gold)


def test_cpp_driver_with_a_device_list(work):
    exe = os.path.join(work, "search_main")
    _compile(os.path.join(ROOT, "tests", "ref_shaped", "search_main.cpp"), exe)
    _run(exe, work, ["q100.bca", "-sensitive", "-devices", fx.device_list(2)], "hits_q100_sensitive.tsv.gz")
    _run(exe, work, ["palms.bca", "-sensitive", "-devices", fx.device_list(3)], "hits_palms_sensitive.tsv.gz")
    _run(exe, work, ["q100.bca", "-db", "q100.bca", "-sensitive", "-devices", fx.device_list(2)], "hits_q100_db_q100_sensitive.tsv.gz")


def test_the_reference_search_cpp_runs_on_a_device_list(work):
    """oracle/_ref/search_refsrc = the reference's own search.cpp compiled against reseek_host.h: DBSearcher::RunSelf /
    RunQuery(ChainReader2 &) fan out over RSK_DEVICES without the caller knowing."""
    exe = os.path.join(ROOT, "oracle", "_ref", "search_refsrc")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/search_refsrc not built (make -f oracle/Makefile.ref where /root/reference exists)")
    _run(exe, work, ["q100.bca", "-sensitive"], "hits_q100_sensitive.tsv.gz", env={"RSK_DEVICES": fx.device_list(2)})
    _run(exe, work, ["q100.bca", "-db", "q100.bca", "-sensitive"], "hits_q100_db_q100_sensitive.tsv.gz", env={"RSK_DEVICES": fx.device_list(3)})


def test_the_callers_current_device_survives_a_several_device_search(ctx, work):
    """a search over a device list creates helper contexts on other devices; the calling thread's current device
    must be what it was (arrays that belong to a chain set are allocated on the SET's device whatever it is), so a chain
    set used afterwards still works.  With two GPUs the list is "0,1" (distinct devices), on a one-GPU box "0,0"; a
    one-entry list names the device of the call."""
    import numpy as np
    import torch
    import reseek_amd
    two = torch.cuda.device_count() >= 2
    devices = "0,1" if two else "0,0"
    torch.cuda.set_device(0)
    before = torch.cuda.current_device()
    out = os.path.join(work, "dev.tsv")
    n, st = ctx.search(os.path.join(work, "q100.bca"), out, "sensitive", columns=COLS, devices=devices)
    assert table(out) == golden("hits_q100_sensitive.tsv.gz")
    assert torch.cuda.current_device() == before
    # a chain set created and used on the caller's context AFTER that call (its lazily built arrays land on its own device)
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 36, int(L)).astype(np.uint8) for L in rng.integers(20, 300, 64)]
    db = reseek_amd.Db.from_mu_seqs(ctx, seqs)
    o = torch.zeros((64, 64), dtype=torch.int16, device="cuda:0")
    ctx.mu_gapless_matrix_dev(db, db, True, o.data_ptr(), 64)
    ctx.sync()
    import oracle_lib as ol
    ia, ib = np.triu_indices(64)
    assert np.array_equal(o.cpu().numpy().astype(np.uint16)[ia, ib].astype(np.int32), ol.mu_gapless_pairs(seqs, ia, ib))
    db.close()
    # one-entry list: the search runs on that device (device 1 when there is one)
    n2, st2 = ctx.search(os.path.join(work, "q100.bca"), out, "sensitive", columns=COLS, devices="1" if two else "0")
    assert table(out) == golden("hits_q100_sensitive.tsv.gz")

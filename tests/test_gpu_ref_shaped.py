"""SURVEY 8b "signatures to keep".  Three C++ callers over reseek_host.h + librsk.so must reproduce the reference binary's
goldens:
* oracle/_ref/search_refsrc -- the REFERENCE'S OWN search.cpp (SelfSearch / Search_NoMuFilter / cmd_search,
  search.cpp:20-111), compiled unmodified in the build container against the name shim tests/ref_shaped/shim/
  (oracle/Makefile.ref; nothing of it is stored in the repo); the binary travels to the GPU box like oracle/_ref/reseek.
  If DBSearcher / ChainReader2 / MuSeqSource / SeqDB / MuPreFilter / PostMuFilter drift from the reference's names or
  argument lists, that build fails; here its output is checked.
* tests/ref_shaped/search_main.cpp -- our own driver over the same classes (compiled here), also with two device contexts.
* tests/ref_shaped/pair_main.cpp -- the per-pair entry points (SetQuery/SetTarget/AlignQueryTarget, ChainBag + AlignBags,
  MuKmerFilter::SetBagQ/AlignBag)."""
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


def _compile(src, exe):
    cxx = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "g++"
    cmd = [cxx] + (["-x", "c++"] if cxx.endswith("hipcc") else []) + [
        "-std=c++17", "-O1", "-I", os.path.join(ROOT, "reseek_amd", "csrc", "host"), src, "-L", os.path.join(ROOT, "reseek_amd"), "-lrsk",
        "-Wl,-rpath," + os.path.join(ROOT, "reseek_amd"), "-pthread", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


@pytest.fixture(scope="module")
def work():
    d = tempfile.mkdtemp(prefix="rsk_refshaped_")
    for name in ("q100.bca", "palms.bca"):
        with gzip.open(os.path.join(fx.GOLDEN, name + ".gz"), "rb") as f, open(os.path.join(d, name), "wb") as g:
            g.write(f.read())
    _compile(os.path.join(ROOT, "tests", "ref_shaped", "search_main.cpp"), os.path.join(d, "search_main"))
    _compile(os.path.join(ROOT, "tests", "ref_shaped", "pair_main.cpp"), os.path.join(d, "pair_main"))
    yield d
    shutil.rmtree(d, ignore_errors=True)


REFSRC = os.path.join(ROOT, "oracle", "_ref", "search_refsrc")
DRIVERS = ["search_main", "search_refsrc"]


@pytest.fixture(params=DRIVERS)
def driver(request, work):
    if request.param == "search_refsrc":
        if not os.path.exists(REFSRC):
            pytest.skip("oracle/_ref/search_refsrc not built (make -f oracle/Makefile.ref where /root/reference exists)")
        return REFSRC
    return os.path.join(work, request.param)


def _run(work, args, golden, exe=None, env=None):
    out = os.path.join(work, "hits.tsv")
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([exe or os.path.join(work, "search_main")] + args + ["-output", out, "-columns", COLS], capture_output=True, text=True, cwd=work,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    got = sorted(open(out).read().splitlines())
    want = sorted("\t".join(x) for x in fx.read_tsv(golden))
    assert got == want, "hit tables differ: %d vs %d rows" % (len(got), len(want))


def test_selfsearch_sequence(work, driver):
    _run(work, ["q100.bca", "-sensitive"], "hits_q100_sensitive.tsv.gz", driver)


def test_selfsearch_long_chains(work, driver):
    _run(work, ["palms.bca", "-sensitive"], "hits_palms_sensitive.tsv.gz", driver)


def test_search_nomufilter_streams_a_chainreader2(work, driver):
    _run(work, ["q100.bca", "-db", "q100.bca", "-sensitive"], "hits_q100_db_q100_sensitive.tsv.gz", driver)


def test_cmd_search_fast_db_two_stage(work, driver):
    """MuSeqSource / SeqDB / MuPreFilter / PostMuFilter with the argument lists of search.cpp:9-18; the hand-off file
    must be the reference's byte for byte (-keeptmp)."""
    _run(work, ["q100.bca", "-db", "q100.bca", "-fast", "-keeptmp"], "hits_q100_db_q100_fast.tsv.gz", driver)
    tmp = open(os.path.join(work, "hits.tsv.prefilter.tmp")).read()
    with gzip.open(os.path.join(fx.GOLDEN, "prefilter_q100_db_q100_fast_tmp.tsv.gz"), "rt") as f:
        assert tmp == f.read()


def test_per_pair_entry_points(work):
    """pair_main: every pair (i <= j) of the first 24 chains of q100 plus the long-chain set through
    DSSAligner::AlignQueryTarget and again through ChainBag + AlignBags; both must reproduce the golden rows."""
    r = subprocess.run([os.path.join(work, "pair_main"), "q100.bca", "24", "palms.bca", "6"], capture_output=True, text=True, cwd=work)
    assert r.returncode == 0, r.stderr
    rows = [ln.split("\t") for ln in r.stdout.splitlines() if ln and not ln.startswith("#")]
    gold = {}
    for name in ("hits_q100_sensitive.tsv.gz", "hits_palms_sensitive.tsv.gz"):
        for x in fx.read_tsv(name):
            gold[(x[0], x[1])] = x
    assert rows, "pair_main printed nothing"
    n_hit = 0
    for x in rows:
        how, line = x[0], x[1:]
        key = (line[0], line[1])
        if key in gold:            # E <= 10 rows of the golden table must match to the last column
            assert line == gold[key], (how, line, gold[key])
            n_hit += 1
    assert n_hit >= 40
    # the two forms print the same rows
    aq = sorted(tuple(x[1:]) for x in rows if x[0] == "AlignQueryTarget")
    ab = sorted(tuple(x[1:]) for x in rows if x[0] == "AlignBags")
    assert aq == ab and aq


def test_per_pair_entry_points_from_several_threads(work):
    """The reference's threading model on the per-pair entry points (one DSSAligner per thread, rows dealt through a shared
    counter: dbsearcher.cpp:98-106, runself.cpp:72-99) with all aligners on ONE device context: the host layer serialises
    their batch-of-one calls per context (an rsk_ctx is not thread-safe; MuKmerFilter::Align used to seed on the
    default context whatever its aligner's).  4 threads print exactly the rows of the one-thread run -- long-chain pairs
    (MKF seeding, X-drop, chaining) included."""
    one = subprocess.run([os.path.join(work, "pair_main"), "q100.bca", "16", "palms.bca", "6"], capture_output=True, text=True, cwd=work)
    assert one.returncode == 0, one.stderr
    want = sorted(ln for ln in one.stdout.splitlines() if ln.startswith("AlignQueryTarget\t"))
    four = subprocess.run([os.path.join(work, "pair_main"), "-threads", "4", "q100.bca", "16", "palms.bca", "6"], capture_output=True, text=True, cwd=work)
    assert four.returncode == 0, four.stderr
    got = sorted(ln for ln in four.stdout.splitlines() if ln.startswith("AlignQueryTarget\t"))
    assert got == want and len(got) >= 30

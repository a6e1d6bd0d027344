#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (lives under tests/ because it executes oracle/_ref/reseek).
End-to-end check against the reference BINARY on a synthetic structure set (GPU box; needs oracle/_ref/reseek, which
travels with the snapshot): writes an N-chain .bca (tools/bench_search.py generator), runs
    oracle/_ref/reseek -search syn.bca -<mode> -output ref.tsv -threads T
and rsk_search on the same file, compares the sorted hit tables line by line and prints both wall times.
usage: python tests/compare_with_reference.py [nchains=1000] [mode=sensitive] [db_chains=0] [threads=1]   (db_chains > 0: -search Q -db DB)

threads = 1 is the default on purpose: with several threads the reference binary is not reproducible on sets with
long-chain (MKF) pairs -- two 16-thread runs of the same command gave 44,193 and 44,195 rows on a 3000-chain set, the
1-thread run 44,189 rows, identical to ours.  (Its banded X-drop traceback reads trace cells next to the path that
the DP never wrote; XDPMem's matrix comes from malloc without clearing, xdpmem.h:96-108 / mx.h:38-54, so the outcome
depends on what the allocator hands back, i.e. on the thread's history.  Here such cells read as 0, which is what a
fresh process sees.)  `threads = 0` uses all usable CPUs (timing comparisons)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))       # bench_search.py: the synthetic .bca writer
import bench  # noqa: E402
import bench_search  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "reseek")
GOLDENS = os.path.join(ROOT, "tests", "golden", "refbin_goldens.json")


def case_key(n, mode, ndb, seed, long_chains, tail):
    return "n%d_%s_db%d_seed%d_long%d_tail%d" % (n, mode, ndb, seed, long_chains, int(bool(tail)))


def write_inputs(td, n, ndb, seed, long_chains, tail):
    """the seeded synthetic .bca file(s) of a case -> (query path, db path or None)"""
    lens = bench.scop40_lengths()
    rng = np.random.default_rng(seed)
    q = os.path.join(td, "q.bca")
    ql = lens[rng.choice(len(lens), n)].copy()
    ql[:long_chains] = [620 + 140 * k for k in range(long_chains)]      # chains that take the MKF / X-drop path for sure
    dbl = lens[rng.choice(len(lens), ndb)].copy() if ndb else None
    if tail:
        # PDB-like lengths: lognormal, median ~250, tail to 5,000 (BASELINE configs[3] / configs[4])
        ql = np.clip(rng.lognormal(np.log(250), 0.6, n), 30, 1500).astype(np.int64)
        if ndb:
            dbl = np.clip(rng.lognormal(np.log(250), 0.75, ndb), 20, 5000).astype(np.int64)
            dbl[:3] = [5000, 2600, 1100]
    bench_search.write_bca(q, ql, rng)
    db = None
    if ndb:
        db = os.path.join(td, "db.bca")
        bench_search.write_bca(db, dbl, rng)
    return q, db


def file_md5(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def table_md5(rows):
    import hashlib
    return hashlib.md5(("\n".join(rows) + "\n").encode()).hexdigest()


def run_reference(td, q, db, mode, threads=1):
    """oracle/_ref/reseek -search on the case's files -> (sorted rows, seconds)"""
    ref_tsv = os.path.join(td, "ref.tsv")
    cmd = [REF, "-search", q, "-" + mode, "-output", ref_tsv, "-threads", str(threads)] + (["-db", db] if db else [])
    t0 = time.perf_counter()
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
    return sorted(open(ref_tsv).read().splitlines()), time.perf_counter() - t0


def compare(n, mode, ndb=0, threads=1, seed=21, keep=None, long_chains=0, tail=False, use_binary=None):
    """-> dict (see main); ["identical"] tells whether the sorted hit tables are equal.  The reference side is the BINARY run here
    when it travelled (use_binary None / True) and, for the cases tests/golden/make_refbin_goldens.py recorded (its one-thread table
    of the same seeded files: row count + md5 of the sorted table, md5 of the input files), ALSO the committed golden -- so the
    case still checks our table where the binary is absent (use_binary False forces that route)."""
    import torch
    import reseek_amd
    have_bin = os.path.exists(REF) if use_binary is None else bool(use_binary)
    golden = None
    try:
        with open(GOLDENS) as f:
            golden = json.load(f).get(case_key(n, mode, ndb, seed, long_chains, tail))
    except (OSError, ValueError):
        pass
    if not have_bin and golden is None:
        raise FileNotFoundError("oracle/_ref/reseek is missing and the case has no committed golden (make -f oracle/Makefile.ref where /root/reference exists)")
    if threads <= 0:
        threads = bench.usable_cpus()
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        with tempfile.TemporaryDirectory() as td:
            q, db = write_inputs(td, n, ndb, seed, long_chains, tail)
            our_tsv, ref_tsv = os.path.join(td, "our.tsv"), os.path.join(td, "ref.tsv")
            a, t_ref = run_reference(td, q, db, mode, threads) if have_bin else (None, float("nan"))
            ctx.search_rskdb(q, our_tsv, mode, db=db)            # warm-up (HIP module load, allocator)
            t0 = time.perf_counter()
            nhits, st = ctx.search_rskdb(q, our_tsv, mode, db=db)
            t_our = time.perf_counter() - t0
            b = sorted(open(our_tsv).read().splitlines())
            res = {"chains": n, "db_chains": ndb, "mode": mode, "pairs": int(st[0]), "long_chain_pairs": int(st[4]), "reference_threads": threads,
                   "reference_seconds": t_ref, "our_seconds": t_our, "speedup": t_ref / t_our, "reference_rows": len(a) if a is not None else golden["rows"],
                   "our_rows": len(b), "identical": a == b if a is not None else True, "reference_binary_ran": a is not None}
            if golden is not None:
                # the inputs are regenerated from the seed: they must be the files the golden's reference run saw
                res["golden"] = {"inputs_match": file_md5(q) == golden["q_md5"] and (db is None or file_md5(db) == golden["db_md5"]),
                                 "rows_match": len(b) == golden["rows"], "md5_match": table_md5(b) == golden["sorted_md5"]}
                res["identical"] = res["identical"] and all(res["golden"].values())
            if a is None:
                a = b if res["identical"] else []
            if a != b:
                sa, sb = set(a), set(b)
                res["only_reference"] = sorted(sa - sb)[:12]
                res["only_ours"] = sorted(sb - sa)[:12]
                res["n_only_reference"], res["n_only_ours"] = len(sa - sb), len(sb - sa)
                if keep:
                    import shutil
                    os.makedirs(keep, exist_ok=True)
                    for f in (q, ref_tsv, our_tsv) + ((db,) if db else ()):
                        shutil.copy(f, keep)
            return res
    finally:
        ctx.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    mode = sys.argv[2] if len(sys.argv) > 2 else "sensitive"
    ndb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    threads = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    res = compare(n, mode, ndb, threads, seed=int(os.environ.get("RSK_COMPARE_SEED", "21")), keep=os.environ.get("RSK_COMPARE_KEEP"),
                  long_chains=int(os.environ.get("RSK_COMPARE_LONG", "0")), tail=os.environ.get("RSK_COMPARE_TAIL", "0") == "1")
    print(json.dumps(res, indent=1))
    if not res["identical"]:
        sys.exit(1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (executes oracle/_ref/reseek; build container only).

Full-size hit-table golden: BASELINE's "identical hit tables on SCOP40 all-vs-all" as data.  Writes the seeded
11,211-chain synthetic .bca (tools/bench_search.py generator, numpy seed 7 -- the set `bench_search.py 0 <mode> bca`
and bench.py's search_bca leg use), runs

    oracle/_ref/reseek -search syn11211.bca -<mode> -output ref.tsv -threads 1

(`search.cpp:20` SelfSearch; one thread because the reference is not reproducible on long-chain pairs with several,
DESIGN section 5) and stores the row count and the md5 of the sorted table under tests/golden/ as
full11211_<mode>[_<tag>].md5.txt.  `tests/test_gpu_full_golden.py` regenerates the same .bca on the GPU box (the md5 of
the .bca is part of the golden, so a generator drift is told apart from a search difference) and requires rsk_search to
reproduce the md5.

A run that was interrupted still pins a well-defined part of the table: with one thread the reference walks the pairs
row-major (GetNextPairSelf runself.cpp:72-99: i, then j >= i) and writes both orientations of a hit at once, so the file is
complete for every pair whose smaller chain index is below the row it was working on.  `--prefix-from FILE` stores the row
count and md5 of exactly those rows (full11211_<mode>_prefix.md5.txt, "rows_with_min_index_below").

usage: make_full_golden.py MODE [--perturb] [--chains N] [--workdir DIR] [--prefix-from PARTIAL.tsv]
  --perturb  run the reference under glibc's MALLOC_PERTURB_=255 (malloc'ed memory reads 0): what a trace cell the
             banded X-drop never wrote holds is then defined (xdpmem.h:96-108 allocates without clearing).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def synth_bca(path, chains=0, seed=7):
    """The seeded set: all 11,211 SCOP40 lengths in file order (chains = 0) or a seeded choice of `chains` of them."""
    import bench
    import bench_search
    lens = bench.scop40_lengths()
    rng = np.random.default_rng(seed)
    if chains:
        lens = lens[rng.choice(len(lens), chains, replace=chains > len(lens))]
    bench_search.write_bca(path, lens, rng)
    return len(lens)


def file_md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def table_md5(path):
    """md5 over the sorted lines (each + '\n') and the row count."""
    with open(path, "rb") as f:
        lines = f.read().splitlines()
    lines.sort()
    h = hashlib.md5()
    for ln in lines:
        h.update(ln + b"\n")
    return h.hexdigest(), len(lines)


def chain_index(label):
    return int(label[3:])                   # "syn01234"


def prefix_md5(lines, cut):
    """md5 over the sorted rows whose two chains' smaller index is < cut (bytes lines without newline)"""
    keep = []
    for ln in lines:
        f = ln.split(b"\t")
        if len(f) >= 2 and min(chain_index(f[0].decode()), chain_index(f[1].decode())) < cut:
            keep.append(ln)
    keep.sort()
    h = hashlib.md5()
    for ln in keep:
        h.update(ln + b"\n")
    return h.hexdigest(), len(keep)


def prefix_golden(mode, partial, bca_md5):
    data = open(partial, "rb").read()
    lines = data.split(b"\n")[:-1]           # whatever follows the last newline is a torn line
    last = lines[-1].split(b"\t")
    cut = min(chain_index(last[0].decode()), chain_index(last[1].decode()))      # the row in progress: everything below it is complete
    md5, rows = prefix_md5(lines, cut)
    rec = {"chains": 11211, "mode": mode, "bca_md5": bca_md5, "rows_with_min_index_below": cut, "rows": rows, "sorted_rows_md5": md5,
           "reference_threads": 1, "command": "reseek -search syn11211.bca -%s -output ref.tsv -threads 1 (interrupted; complete for the rows counted here)" % mode}
    out = os.path.join(ROOT, "tests", "golden", "full11211_%s_prefix.md5.txt" % mode)
    with open(out, "w") as f:
        f.write(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode")
    ap.add_argument("--perturb", action="store_true")
    ap.add_argument("--chains", type=int, default=0)
    ap.add_argument("--workdir", default="/tmp/full_golden")
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--prefix-from", default="")
    a = ap.parse_args()
    os.makedirs(a.workdir, exist_ok=True)
    if a.prefix_from:
        bca = os.path.join(a.workdir, "syn11211.bca")
        return prefix_golden(a.mode, a.prefix_from, file_md5(bca))
    n = a.chains or 11211
    bca = os.path.join(a.workdir, "syn%d.bca" % n)
    if not os.path.exists(bca):
        synth_bca(bca + ".tmp%d" % os.getpid(), a.chains)
        os.replace(bca + ".tmp%d" % os.getpid(), bca)
    tag = a.mode + ("_perturb" if a.perturb else "") + ("_t%d" % a.threads if a.threads != 1 else "")
    tsv = os.path.join(a.workdir, "ref_%d_%s.tsv" % (n, tag))
    env = dict(os.environ)
    if a.perturb:
        env["MALLOC_PERTURB_"] = "255"
    t0 = time.time()
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "reseek"), "-search", bca, "-" + a.mode, "-output", tsv,
                    "-threads", str(a.threads)], check=True, env=env, cwd=a.workdir, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    md5, rows = table_md5(tsv)
    rec = {"chains": n, "mode": a.mode, "bca_md5": file_md5(bca), "rows": rows, "sorted_table_md5": md5,
           "reference_seconds": round(dt, 1), "reference_threads": a.threads, "malloc_perturb": 255 if a.perturb else None,
           "command": "reseek -search syn%d.bca -%s -output ref.tsv -threads %d" % (n, a.mode, a.threads)}
    out = os.path.join(ROOT, "tests", "golden", "full%d_%s.md5.txt" % (n, tag))
    with open(out, "w") as f:
        f.write(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()

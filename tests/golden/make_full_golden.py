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


# ---- `-search Q.bca -db Q.bca -fast` (BASELINE configs[2]; search.cpp:76-111: MuPreFilter -> hand-off file -> PostMuFilter) ----
# The literal one-thread command needs 3-4 CPU-hours on the 11,211-chain set, so both stages are cut into target ranges for
# several ONE-thread processes of the reference's own code (one thread each: the reference's bags and its long-chain
# alignments are only reproducible with one, SURVEY 0.6 / DESIGN 5):
#   stage 1  oracle/_ref/ref_harness prefrange (the reference's PrefilterMu::Search per target, a bag that never truncates)
#            per range -> all triples; ref_harness rsbreplay feeds them to the reference's RankedScoresBag in target order
#            -> the hand-off file (what `-keeptmp` keeps);
#   stage 2  `reseek -postmufilter Q.bca -db Q.bca -filin <piece of the hand-off file> -threads 1` (cmd_postmufilter
#            postmufilter.cpp:303 = the PostMuFilter call of cmd_search with the same DM_AlwaysSensitive preset) per piece of
#            the hand-off file's target lines; a candidate pair's alignment does not depend on the other lines.
# `--validate` runs this route AND the literal command (`-search -db -fast -keeptmp -threads 1`) on a sample and requires
# the same hand-off bytes and the same sorted hit table, also with bags that overflow (-rsb_size 20).
def _run(cmd, cwd, env=None):
    subprocess.run(cmd, check=True, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _pool(jobs, procs):
    """jobs = [(done_marker, cmd, cwd)]: runs those without a marker, `procs` at a time; a marker is written when a job ends
    with status 0 (a run that is killed resumes with the pieces that are left)."""
    todo = [j for j in jobs if not os.path.exists(j[0])]
    running = []
    while todo or running:
        while todo and len(running) < procs:
            j = todo.pop(0)
            running.append((j, subprocess.Popen(j[1], cwd=j[2], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        time.sleep(0.2)
        still = []
        for j, p in running:
            rc = p.poll()
            if rc is None:
                still.append((j, p))
            elif rc != 0:
                raise SystemExit("failed: " + " ".join(j[1]))
            else:
                open(j[0], "w").close()
        running = still


def fastdb_split_route(bca, workdir, nchains, procs, rsb_size=None, tag="split", pieces=None):
    ref = os.path.join(ROOT, "oracle", "_ref", "reseek")
    har = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    pieces = pieces or procs
    bounds = [nchains * r // pieces for r in range(pieces + 1)]
    t0 = time.time()
    bins = [os.path.join(workdir, "%s_tri_%d.bin" % (tag, r)) for r in range(pieces)]
    _pool([(bins[r] + ".done", [har, "prefrange", bca, bca, str(bounds[r]), str(bounds[r + 1]), bins[r], "--", "-fast"], workdir)      # cmd_search's Params: DM_UseCommandLineOption under -fast
           for r in range(pieces)], procs)
    t1 = time.time()
    handoff = os.path.join(workdir, "%s_handoff.tsv" % tag)
    _run([har, "rsbreplay", str(nchains), str(rsb_size or 1500), handoff] + bins, workdir)
    # stage 2: the target lines dealt round-robin (long and short targets in every piece)
    lines = open(handoff, "rb").read().split(b"\n")
    assert lines[0].startswith(b"prefilter\t") and lines[-1] == b""
    body = lines[1:-1]
    parts, jobs = [], []
    for r in range(pieces):
        mine = body[r::pieces]
        fn = os.path.join(workdir, "%s_handoff_%d.tsv" % (tag, r))
        out = os.path.join(workdir, "%s_hits_%d.tsv" % (tag, r))
        parts.append(out)
        if not mine:
            open(out, "w").close()
            continue
        if not os.path.exists(out + ".done"):
            with open(fn, "wb") as f:
                f.write(b"prefilter\t%d\n" % len(mine) + b"".join(x + b"\n" for x in mine))
        jobs.append((out + ".done", [ref, "-postmufilter", bca, "-db", bca, "-filin", fn, "-output", out, "-dbsize", str(nchains), "-threads", "1"], workdir))
    _pool(jobs, procs)
    t2 = time.time()
    hits = os.path.join(workdir, "%s_hits.tsv" % tag)
    with open(hits, "wb") as f:
        for out in parts:
            f.write(open(out, "rb").read())
    return handoff, hits, {"stage1_wall_s_this_run": round(t1 - t0, 1), "stage2_wall_s_this_run": round(t2 - t1, 1), "processes": procs, "pieces": pieces}


def fastdb_validate(workdir):
    ref = os.path.join(ROOT, "oracle", "_ref", "reseek")
    ok = True
    for n, rsb in ((300, None), (300, 20)):
        bca = os.path.join(workdir, "val%d.bca" % n)
        if not os.path.exists(bca):
            synth_bca(bca, n)
        tag = "val%d_%s" % (n, rsb or "std")
        lit = os.path.join(workdir, tag + "_literal.tsv")
        log = os.path.join(workdir, tag + "_literal.log")
        cmd = [ref, "-search", bca, "-db", bca, "-fast", "-keeptmp", "-threads", "1", "-output", lit, "-log", log]
        if rsb:
            cmd += ["-rsb_size", str(rsb)]
        _run(cmd, workdir)
        tmpfn = [ln.split("=", 1)[1].strip() for ln in open(log) if ln.startswith("MuFilterTsvFN=")][0]
        lit_handoff = open(tmpfn, "rb").read()
        os.remove(tmpfn)
        for f in os.listdir(workdir):
            if f.startswith(tag + "_split"):
                os.remove(os.path.join(workdir, f))
        handoff, hits, _ = fastdb_split_route(bca, workdir, n, 3, rsb, tag + "_split")
        same_h = open(handoff, "rb").read() == lit_handoff
        same_t = table_md5(hits) == table_md5(lit)
        print("validate %d chains rsb_size %s: hand-off %s (%d bytes), hit table %s (%d rows)" %
              (n, rsb or 1500, "identical" if same_h else "DIFFERENT", len(lit_handoff), "identical" if same_t else "DIFFERENT", table_md5(lit)[1]))
        ok = ok and same_h and same_t
    return ok


def fastdb_golden(workdir, procs):
    bca = os.path.join(workdir, "syn11211.bca")
    if not os.path.exists(bca):
        synth_bca(bca + ".tmp%d" % os.getpid(), 0)
        os.replace(bca + ".tmp%d" % os.getpid(), bca)
    handoff, hits, tm = fastdb_split_route(bca, workdir, 11211, procs, None, "fastdb", pieces=96)
    md5, rows = table_md5(hits)
    rec = {"chains": 11211, "mode": "fast", "db": "the same file (-search Q.bca -db Q.bca -fast)", "bca_md5": file_md5(bca), "rows": rows,
           "sorted_table_md5": md5, "handoff_md5": file_md5(handoff), "handoff_bytes": os.path.getsize(handoff),
           "handoff_target_lines": int(open(handoff, "rb").readline().split(b"\t")[1]),
           "candidates": sum(int(ln.split(b"\t", 2)[1]) for ln in open(handoff, "rb").read().split(b"\n")[1:-1]),
           "reference_threads": 1, "route": "96 target ranges, %d one-thread processes at a time, of the reference's code (ref_harness prefrange / rsbreplay, "
           "reseek -postmufilter); validated against the literal `reseek -search Q -db Q -fast -keeptmp -threads 1` on samples "
           "(make_full_golden.py --fastdb --validate)" % procs, **tm}
    out = os.path.join(ROOT, "tests", "golden", "full11211_fastdb.md5.txt")
    with open(out, "w") as f:
        f.write(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))


def main():
    if "--fastdb" in sys.argv:
        ap = argparse.ArgumentParser()
        ap.add_argument("--fastdb", action="store_true")
        ap.add_argument("--validate", action="store_true")
        ap.add_argument("--procs", type=int, default=7)
        ap.add_argument("--workdir", default="/tmp/full_golden")
        a = ap.parse_args()
        os.makedirs(a.workdir, exist_ok=True)
        if a.validate:
            sys.exit(0 if fastdb_validate(a.workdir) else 1)
        return fastdb_golden(a.workdir, a.procs)
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

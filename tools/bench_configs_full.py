#!/usr/bin/env python3
"""BASELINE configs[3] and configs[4] at their STATED size on ONE GPU, as 8 sequential shards (what 8 GPUs run side by side):

  config3  256 synthetic queries x 1,000,000-chain synthetic .bca DB, -sensitive
  config4  1,000 queries x 700,000-chain .bca DB, -verysensitive (every pair is a hit row: 7e8 rows, ~30 GB of TSV -- the hit
           lines go to rsk_search_opts.hits_digest, an order-independent digest, instead of a file)

For each: the DB is written once (tools/bench_search.py write_bca_fast, seeded, SCOP40 length distribution, in a child process
so that the generator's arrays do not count as the search's memory), then ONE child process per shard count S in (8, 3, 1)
runs shards 0 .. S-1 of `rsk_search` (shard_index / shard_count: contiguous DB ranges balanced by residues, the reader
streams only its range, runquery.cpp:82-125) one after the other and reports per shard: seconds, pairs, hits, digest; per
process: peak host RSS (VmHWM) and peak device memory in use (hipMemGetInfo sampled every 50 ms by a thread).  Checked here:
the digests of the 8 shards combine to the digests of the 3-shard and the 1-shard run (same multiset of hit lines), pairs and
hits add up; imbalance = max / mean of the shard seconds.  Reference equality: oracle/_ref/reseek -threads 1 on a sample of the
same files that yields >= 10,000 rows (config3: all 256 queries x every 100th DB chain; config4: 16 queries x 700 chains).

  python tools/bench_configs_full.py [config3] [config4] [--scale F] > profiles/r04_configs_full.json
(--scale 0.01 for a quick run of the machinery.)  Prints one JSON object."""
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

CONFIGS = {"config3": (256, 1_000_000, "sensitive", 11), "config4": (1000, 700_000, "verysensitive", 12)}


def child_generate(path, n, seed, prefix):
    import bench
    import bench_search
    lens = bench.scop40_lengths()
    rng = np.random.default_rng(seed)
    bench_search.write_bca_fast(path, lens[rng.choice(len(lens), n)], rng, prefix)


def vm_hwm_kb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmHWM:"):
            return int(ln.split()[1])
    return 0


def child_shards(q, db, mode, nshards, digest):
    """runs the shards of one shard count in THIS process; prints a JSON line"""
    import torch
    import reseek_amd
    from reseek_amd import capi
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    free0, total = torch.cuda.mem_get_info()
    peak = [0]
    stop = threading.Event()

    def sample():
        while not stop.is_set():
            f, _ = torch.cuda.mem_get_info()
            peak[0] = max(peak[0], total - f)
            time.sleep(0.05)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    shards = []
    with tempfile.TemporaryDirectory() as td:
        for k in range(nshards):
            out = os.path.join(td, "hits_%d.tsv" % k)
            t0 = time.perf_counter()
            nh, st = ctx.search(q, out, mode, db=db, shard_index=k, shard_count=nshards, hits_digest=1 if digest else 0)
            dt = time.perf_counter() - t0
            if digest:
                d = capi.read_hits_digest(out)
            else:
                # small tables: digest of the file through the same definition (count, bytes, and python-side md5 of the sorted lines)
                import hashlib
                lines = open(out, "rb").read().splitlines()
                d = (len(lines), sum(len(x) + 1 for x in lines), int(hashlib.md5(b"\n".join(sorted(lines))).hexdigest()[:16], 16), 0)
                d = d + (sorted(lines),)
            shards.append({"shard": k, "seconds": dt, "pairs": int(st[0]), "sw_pairs": int(st[5]), "long_chain_pairs": int(st[4]), "hits": int(nh),
                           "digest": list(d[:4]), "_lines": d[4] if len(d) > 4 else None})
            os.remove(out)
    stop.set()
    th.join()
    ctx.close()
    res = {"nshards": nshards, "shards": shards, "peak_host_rss_gb": vm_hwm_kb() / 1048576.0, "peak_device_bytes_in_use_gb": peak[0] / 2**30,
           "device_bytes_in_use_before_gb": (total - free0) / 2**30}
    if not digest:
        # text route: the union's sorted-line md5
        import hashlib
        allines = sorted(x for s in shards for x in s["_lines"])
        res["union_sorted_md5"] = hashlib.md5(b"\n".join(allines)).hexdigest()
        res["union_rows"] = len(allines)
    for s in shards:
        s.pop("_lines", None)
    print("RESULT " + json.dumps(res), flush=True)


def run_child(args):
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + args, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    if r.returncode != 0:
        raise SystemExit("child %s failed:\n%s\n%s" % (args, r.stdout[-2000:], r.stderr[-4000:]))
    for ln in r.stdout.splitlines():
        if ln.startswith("RESULT "):
            return json.loads(ln[7:])
    return None


def reference_sample(td, q, db, mode, nq_s, step, tag):
    """oracle/_ref/reseek -threads 1 vs rsk_search on the first nq_s queries x every step-th DB chain"""
    import torch
    import reseek_amd
    import bench_search
    import make_tail_bca
    ref = os.path.join(ROOT, "oracle", "_ref", "reseek")
    if not os.path.exists(ref):
        return {"identical": None, "note": "oracle/_ref/reseek did not travel"}
    qs, dbs = os.path.join(td, tag + "_qs.bca"), os.path.join(td, tag + "_dbs.bca")
    rq, lq = make_tail_bca.read_bca(q)
    bench_search.write_bca_records(qs, rq[:nq_s], labels=lq[:nq_s])
    # sample of the DB: the raw records of every step-th chain, located through the file's length table (the file has 10^6
    # chains: it is not parsed as a whole)
    import struct
    with open(db, "rb") as f:
        magic, nchains, pos, lab_bytes = struct.unpack("<IQQQ", f.read(28))
        assert magic == 0xBCABCA
        f.seek(pos)
        lens = np.frombuffer(f.read(4 * nchains), np.uint32).astype(np.int64)
        all_labels = f.read(lab_bytes).split(b"\0")
        starts = 28 + 7 * np.concatenate([[0], np.cumsum(lens)[:-1]])
        idx = list(range(0, nchains, step))
        recs, labels = [], []
        for i in idx:
            L = int(lens[i])
            f.seek(int(starts[i]))
            raw = f.read(7 * L)
            recs.append((raw[:L], raw[L:], L))
            labels.append(all_labels[i].decode())
    bench_search.write_bca_records(dbs, recs, labels=labels)
    ours = os.path.join(td, tag + "_ours.tsv")
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.search(qs, ours, mode, db=dbs)
    ctx.close()
    # the reference on the same sample: ONE thread per process (reproducible), the DB sample dealt to as many processes as
    # the box has CPUs for us (a pair's hit line does not depend on the other DB chains)
    import bench
    P = max(1, min(bench.usable_cpus(), 16, len(recs) // 50 + 1))
    t0 = time.perf_counter()
    procs, parts = [], []
    for k in range(P):
        piece = os.path.join(td, "%s_dbs_%d.bca" % (tag, k))
        bench_search.write_bca_records(piece, recs[k::P], labels=labels[k::P])
        part = os.path.join(td, "%s_ref_%d.tsv" % (tag, k))
        parts.append(part)
        procs.append(subprocess.Popen([ref, "-search", qs, "-db", piece, "-" + mode, "-output", part, "-threads", "1"], stdout=subprocess.DEVNULL,
                                      stderr=subprocess.DEVNULL, cwd=td))
    for pr in procs:
        if pr.wait(timeout=3000) != 0:
            raise SystemExit("reference run failed")
    tr = time.perf_counter() - t0
    theirs = os.path.join(td, tag + "_ref.tsv")
    with open(theirs, "w") as f:
        for part in parts:
            f.write(open(part).read())
    a, b = sorted(open(theirs).read().splitlines()), sorted(open(ours).read().splitlines())
    return {"identical": a == b, "rows": len(a), "sample": "%d queries x %d DB chains (every %d-th); reference: %d one-thread processes over slices of the sample, %.0f s" % (nq_s, len(idx), step, P, tr)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child-generate":
        return child_generate(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    if len(sys.argv) > 1 and sys.argv[1] == "--child-shards":
        return child_shards(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6] == "1")
    which = [a for a in sys.argv[1:] if a in CONFIGS] or ["config3", "config4"]
    scale = float(sys.argv[sys.argv.index("--scale") + 1]) if "--scale" in sys.argv else 1.0
    nosample = "--no-reference" in sys.argv
    from reseek_amd import capi
    out = {"what": __doc__.split("\n\n")[0], "scale": scale}
    workroot = os.environ.get("RSK_BENCH_TMP", tempfile.gettempdir())
    with tempfile.TemporaryDirectory(dir=workroot) as td:
        for key in which:
            nq, ndb, mode, seed = CONFIGS[key]
            ndb = max(64, int(ndb * scale))
            q, db = os.path.join(td, key + "_q.bca"), os.path.join(td, key + "_db.bca")
            t0 = time.perf_counter()
            run_child(["--child-generate", q, str(nq), str(seed), "q"])
            run_child(["--child-generate", db, str(ndb), str(seed + 100), "d"])
            tgen = time.perf_counter() - t0
            digest = mode == "verysensitive"
            e = {"what": "-search Q -db DB -%s, %d queries x %d-chain .bca DB (DSS featurisation + self-rev of every DB chain inside the calls)" % (mode, nq, ndb),
                 "db_file_gb": os.path.getsize(db) / 2**30, "generation_seconds": tgen, "hit_lines": "digest (hits_digest)" if digest else "files"}
            runs = {}
            for S in [int(x) for x in os.environ.get("RSK_CFGFULL_SHARDS", "8,3,1").split(",")]:
                r = run_child(["--child-shards", q, db, mode, str(S), "1" if digest else "0"])
                secs = [s["seconds"] for s in r["shards"]]
                r["seconds_total"] = float(sum(secs))
                r["seconds_max_shard"] = float(max(secs))
                r["imbalance_max_over_mean"] = float(max(secs) / (sum(secs) / len(secs)))
                r["pairs"] = int(sum(s["pairs"] for s in r["shards"]))
                r["hits"] = int(sum(s["hits"] for s in r["shards"]))
                r["chain_pairs_per_sec_one_gpu"] = r["pairs"] / r["seconds_total"]
                r["chain_pairs_per_sec_if_shards_ran_side_by_side"] = r["pairs"] / r["seconds_max_shard"]
                if digest:
                    r["union_digest"] = list(capi.combine_hits_digests([tuple(s["digest"]) for s in r["shards"]]))
                runs["shards_%d" % S] = r
            e.update(runs)
            if digest:
                e["union_8_equals_3_equals_1"] = runs["shards_8"]["union_digest"] == runs["shards_3"]["union_digest"] == runs["shards_1"]["union_digest"]
            else:
                e["union_8_equals_3_equals_1"] = (runs["shards_8"]["union_sorted_md5"] == runs["shards_3"]["union_sorted_md5"] == runs["shards_1"]["union_sorted_md5"]
                                                  and runs["shards_8"]["union_rows"] == runs["shards_1"]["union_rows"])
            e["pairs_and_hits_add_up"] = (runs["shards_8"]["pairs"] == runs["shards_3"]["pairs"] == runs["shards_1"]["pairs"] == nq * ndb and
                                          runs["shards_8"]["hits"] == runs["shards_3"]["hits"] == runs["shards_1"]["hits"])
            if not nosample:
                if key == "config3":
                    e["vs_reference_on_sample"] = reference_sample(td, q, db, mode, nq, max(1, int(100 * min(1.0, scale * 8))), key)
                else:
                    e["vs_reference_on_sample"] = reference_sample(td, q, db, mode, 16, max(1, ndb // 700), key)
            out[key] = e
            os.remove(db)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""End-to-end `-search` timing (SURVEY 8d metric ii: chain-pairs/s incl. filtered pairs) on a synthetic
SCOP40-shaped structure set: Mu letters as bench.py, profile bytes iid, CA random walk, self-rev 0.
Writes an .rskdb container, then runs rsk_search_rskdb (load + upload + Mu filter + SW/traceback/LDDT +
MKF on the host threads + hit replay + TSV).  Not the driver's bench line."""
import json
import os
import struct
import sys
import tempfile
import time

import numpy as np
import torch
from scipy.signal import lfilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import reseek_amd  # noqa: E402


def write_rskdb(path, seqs, rng):
    with open(path, "wb") as f:
        f.write(b"RSKDB1\0\0" + struct.pack("<II", len(seqs), 8))
        for k, mu in enumerate(seqs):
            L = len(mu)
            label = ("syn%05d" % k).encode()
            f.write(struct.pack("<II", L, len(label)) + label)
            aa = rng.integers(0, 20, L)
            f.write(bytes(b"ACDEFGHIKLMNPQRSTVWY"[int(a)] for a in aa))
            f.write(mu.astype(np.uint8).tobytes())
            prof = np.concatenate([aa[None, :], rng.integers(0, 16, (7, L))]).astype(np.uint8)
            f.write(prof.tobytes())
            xyz = np.cumsum(rng.normal(0, 2.2, (3, L)), axis=1).astype(np.float32)
            f.write(xyz.tobytes())
            f.write(struct.pack("<f", 0.0))
            km = (mu[:-2].astype(np.uint32) * 36 + mu[1:-1]) * 36 + mu[2:] if L >= 3 else np.zeros(0, np.uint32)
            f.write(struct.pack("<I", len(km)) + km.astype(np.uint32).tobytes())


def gen_bca_chains(lens, rng):
    """Synthetic chains for a .bca file: CA traces = random walks with 3.8 A steps whose direction persists (helix-like /
    strand-like stretches), amino acids iid.  -> list of (amino-acid bytes, interleaved uint16 x,y,z bytes, L)."""
    recs = []
    for L in lens:
        L = int(L)
        aa = rng.integers(0, 20, L)
        seq = bytes(b"ACDEFGHIKLMNPQRSTVWY"[int(a)] for a in aa)
        d = lfilter([0.6], [1.0, -0.8], rng.normal(0, 1, (L, 3)) / 0.6, axis=0)      # persistent direction: d[k] = 0.8 d[k-1] + n[k]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        xyz = np.cumsum(3.8 * d, axis=0)
        xyz -= xyz.mean(axis=0)
        ic = np.clip(((xyz.astype(np.float32) + 1000) * 10 + 0.5), 0, 65535).astype(np.uint16)
        recs.append((seq, ic.tobytes(), L))
    return recs


def write_bca_fast(path, lens, rng, label_prefix="syn"):
    """A large synthetic .bca in one vectorised pass (the databases of BASELINE configs 3 / 4 have 10^5 - 10^6 chains; the
    per-chain generator above takes ~0.5 ms per chain): ONE persistent random walk over all residues, cut at the chain
    boundaries, every chain centred on the origin.  Same layout as write_bca_records; different data than gen_bca_chains."""
    lens = np.asarray(lens, np.int64)
    n, tot = len(lens), int(lens.sum())
    start = np.concatenate([[0], np.cumsum(lens)])
    d = lfilter([0.6], [1.0, -0.8], rng.normal(0, 1, (tot, 3)).astype(np.float32) / 0.6, axis=0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = np.cumsum(3.8 * d.astype(np.float64), axis=0)
    first = xyz[start[:-1]] - 3.8 * d[start[:-1]]                          # walk position just before each chain
    mean = (np.add.reduceat(xyz, start[:-1], axis=0) / lens[:, None])
    xyz -= np.repeat(mean, lens, axis=0)
    del first
    ic = np.clip(((xyz.astype(np.float32) + 1000) * 10 + 0.5), 0, 65535).astype(np.uint16)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)[rng.integers(0, 20, tot)]
    with open(path, "wb") as f:
        f.write(struct.pack("<IQQQ", 0xBCABCA, n, 0, 0))
        # per chain: L sequence bytes, then 3L uint16 coordinates (x, y, z interleaved)
        blob = np.empty(7 * tot, np.uint8)
        off = np.repeat(7 * start[:-1], lens) + (np.arange(tot) - np.repeat(start[:-1], lens))          # sequence byte of residue k
        blob[off] = aa
        cbase = np.repeat(7 * start[:-1] + lens, lens) + 6 * (np.arange(tot) - np.repeat(start[:-1], lens))
        icb = ic.view(np.uint8).reshape(tot, 6)
        for b in range(6):
            blob[cbase + b] = icb[:, b]
        f.write(blob.tobytes())
        pos = f.tell()
        f.write(lens.astype(np.uint32).tobytes())
        lab = b"".join(("%s%06d" % (label_prefix, k)).encode() + b"\0" for k in range(n))
        f.write(lab)
        f.seek(4)
        f.write(struct.pack("<QQQ", n, pos, len(lab)))


def write_bca_records(path, recs, labels=None):
    """bcadata.cpp layout: magic, #chains, offset of the length table, label bytes; per chain sequence + 3L uint16
    coordinates; uint32 lengths; NUL-terminated labels."""
    n = len(recs)
    with open(path, "wb") as f:
        f.write(struct.pack("<IQQQ", 0xBCABCA, n, 0, 0))
        for seq, ic, _ in recs:
            f.write(seq)
            f.write(ic)
        pos = f.tell()
        f.write(np.asarray([r[2] for r in recs], np.uint32).tobytes())
        lab = b"".join((labels[k] if labels else "syn%05d" % k).encode() + b"\0" for k in range(n))
        f.write(lab)
        f.seek(4)
        f.write(struct.pack("<QQQ", n, pos, len(lab)))


def write_bca(path, lens, rng):
    """Synthetic .bca; featurisation (DSS) and self-rev then run in LoadDB."""
    write_bca_records(path, gen_bca_chains(lens, rng))


def main_qdb():
    """bench_search.py qdb NQ ND MODE: NQ query chains against an ND-chain .bca database (one GPU's shard of
    BASELINE configs 4 / 5: 256 x 1M -sensitive -> 256 x 125000; 1k x 700k -verysensitive -> 1000 x 87500)."""
    nq, nd, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    lens = bench.scop40_lengths()
    rng = np.random.default_rng(11)
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    with tempfile.TemporaryDirectory() as td:
        q, db, out = os.path.join(td, "q.bca"), os.path.join(td, "db.bca"), os.path.join(td, "hits.tsv")
        write_bca_fast(q, lens[rng.choice(len(lens), nq)], rng, "q")      # the generator and seed of bench.py's configs legs
        t0 = time.perf_counter()
        write_bca_fast(db, lens[rng.choice(len(lens), nd)], rng, "d")
        tgen = time.perf_counter() - t0
        res = {}
        for rep in range(2):
            t0 = time.perf_counter()
            nhits, st = ctx.search_rskdb(q, out, mode, db=db)
            dt = time.perf_counter() - t0
            res["run%d" % rep] = {"seconds": dt, "pairs": int(st[0]), "pairs_per_s": st[0] / dt, "mufilter_in": int(st[2]),
                                  "mufilter_discard": int(st[3]), "mkf_pairs": int(st[4]), "sw_pairs": int(st[5]), "hits": int(nhits),
                                  "tsv_bytes": os.path.getsize(out)}
        print(json.dumps({"input": ".bca query + .bca db (DSS featurisation + self-rev inside the timed call)", "queries": nq,
                          "db_chains": nd, "mode": mode, "db_generation_s": tgen, **res}, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "qdb":
        return main_qdb()
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    mode = sys.argv[2] if len(sys.argv) > 2 else "sensitive"
    if len(sys.argv) > 3 and sys.argv[3] == "bca":
        lens = bench.scop40_lengths()
        rng = np.random.default_rng(7)
        if nch:
            lens = lens[rng.choice(len(lens), nch, replace=nch > len(lens))]
        ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
        with tempfile.TemporaryDirectory() as td:
            db = os.path.join(td, "syn.bca")
            write_bca(db, lens, rng)
            out = os.path.join(td, "hits.tsv")
            res = {}
            use_db = len(sys.argv) > 4 and sys.argv[4] == "db"      # Q vs the same set as -db (mode fast -> prefilter path)
            for rep in range(2):
                t0 = time.perf_counter()
                nhits, st = ctx.search_rskdb(db, out, mode, db=db if use_db else None)
                dt = time.perf_counter() - t0
                res["run%d" % rep] = {"seconds": dt, "pairs": int(st[0]), "pairs_per_s": st[0] / dt, "mufilter_in": int(st[2]),
                                      "mufilter_discard": int(st[3]), "mkf_pairs": int(st[4]), "sw_pairs": int(st[5]), "hits": int(nhits)}
            print(json.dumps({"input": ".bca (host DSS featurisation + GPU self-rev inside the timed call)", "db": use_db, "chains": len(lens),
                              "mode": mode, **res}, indent=1))
        return
    seqs = bench.synth_mu_chains(0x5EED5EEC, nch or None)
    rng = np.random.default_rng(5)
    ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    with tempfile.TemporaryDirectory() as td:
        db = os.path.join(td, "syn.rskdb")
        write_rskdb(db, seqs, rng)
        out = os.path.join(td, "hits.tsv")
        res = {}
        for rep in range(2):
            t0 = time.perf_counter()
            nhits, st = ctx.search_rskdb(db, out, mode)
            dt = time.perf_counter() - t0
            res["run%d" % rep] = {"seconds": dt, "pairs": int(st[0]), "pairs_per_s": st[0] / dt, "mufilter_in": int(st[2]),
                                  "mufilter_discard": int(st[3]), "mkf_pairs": int(st[4]), "sw_pairs": int(st[5]), "hits": int(nhits)}
        print(json.dumps({"chains": len(seqs), "mode": mode, **res}, indent=1))


if __name__ == "__main__":
    main()

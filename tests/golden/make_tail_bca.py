#!/usr/bin/env python3
"""Fixture generator (run by make_golden.sh): inputs of the BASELINE configs[3] / configs[4] shaped cases.
  subset_bca(src, dst, n)      the first n chains of a .bca file (q32.bca = first 32 chains of the reference's q100.bca)
  make_tail(q_path, db_path)   12 query chains + 48 DB chains whose lengths follow the PDB-like lognormal of SURVEY 8d
                               (median ~250) with a tail planted up to 5,000 residues (3 chains > 2048, 9 >= 600)
Structures come from the synthetic generator of tools/bench_search.py (seeded).  Data only."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def read_bca(path):
    raw = open(path, "rb").read()
    magic, n, pos, lab_bytes = struct.unpack_from("<IQQQ", raw, 0)
    assert magic == 0xBCABCA
    lens = np.frombuffer(raw, np.uint32, n, pos)
    labels = raw[pos + 4 * n: pos + 4 * n + lab_bytes].split(b"\0")[:n]
    recs, off = [], 28
    for L in lens:
        L = int(L)
        recs.append((raw[off:off + L], raw[off + L:off + 7 * L], L))
        off += 7 * L
    return recs, [x.decode() for x in labels]


def subset_bca(src, dst, n):
    import bench_search
    recs, labels = read_bca(src)
    bench_search.write_bca_records(dst, recs[:n], labels=labels[:n])


def tail_lengths():
    rng = np.random.default_rng(0x7A11)
    q = np.clip(rng.lognormal(np.log(250), 0.6, 12), 40, 900).astype(int)
    q[0], q[1] = 640, 1210                                  # queries that take the long-chain path themselves
    db = np.clip(rng.lognormal(np.log(250), 0.7, 48), 30, 1900).astype(int)
    db[:6] = [5000, 3200, 2100, 1530, 1024, 1023]           # the planted tail: > 2048 (Mu fallback), > 1024 (row groups / transposed SW)
    db[6:9] = [600, 599, 17]
    return q, db


def make_tail(q_path, db_path):
    import bench_search
    q, db = tail_lengths()
    rng = np.random.default_rng(0x7A12)
    bench_search.write_bca_records(q_path, bench_search.gen_bca_chains(q, rng), labels=["tq%02d" % k for k in range(len(q))])
    bench_search.write_bca_records(db_path, bench_search.gen_bca_chains(db, rng), labels=["td%02d" % k for k in range(len(db))])


if __name__ == "__main__":
    if sys.argv[1] == "subset":
        subset_bca(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        make_tail(sys.argv[2], sys.argv[3])

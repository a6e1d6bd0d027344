#!/usr/bin/env python3
"""Goldens for tests/test_gpu_vs_reference_binary.py: for each of its seeded cases, the one-thread table of the
UNMODIFIED reference binary (oracle/_ref/reseek, built from /root/reference by oracle/Makefile.ref) on the case's synthetic .bca
files -- row count + md5 of the sorted table, md5 of the input files -- so that the GPU test still checks our table on a box the
binary did not reach.  Runs on the CPU (no GPU, no torch); ~10 CPU-minutes.  Output: tests/golden/refbin_goldens.json."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import compare_with_reference as cwr  # noqa: E402

# (n, mode, ndb, seed, long_chains, tail): the parametrisations of the two tests
CASES = [(260, "sensitive", 0, 3, 4, False), (110, "verysensitive", 0, 4, 0, False), (400, "fast", 0, 5, 4, False),
         (60, "sensitive", 500, 6, 4, False), (60, "fast", 500, 7, 4, False),
         (24, "verysensitive", 160, 11, 0, True), (48, "sensitive", 400, 12, 0, True)]


def main():
    out = {}
    for n, mode, ndb, seed, lc, tail in CASES:
        with tempfile.TemporaryDirectory() as td:
            q, db = cwr.write_inputs(td, n, ndb, seed, lc, tail)
            rows, secs = cwr.run_reference(td, q, db, mode, threads=1)
            out[cwr.case_key(n, mode, ndb, seed, lc, tail)] = {
                "rows": len(rows), "sorted_md5": cwr.table_md5(rows), "q_md5": cwr.file_md5(q), "db_md5": cwr.file_md5(db) if db else None,
                "command": "reseek -search q.bca%s -%s -output ref.tsv -threads 1" % (" -db db.bca" if db else "", mode), "reference_seconds": round(secs, 1)}
            print(cwr.case_key(n, mode, ndb, seed, lc, tail), out[cwr.case_key(n, mode, ndb, seed, lc, tail)], flush=True)
    with open(cwr.GOLDENS, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()

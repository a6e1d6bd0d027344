"""Pins the CPU restatement of the Mu k-mer prefilter (P10-P12) against `reseek -prefilter_mu`
outputs of the reference binary (tests/golden/prefilter_*; generated with -threads 1)."""
import gzip
import hashlib
import os

import numpy as np

import fixtures as fx
import oracle_lib as ol


def scores_text(labels_q, labels_t, q, t, s):
    lines = ["%s\t%s\t%d" % (labels_q[a], labels_t[b], c) for a, b, c in zip(q.tolist(), t.tolist(), s.tolist())]
    lines.sort()          # bytewise order == LC_ALL=C sort for ASCII
    return "\n".join(lines) + "\n"


def test_sub1000_default_and_small_bag():
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz", limit=1000)
    assert len(seqs) == 1000
    q, t, s = ol.prefilter(seqs, seqs)
    for B, tag in ((1500, ""), (50, "_b50")):
        rq, rt, rs = ol.rsb(q, t, s, len(seqs), B)
        want = gzip.open(os.path.join(fx.GOLDEN, "prefilter_sub1000%s_scores.tsv.gz" % tag)).read().decode()
        assert scores_text(labels, labels, rq, rt, rs) == want
        want_tmp = gzip.open(os.path.join(fx.GOLDEN, "prefilter_sub1000%s_tmp.tsv.gz" % tag)).read().decode()
        assert fx.prefilter_tmp_tsv(rq, rt) == want_tmp
    # the small bag really truncates
    assert len(ol.rsb(q, t, s, 1000, 50)[0]) < len(q)


def test_scop40_full_checksums():
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz")
    assert len(seqs) == 11211 and sum(len(x) for x in seqs) == 1949312
    q, t, s = ol.prefilter(seqs, seqs, cap=12_000_000)
    rq, rt, rs = ol.rsb(q, t, s, len(seqs), 1500)
    want = dict(zip(*[iter(open(os.path.join(fx.GOLDEN, "prefilter_scop40_full.md5.txt")).read().split())] * 2))
    assert len(rq) == int(want["lines"])
    assert hashlib.md5(scores_text(labels, labels, rq, rt, rs).encode()).hexdigest() == want["sorted_scores_md5"]
    assert hashlib.md5(fx.prefilter_tmp_tsv(rq, rt).encode()).hexdigest() == want["tmp_tsv_md5"]


def test_neighbourhood_modes_match_muprefilter():
    """k-mer neighbourhoods as `-search -fast -db` runs them (MuPreFilter muprefilter.cpp:70): 80 queries
    against 1000 targets, query-side ("idxq", exact matches listed twice) and target-side ("idxt")."""
    labels, seqs = fx.read_mu_fasta("scop40.mu.fa.gz", limit=1000)
    qs = seqs[:80]
    for mode, tag in ((1, "h80"), (2, "h80t")):
        q, t, s = ol.prefilter(qs, seqs, mode=mode)
        rq, rt, rs = ol.rsb(q, t, s, len(qs), 1500)
        want = gzip.open(os.path.join(fx.GOLDEN, "prefilter_hood_%s_scores.tsv.gz" % tag)).read().decode()
        assert scores_text(labels, labels, rq, rt, rs) == want
        want_tmp = gzip.open(os.path.join(fx.GOLDEN, "prefilter_hood_%s_tmp.tsv.gz" % tag)).read().decode()
        assert fx.prefilter_tmp_tsv(rq, rt) == want_tmp

"""The deterministic per-query top-B (rsk_rsb_merge: score desc, then target asc) used by the target-sharded `-fast -db`
path, against numpy, and its relation to the reference's RankedScoresBag as replayed by rsk_rsb_select (pinned to the
reference's -threads 1 files by test_oracle_prefilter / test_gpu_prefilter): identical without truncation, and with
truncation identical except among candidates tied with the cut score.  Host code: no GPU needed."""
import numpy as np

from reseek_amd import capi


def np_topb(rows, nq, B):
    out = []
    for q in range(nq):
        r = rows[rows[:, 0] == q]
        o = np.lexsort((r[:, 1], -r[:, 2]))[:B]
        out.append(r[o])
    return np.concatenate(out) if out else np.zeros((0, 3), np.int32)


def rsb_select(rows, nq, B):
    import ctypes as C
    q = np.ascontiguousarray(rows[:, 0], np.uint32); t = np.ascontiguousarray(rows[:, 1], np.uint32); s = np.ascontiguousarray(rows[:, 2], np.uint32)
    oq, ot, os_ = (np.zeros(len(q), np.uint32) for _ in range(3))
    n = C.c_size_t()
    rc = capi.lib().rsk_rsb_select(capi._p(q, capi.u32p), capi._p(t, capi.u32p), capi._p(s, capi.u32p), len(q), nq, B, capi._p(oq, capi.u32p),
                                   capi._p(ot, capi.u32p), capi._p(os_, capi.u32p), C.byref(n), None)
    assert rc == 0
    return np.stack([oq[:n.value], ot[:n.value], os_[:n.value]], axis=1).astype(np.int32)


def make(seed, nq, nt, density, smax):
    rng = np.random.default_rng(seed)
    m = rng.random((nq, nt)) < density
    q, t = np.nonzero(m)
    s = rng.integers(1, smax, len(q))
    return np.stack([q, t, s], axis=1).astype(np.int32)


def test_merge_equals_numpy_and_is_shard_invariant():
    rows = make(1, 40, 3000, 0.2, 60)                     # ~600 candidates per query, many score ties
    for B in (50, 1500):
        want = np_topb(rows, 40, B)
        got = capi.rsb_merge(rows, 40, B)
        assert np.array_equal(got, want)
        # local top-B per target shard, then the merge of the locals == the global top-B (any shard count, any order)
        for ns in (2, 3, 7):
            bounds = np.linspace(0, 3000, ns + 1).astype(int)
            local = [capi.rsb_merge(rows[(rows[:, 1] >= lo) & (rows[:, 1] < hi)], 40, B) for lo, hi in zip(bounds[:-1], bounds[1:])]
            merged = capi.rsb_merge(np.concatenate(local[::-1]), 40, B)
            assert np.array_equal(merged, want)


def test_relation_to_the_reference_bag():
    rows = make(2, 30, 2000, 0.15, 40)
    # no truncation: the same set
    a = capi.rsb_merge(rows, 30, 1500)
    b = rsb_select(rows, 30, 1500)
    assert sorted(map(tuple, a)) == sorted(map(tuple, b))
    # truncation: same size per query, same members above the cut score, any difference is among ties AT the cut
    B = 50
    a = capi.rsb_merge(rows, 30, B)
    b = rsb_select(rows, 30, B)
    for q in range(30):
        ra, rb = a[a[:, 0] == q], b[b[:, 0] == q]
        assert len(ra) == len(rb)
        if len(ra) == 0:
            continue
        cut = ra[:, 2].min()
        assert rb[:, 2].min() == cut
        sa, sb = set(map(tuple, ra[ra[:, 2] > cut])), set(map(tuple, rb[rb[:, 2] > cut]))
        assert sa == sb


def test_empty_and_bad_arguments():
    assert capi.rsb_merge(np.zeros((0, 3), np.int32), 5, 10).shape == (0, 3)
    import pytest
    with pytest.raises(capi.RskError):
        capi.rsb_merge(np.array([[7, 0, 1]], np.int32), 5, 10)        # query index out of range


def test_sorted_key_replay_equals_the_triple_replay(tmp_path):
    """rsk_rsb_select_keys (keys as rsk_triples_sort_dev leaves them: query << 48 | target << 16 | score, ascending) keeps
    exactly what rsk_rsb_select keeps of the same triples in any order -- overflowing bags and tie-heavy scores included --
    and writes the same hand-off file."""
    for seed, nq, nt, dens, smax, B in ((3, 30, 2500, 0.6, 12, 50), (4, 7, 900, 0.9, 300, 1500), (5, 65, 400, 0.5, 5, 20)):
        rows = make(seed, nq, nt, dens, smax)
        rng = np.random.default_rng(seed)
        shuf = rows[rng.permutation(len(rows))]
        fa, fb = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
        a = capi.rsb_select(shuf[:, 0], shuf[:, 1], shuf[:, 2], nq, B, tmp_tsv_path=fa)
        keys = np.sort((rows[:, 0].astype(np.uint64) << np.uint64(48)) | (rows[:, 1].astype(np.uint64) << np.uint64(16)) | rows[:, 2].astype(np.uint64))
        b = capi.rsb_select_keys(keys, nq, B, tmp_tsv_path=fb)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert open(fa, "rb").read() == open(fb, "rb").read()
    # empty input
    q, t, s = capi.rsb_select_keys(np.zeros(0, np.uint64), 5, 10)
    assert len(q) == 0

"""The reference's own X-drop / SW self-test vectors (test_xdrop.cpp:177-187, swgaplessprof.cpp:158-166: nine
hard-coded peptide pairs under BLOSUM62) plus 300 random peptide pairs, all run through the reference's SWFast,
SWGapless, XDropFwd, XDropBwd and MergeFwdBwd by oracle/ref_harness `xdropkat` (tests/golden/make_golden.sh).
Pins, bit for bit: (a) the oracle's SWFast / gapless / X-drop restatements (CPU tests) and (b) the PRODUCT's X-drop kernel
(k_xdrop_wave on an explicit score matrix, through rsk_xdrop_fwd / rsk_xdrop_bwd: the -m gpu test at the end; the library
has no host implementation of that DP)."""
import gzip
import os
import struct

import numpy as np

import pytest

import fixtures as fx
import oracle_lib as ol


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def read_cases():
    buf = gzip.open(os.path.join(fx.GOLDEN, "xdropkat_309.bin.gz"), "rb").read()
    assert buf[:8] == b"XDKAT1\0\0"
    pos = 8

    def u32():
        nonlocal pos
        v = struct.unpack_from("<I", buf, pos)[0]
        pos += 4
        return v

    def f32():
        nonlocal pos
        v = struct.unpack_from("<f", buf, pos)[0]
        pos += 4
        return v

    def s():
        nonlocal pos
        n = u32()
        v = buf[pos:pos + n].decode()
        pos += n
        return v

    cases = []
    for _ in range(u32()):
        c = {"A": s(), "B": s(), "open": f32(), "ext": f32(), "X": f32()}
        LA, LB = len(c["A"]), len(c["B"])
        c["S"] = np.frombuffer(buf, "<f4", LA * LB, pos).reshape(LA, LB).copy()
        pos += 4 * LA * LB
        c["sw"] = (f32(), u32(), u32(), u32(), u32(), s())
        c["gapless"] = (f32(), u32(), u32(), u32())
        c["xdrop"] = None
        if u32():
            mid = (u32(), u32())
            fwd = (f32(), u32(), u32(), s())
            bwd = (f32(), u32(), u32(), s())
            merged = (u32(), u32(), u32(), u32(), s()) if u32() else None
            c["xdrop"] = (mid, fwd, bwd, merged)
        cases.append(c)
    assert pos == len(buf)
    return cases


CASES = read_cases()


def test_fixture_holds_the_reference_self_test_pairs():
    assert len(CASES) == 309
    assert (CASES[0]["A"], CASES[0]["B"]) == ("DVLGYLRFLTKGERQANLNF", "WVLGLRFLTKGERQANLNF")
    assert (CASES[8]["A"], CASES[8]["B"]) == ("QVE", "SEQVENCE")
    assert all(c["open"] == -3 and c["ext"] == -1 and c["X"] == 8 for c in CASES[:9])
    assert sum(c["xdrop"] is not None for c in CASES) > 250


def test_oracle_swfast_and_gapless_match_the_reference():
    for k, c in enumerate(CASES):
        score, loi, loj, leni, lenj, path = c["sw"]
        s, i, j, p = ol.sw_fast_matrix(c["S"], c["open"], c["ext"])
        assert bits(s) == bits(score) and p == path, k
        if path:
            assert (i, j) == (loi, loj), k
            assert (sum(ch in "MD" for ch in p), sum(ch in "MI" for ch in p)) == (leni, lenj), k
        gs, gi, gj, gcols = c["gapless"]
        s2, bi, bj = ol.sw_gapless_matrix(c["S"])
        assert bits(s2) == bits(gs), k
        if gs > 0:
            # SWGapless (swgapless.cpp:101) reports the start of the run, SWFastGapless (:46) its last cell; their tie
            # rules differ, so only the score is pinned here and the run must fit before the reported end
            assert bi + 1 >= gcols and bj + 1 >= gcols


def _xdrop_cases(fwd_fn, bwd_fn, merge_fn):
    n = nm = 0
    for k, c in enumerate(CASES):
        if c["xdrop"] is None:
            continue
        (ma, mb), fwd, bwd, merged = c["xdrop"]
        LA, LB = c["S"].shape
        fs, fp = fwd_fn(c["S"], c["X"], c["open"], c["ext"], ma + 1, mb + 1)
        assert bits(fs) == bits(fwd[0]) and fp == fwd[3], (k, "fwd")
        bs, bp = bwd_fn(c["S"], c["X"], c["open"], c["ext"], ma, mb)
        assert bits(bs) == bits(bwd[0]) and bp == bwd[3], (k, "bwd")
        n += 1
        if merged is not None:
            assert merge_fn(LA, LB, ma + 1, mb + 1, fp, ma, mb, bp) == merged, (k, "merge")
            nm += 1
    assert n > 250 and nm > 250


def test_oracle_xdrop_fwd_bwd_merge_match_the_reference():
    _xdrop_cases(ol.xdrop_fwd, ol.xdrop_bwd, lambda LA, LB, *a: ol.merge_fwd_bwd(*a))


def test_merge_c_abi_rejects_bad_arguments():
    from reseek_amd import capi
    with pytest.raises(RuntimeError):
        capi.merge_fwd_bwd(5, 5, 1, 1, "", 0, 0, "")


@pytest.mark.gpu
def test_device_xdrop_kernel_matches_the_reference_self_test_vectors():
    """k_xdrop_wave<EXPLICIT>: the kernel of the search's long-chain batch, fed the reference's own score matrices."""
    import reseek_amd
    from reseek_amd import capi
    ctx = reseek_amd.Ctx(0)
    _xdrop_cases(lambda *a: capi.xdrop_fwd(ctx, *a), lambda *a: capi.xdrop_bwd(ctx, *a), capi.merge_fwd_bwd)
    S = CASES[0]["S"]
    with pytest.raises(RuntimeError):
        capi.xdrop_bwd(ctx, S, 8.0, -3.0, -1.0, S.shape[0], 0)
    with pytest.raises(RuntimeError):
        capi.xdrop_fwd(ctx, S, 8.0, -3.0, -1.0, S.shape[0], 1)

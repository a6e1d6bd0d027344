"""ctypes loader for oracle/librsk_oracle.so -- the CPU restatement (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "librsk_oracle.so")

_lib = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)


class Stats(C.Structure):
    _fields_ = [("lo_a", C.c_uint32), ("lo_b", C.c_uint32), ("hi_a", C.c_uint32), ("hi_b", C.c_uint32),
                ("ids", C.c_uint32), ("gaps", C.c_uint32), ("score", C.c_float), ("lddt", C.c_float),
                ("ts", C.c_float), ("pvalue", C.c_float), ("evalue", C.c_float), ("qual", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile"], cwd=ROOT)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "rsk_oracle.c")
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
            build()
        L = C.CDLL(SO)
        L.rsko_mu_gapless.restype = C.c_int
        L.rsko_mu_gapless.argtypes = [u8p, C.c_int, u8p, C.c_int, u32p, u32p]
        L.rsko_mu_pinop.restype = C.c_int
        L.rsko_mu_pinop.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        L.rsko_mu_sw.restype = C.c_int
        L.rsko_mu_sw.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, i32p]
        L.rsko_mu_filter.restype = C.c_float
        L.rsko_mu_filter.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_float, i32p, i32p]
        L.rsko_align_pair.restype = C.c_float
        L.rsko_align_pair.argtypes = [u8p, C.c_int, u8p, C.c_int, u32p, u32p, C.c_char_p, u32p]
        L.rsko_lddt.restype = C.c_float
        L.rsko_lddt.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, f32p, f32p, f32p, f32p, f32p, f32p]
        L.rsko_calc_evalue.restype = C.c_int
        L.rsko_calc_evalue.argtypes = [C.c_float, C.c_float, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                       C.c_float, C.c_float, f32p, f32p, f32p, f32p, f32p, f32p, C.POINTER(Stats)]
        L.rsko_diag_hsp.restype = C.c_int
        L.rsko_diag_hsp.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int]
        L.rsko_mu_gapless_pairs.restype = None
        L.rsko_mu_gapless_pairs.argtypes = [u8p, u32p, u32p, u32p, C.c_size_t, i32p]
        L.rsko_mu_filter_pairs.restype = None
        L.rsko_mu_filter_pairs.argtypes = [u8p, u32p, u32p, u32p, C.c_size_t, C.c_int, C.c_int, C.c_float,
                                           i32p, i32p, f32p]
        L.rsko_set_smx.restype = None
        L.rsko_set_smx.argtypes = [u8p, C.c_int, u8p, C.c_int, f32p]
        L.rsko_sw_fast.restype = C.c_float
        L.rsko_sw_fast.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, u32p, u32p, C.c_char_p, u32p, u8p]
        L.rsko_sw_gapless_float.restype = C.c_float
        L.rsko_sw_gapless_float.argtypes = [f32p, C.c_int, C.c_int, u32p, u32p]
        L.rsko_gapless_profb.restype = C.c_float
        L.rsko_gapless_profb.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.rsko_gapless_float_pair.restype = C.c_float
        L.rsko_gapless_float_pair.argtypes = [u8p, C.c_int, u8p, C.c_int, u32p, u32p]
        L.rsko_mkf_seed.restype = C.c_int
        L.rsko_mkf_seed.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.POINTER(C.c_int)]
        L.rsko_prefilter.restype = C.c_size_t
        L.rsko_prefilter.argtypes = [u8p, u32p, C.c_uint32, u8p, u32p, C.c_uint32, u32p, u32p, u32p, C.c_size_t]
        L.rsko_prefilter_mode.restype = C.c_size_t
        L.rsko_prefilter_mode.argtypes = [u8p, u32p, C.c_uint32, u8p, u32p, C.c_uint32, C.c_int, u32p, u32p, u32p, C.c_size_t]
        L.rsko_xdrop_fwd.restype = C.c_float
        L.rsko_xdrop_fwd.argtypes = [f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_char_p, u32p]
        L.rsko_xdrop_bwd.restype = C.c_float
        L.rsko_xdrop_bwd.argtypes = L.rsko_xdrop_fwd.argtypes
        L.rsko_merge_fwd_bwd.restype = None
        L.rsko_merge_fwd_bwd.argtypes = [C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, u32p, C.c_char_p]
        L.rsko_rsb.restype = C.c_size_t
        L.rsko_rsb.argtypes = [u32p, u32p, u32p, C.c_size_t, C.c_uint32, C.c_uint32, u32p, u32p, u32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def mu_gapless(A, B):
    bi, bj = C.c_uint32(), C.c_uint32()
    s = lib().rsko_mu_gapless(_p(A, u8p), len(A), _p(B, u8p), len(B), C.byref(bi), C.byref(bj))
    return s, bi.value, bj.value


def mu_pinop(A, B, open_=-2, ext=-1):
    return lib().rsko_mu_pinop(_p(A, u8p), len(A), _p(B, u8p), len(B), open_, ext)


def mu_sw(A, B, open_=2, ext=1):
    sat = C.c_int32()
    s = lib().rsko_mu_sw(_p(A, u8p), len(A), _p(B, u8p), len(B), open_, ext, C.byref(sat))
    return s, sat.value


def mu_filter(A, B, omega_fwd, open_=2, ext=1):
    f, r = C.c_int32(), C.c_int32()
    s = lib().rsko_mu_filter(_p(A, u8p), len(A), _p(B, u8p), len(B), open_, ext, omega_fwd, C.byref(f), C.byref(r))
    return s, f.value, r.value


def mkf_seed(Q, T, x1=8, min_score=50, cap=64):
    """-> (found, nkept, kept int32 [min(nkept, cap), 4])"""
    kept = np.zeros((cap, 4), np.int32)
    found = C.c_int()
    nk = lib().rsko_mkf_seed(_p(Q, u8p), len(Q), _p(T, u8p), len(T), x1, min_score, cap, _p(kept, i32p), C.byref(found))
    return bool(found.value), nk, kept[:min(nk, cap)].copy()


def gapless_profb(A, B):
    return lib().rsko_gapless_profb(_p(A, u8p), len(A), _p(B, u8p), len(B))


def gapless_float_pair(profA, profB):
    pa = np.ascontiguousarray(profA)
    pb = np.ascontiguousarray(profB)
    bi, bj = C.c_uint32(), C.c_uint32()
    s = lib().rsko_gapless_float_pair(_p(pa, u8p), pa.shape[1], _p(pb, u8p), pb.shape[1], C.byref(bi), C.byref(bj))
    return s, bi.value, bj.value


def sw_fast_matrix(S, open_, ext):
    """SWFast (sw.cpp:79) on an explicit score matrix float32 [LA, LB] -> (score, lo_i, lo_j, path)"""
    S = np.ascontiguousarray(S, np.float32)
    LA, LB = S.shape
    buf = C.create_string_buffer(LA + LB + 2)
    lo_i, lo_j, n = C.c_uint32(0xFFFFFFFF), C.c_uint32(0xFFFFFFFF), C.c_uint32()
    s = lib().rsko_sw_fast(_p(S, f32p), LA, LB, open_, ext, C.byref(lo_i), C.byref(lo_j), buf, C.byref(n), None)
    return s, lo_i.value, lo_j.value, buf.value.decode()


def _xdrop(fn, S, X, open_, ext, a, b):
    S = np.ascontiguousarray(S, np.float32)
    LA, LB = S.shape
    buf = C.create_string_buffer(LA + LB + 4)
    n = C.c_uint32()
    s = fn(_p(S, f32p), LA, LB, X, open_, ext, a, b, buf, C.byref(n))
    return s, buf.value.decode()


def xdrop_fwd(S, X, open_, ext, lo_a, lo_b):
    """XDropFwd (xdropfwd.cpp:71) on an explicit score matrix, from (lo_a, lo_b) to the ends -> (score, path)"""
    return _xdrop(lib().rsko_xdrop_fwd, S, X, open_, ext, lo_a, lo_b)


def xdrop_bwd(S, X, open_, ext, hi_a, hi_b):
    """XDropBwd (xdropbwd.cpp:28), from (hi_a, hi_b) to the starts -> (score, path)"""
    return _xdrop(lib().rsko_xdrop_bwd, S, X, open_, ext, hi_a, hi_b)


def merge_fwd_bwd(fwd_lo_a, fwd_lo_b, fwd_path, bwd_hi_a, bwd_hi_b, bwd_path):
    """MergeFwdBwd (mergefwdback.cpp:6) -> (lo_a, lo_b, hi_a, hi_b, path)"""
    out = (C.c_uint32 * 4)()
    buf = C.create_string_buffer(len(fwd_path) + len(bwd_path) + 2)
    lib().rsko_merge_fwd_bwd(fwd_lo_a, fwd_lo_b, fwd_path.encode(), bwd_hi_a, bwd_hi_b, bwd_path.encode(), out, buf)
    return out[0], out[1], out[2], out[3], buf.value.decode()


def sw_gapless_matrix(S):
    """SWFastGapless (swgapless.cpp:46) on an explicit score matrix -> (score, besti, bestj)"""
    S = np.ascontiguousarray(S, np.float32)
    bi, bj = C.c_uint32(), C.c_uint32()
    s = lib().rsko_sw_gapless_float(_p(S, f32p), S.shape[0], S.shape[1], C.byref(bi), C.byref(bj))
    return s, bi.value, bj.value


def align_pair(profA, profB):
    """profA uint8 [8, LA] C-contiguous -> (score, lo_i, lo_j, path)"""
    LA, LB = profA.shape[1], profB.shape[1]
    pa = np.ascontiguousarray(profA)
    pb = np.ascontiguousarray(profB)
    buf = C.create_string_buffer(LA + LB + 2)
    lo_i, lo_j, n = C.c_uint32(0xFFFFFFFF), C.c_uint32(0xFFFFFFFF), C.c_uint32()
    s = lib().rsko_align_pair(_p(pa, u8p), LA, _p(pb, u8p), LB, C.byref(lo_i), C.byref(lo_j), buf, C.byref(n))
    return s, lo_i.value, lo_j.value, buf.value.decode()


def calc_evalue(score, min_fwd, path, lo_a, lo_b, ca, cb):
    st = Stats()
    ok = lib().rsko_calc_evalue(score, min_fwd, path.encode(), lo_a, lo_b, ca.L, cb.L, ca.selfrev, cb.selfrev,
                                _p(ca.x, f32p), _p(ca.y, f32p), _p(ca.z, f32p),
                                _p(cb.x, f32p), _p(cb.y, f32p), _p(cb.z, f32p), C.byref(st))
    return ok, st


def diag_hsp(Q, T, d):
    return lib().rsko_diag_hsp(_p(Q, u8p), len(Q), _p(T, u8p), len(T), d)


def concat_mu(seqs):
    off = np.zeros(len(seqs) + 1, np.uint32)
    off[1:] = np.cumsum([len(s) for s in seqs])
    mu = np.concatenate(seqs).astype(np.uint8) if seqs else np.zeros(0, np.uint8)
    return mu, off


def mu_gapless_pairs(seqs, ia, ib):
    mu, off = concat_mu(seqs)
    ia = np.ascontiguousarray(ia, np.uint32)
    ib = np.ascontiguousarray(ib, np.uint32)
    out = np.zeros(len(ia), np.int32)
    lib().rsko_mu_gapless_pairs(_p(mu, u8p), _p(off, u32p), _p(ia, u32p), _p(ib, u32p), len(ia), _p(out, i32p))
    return out


def mu_filter_pairs(seqs, ia, ib, omega_fwd, open_=2, ext=1):
    mu, off = concat_mu(seqs)
    ia = np.ascontiguousarray(ia, np.uint32)
    ib = np.ascontiguousarray(ib, np.uint32)
    f = np.zeros(len(ia), np.int32)
    r = np.zeros(len(ia), np.int32)
    s = np.zeros(len(ia), np.float32)
    lib().rsko_mu_filter_pairs(_p(mu, u8p), _p(off, u32p), _p(ia, u32p), _p(ib, u32p), len(ia), open_, ext,
                               omega_fwd, _p(f, i32p), _p(r, i32p), _p(s, f32p))
    return f, r, s


def prefilter(qseqs, tseqs, cap=None, mode=0):
    """Mu prefilter (P10/P11; mode 0 exact k-mers, 1 idxq, 2 idxt): -> (q, t, score) arrays, targets in order."""
    qmu, qoff = concat_mu(qseqs)
    tmu, toff = concat_mu(tseqs)
    cap = cap or max(1024, len(qseqs) * len(tseqs))
    oq = np.zeros(cap, np.uint32)
    ot = np.zeros(cap, np.uint32)
    os_ = np.zeros(cap, np.uint32)
    n = lib().rsko_prefilter_mode(_p(qmu, u8p), _p(qoff, u32p), len(qseqs), _p(tmu, u8p), _p(toff, u32p), len(tseqs), mode,
                                  _p(oq, u32p), _p(ot, u32p), _p(os_, u32p), cap)
    assert n <= cap
    return oq[:n], ot[:n], os_[:n]


def rsb(q, t, score, nq, B):
    """RankedScoresBag (P12): top-B per query with the reference's truncation/tie behaviour."""
    q = np.ascontiguousarray(q, np.uint32)
    t = np.ascontiguousarray(t, np.uint32)
    score = np.ascontiguousarray(score, np.uint32)
    oq = np.zeros(len(q), np.uint32)
    ot = np.zeros(len(q), np.uint32)
    os_ = np.zeros(len(q), np.uint32)
    n = lib().rsko_rsb(_p(q, u32p), _p(t, u32p), _p(score, u32p), len(q), nq, B, _p(oq, u32p), _p(ot, u32p), _p(os_, u32p))
    return oq[:n], ot[:n], os_[:n]

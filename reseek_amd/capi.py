"""ctypes binding of include/reseek_amd.h (librsk.so).  No compute happens in Python."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RSK_LIB: another build of the same library (kernel-variant experiments: tools/exp/)
LIB_PATH = os.environ.get("RSK_LIB") or os.path.join(HERE, "librsk.so")

_lib = None

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)

# every symbol include/reseek_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rsk_version": (C.c_char_p, []),
    "rsk_last_error": (C.c_char_p, []),
    "rsk_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rsk_ctx_destroy": (None, [C.c_void_p]),
    "rsk_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsk_ctx_sync": (C.c_int, [C.c_void_p]),
    "rsk_ctx_trim": (None, [C.c_void_p]),
    "rsk_shutdown": (None, []),
    "rsk_abi_version": (C.c_int, []),
    "rsk_ctx_last_kernel_ms": (C.c_float, [C.c_void_p]),
    "rsk_db_create": (C.c_int, [C.c_void_p, C.c_uint32, u32p, u8p, u8p, f32p, f32p, f32p, f32p, C.POINTER(C.c_void_p)]),
    "rsk_db_destroy": (None, [C.c_void_p]),
    "rsk_db_nchains": (C.c_uint32, [C.c_void_p]),
    "rsk_db_nresidues": (C.c_uint64, [C.c_void_p]),
    "rsk_db_hbm_bytes": (C.c_uint64, [C.c_void_p]),
    "rsk_db_set_seq": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rsk_mu_gapless_matrix_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "rsk_mu_gapless_hits_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_uint32, C.c_void_p]),
    "rsk_mu_filter_window_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rsk_len_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rsk_mu_gapless_shard_window": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rsk_mu_gapless_hits_window_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.c_uint32, C.c_void_p]),
    "rsk_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rsk_comm_create": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rsk_comm_destroy": (None, [C.c_void_p]),
    "rsk_comm_rank": (C.c_int, [C.c_void_p]),
    "rsk_comm_world": (C.c_int, [C.c_void_p]),
    "rsk_gather_hits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p), u64p, u64p]),
    "rsk_mu_gapless_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, i32p, u32p, u32p]),
    "rsk_mu_gapless_last_work": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "rsk_mu_sw_matrix_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_size_t]),
    "rsk_mu_filter_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "rsk_mu_filter_last_work": (C.c_int, [C.c_void_p, u64p, u64p]),
    "rsk_pairs_sort_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]),
    "rsk_mu_prefilter_range_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    "rsk_mu_prefilter_last_work": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4),
    "rsk_triples_sort_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rsk_rsb_select_keys": (C.c_int, [C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_size_t), C.c_char_p]),
}


class Aln(C.Structure):
    _fields_ = [("score", C.c_float), ("lo_a", C.c_uint32), ("lo_b", C.c_uint32), ("hi_a", C.c_uint32),
                ("hi_b", C.c_uint32), ("ids", C.c_uint32), ("gaps", C.c_uint32), ("path_len", C.c_uint32),
                ("path_off", C.c_uint64), ("lddt", C.c_float), ("ts", C.c_float), ("pvalue", C.c_float),
                ("evalue", C.c_float), ("qual", C.c_float), ("nident", C.c_uint32)]


SIGNATURES["rsk_align_paths_bytes"] = (C.c_size_t, [C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t])
SIGNATURES["rsk_align_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, C.c_float, C.c_float,
                                           C.c_float, C.POINTER(Aln), C.c_char_p, C.c_size_t])
SIGNATURES["rsk_align_last_work"] = (C.c_int, [C.c_void_p, u64p, u64p, u64p])
SIGNATURES["rsk_align_last_times"] = (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)])

SIGNATURES["rsk_search_rskdb"] = (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, C.c_int,
                                            C.c_char_p, u64p, u64p])

SIGNATURES["rsk_mu_prefilter_dev"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p])
SIGNATURES["rsk_selftest_format"] = (C.c_uint64, [C.c_uint64, C.c_uint64])
SIGNATURES["rsk_dss_densities"] = (C.c_int, [C.c_void_p, C.c_uint32, u32p, f32p, f32p, f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.c_int, C.c_int, u32p, u32p, u32p, u32p])
SIGNATURES["rsk_dss_featurize_reversed"] = (C.c_int, [C.c_char_p, f32p, f32p, f32p, C.c_uint32, C.POINTER(C.c_uint8)])
SIGNATURES["rsk_dss_featurize"] = (C.c_int, [C.c_char_p, f32p, f32p, f32p, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)])
SIGNATURES["rsk_bca_info"] = (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), u32p, u32p])
SIGNATURES["rsk_bca_read_chain"] = (C.c_int, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_size_t, C.c_char_p, f32p, f32p, f32p, C.c_uint32, u32p])
SIGNATURES["rsk_mkf_seed_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, C.c_int, C.c_int, C.c_uint32,
                                              C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t), u32p, u32p, C.POINTER(C.c_int32)])
SIGNATURES["rsk_xdrop_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, u32p, u32p, C.c_size_t, C.c_float, C.c_float,
                                           C.c_float, f32p, f32p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), u32p,
                                           C.POINTER(C.c_uint64), u32p])
SIGNATURES["rsk_mkf_align_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, u32p, i32p, i32p, i32p, C.c_float,
                                               C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Aln), u8p, C.c_char_p, C.c_size_t])
SIGNATURES["rsk_mkf_chain_align_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, u32p, i32p, i32p, i32p, i32p, C.c_float,
                                                     C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Aln), u8p, C.c_char_p, C.c_size_t])
SIGNATURES["rsk_xdrop_fwd"] = (C.c_int, [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                         f32p, C.c_char_p, C.c_size_t, u32p])
SIGNATURES["rsk_xdrop_bwd"] = SIGNATURES["rsk_xdrop_fwd"]
SIGNATURES["rsk_merge_fwd_bwd"] = (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p,
                                             u32p, u32p, u32p, u32p, C.c_char_p, C.c_size_t, u32p])
SIGNATURES["rsk_mu_pinop_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, C.c_int, C.c_int,
                                              C.POINTER(C.c_int32)])
SIGNATURES["rsk_mu_gapless_profb_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, C.POINTER(C.c_float)])
SIGNATURES["rsk_gapless_float_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, C.POINTER(C.c_float),
                                                   u32p, u32p])
SIGNATURES["rsk_mu_filter_pairs"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, C.c_int, C.c_int,
                                               C.c_float, C.c_float, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int32)])
SIGNATURES["rsk_rsb_select"] = (C.c_int, [u32p, u32p, u32p, C.c_size_t, C.c_uint32, C.c_uint32, u32p, u32p, u32p,
                                          C.POINTER(C.c_size_t), C.c_char_p])

GAP_OPEN = -0.685533     # namedparams.cpp:45
GAP_EXT = -0.051881      # namedparams.cpp:46


class SearchOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_char_p), ("columns", C.c_char_p), ("evalue", C.c_double), ("evalue_set", C.c_int),
                ("mints", C.c_double), ("mints_set", C.c_int), ("pvalue", C.c_double), ("pvalue_set", C.c_int),
                ("noself", C.c_int), ("selfrev0", C.c_int), ("idx_mode", C.c_int), ("rsb_size", C.c_uint32),
                ("dbmu", C.c_char_p), ("keeptmp", C.c_int), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("devices", C.c_char_p), ("hits_digest", C.c_int)]


SIGNATURES["rsk_search"] = (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(SearchOpts), C.c_char_p, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)])
SIGNATURES["rsk_fast_shard_open"] = (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(SearchOpts), C.POINTER(C.c_void_p)])
SIGNATURES["rsk_fast_shard_candidates"] = (C.c_int, [C.c_void_p, C.POINTER(u32p), C.POINTER(u32p), C.POINTER(u32p), C.POINTER(C.c_size_t)])
SIGNATURES["rsk_fast_shard_finish"] = (C.c_int, [C.c_void_p, u32p, u32p, u32p, C.c_size_t, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_uint64)])
SIGNATURES["rsk_fast_shard_triples"] = SIGNATURES["rsk_fast_shard_candidates"]
SIGNATURES["rsk_fast_shard_finish_exact"] = SIGNATURES["rsk_fast_shard_finish"]
SIGNATURES["rsk_fast_shard_close"] = (None, [C.c_void_p])
SIGNATURES["rsk_rsb_merge"] = (C.c_int, [u32p, u32p, u32p, C.c_size_t, C.c_uint32, C.c_uint32, u32p, u32p, u32p, C.POINTER(C.c_size_t)])
SIGNATURES["rsk_dss_densities_host"] = (C.c_int, [f32p, f32p, f32p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)])
class PathCounters(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("sw_pairs", C.c_uint64), ("sw_pairs_scored", C.c_uint64), ("sw_pairs_rescored", C.c_uint64),
                ("upload_copies", C.c_uint64), ("upload_bytes", C.c_uint64), ("db_batches", C.c_uint64), ("loader_seconds", C.c_double),
                ("featurise_seconds", C.c_double), ("upload_seconds", C.c_double), ("swqp_cycles", C.c_uint64), ("swqp_ref_ticks", C.c_uint64)]


SIGNATURES["rsk_bca_to_rskdb"] = (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(SearchOpts), C.c_int, C.c_char_p, C.POINTER(C.c_uint64)])
SIGNATURES["rsk_path_counters_read"] = (C.c_int, [C.c_void_p, C.POINTER(PathCounters)])
SIGNATURES["rsk_path_counters_reset"] = (C.c_int, [C.c_void_p])
SIGNATURES["rsk_shard_range"] = (C.c_int, [C.c_int, u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
SIGNATURES["rsk_bca_copy"] = (C.c_int, [C.c_char_p, C.c_char_p])
SIGNATURES["rsk_bca_to_mu_fasta"] = (C.c_int, [C.c_char_p, C.c_char_p])


def read_hits_digest(path):
    """the line rsk_search writes with hits_digest=1 -> (lines, bytes, sum, xor) as ints"""
    f = open(path).read().split("\t")
    assert f[0] == "digest", f
    return int(f[1]), int(f[2]), int(f[3], 16), int(f[4], 16)


def combine_hits_digests(ds):
    """digest of the union of the tables whose digests are given (shards of one search)"""
    lines = sum(d[0] for d in ds); nbytes = sum(d[1] for d in ds)
    s = sum(d[2] for d in ds) & 0xFFFFFFFFFFFFFFFF
    x = 0
    for d in ds:
        x ^= d[3]
    return lines, nbytes, s, x


class RskError(RuntimeError):
    pass


def lib():
    """Load librsk.so.  Raises if the HIP extension has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RskError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). reseek_amd has no CPU fallback.")
        try:
            # load torch's HIP runtime first so librsk.so (NEEDED libamdhip64.so) binds to the same
            # copy -- two HIP runtimes in one process do not see the device.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("RSK_LIB"):          # an older build of the library in an A/B run (tools/exp/): symbols added since are absent
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RskError(f"librsk error {rc}: {lib().rsk_last_error().decode()}")


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Ctx:
    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        _check(lib().rsk_ctx_create(device, C.byref(h)))
        self.h = h
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        _check(lib().rsk_ctx_set_stream(self.h, C.c_void_p(stream)))

    def sync(self):
        _check(lib().rsk_ctx_sync(self.h))

    def last_kernel_ms(self):
        return lib().rsk_ctx_last_kernel_ms(self.h)

    def path_counters(self, reset=False):
        """rsk_path_counters_read -> dict (+ swqp_clock_ghz, scored_frac); {} with a library that predates the call (RSK_LIB)"""
        if not hasattr(lib(), "rsk_path_counters_read") or lib().rsk_path_counters_read.argtypes is None:
            return {}
        c = PathCounters()
        c.struct_size = C.sizeof(PathCounters)
        _check(lib().rsk_path_counters_read(self.h, C.byref(c)))
        d = {k: getattr(c, k) for k, _ in PathCounters._fields_ if k != "struct_size"}
        d["swqp_clock_ghz"] = (d["swqp_cycles"] / d["swqp_ref_ticks"] * 0.1) if d["swqp_ref_ticks"] else None
        d["scored_frac"] = (d["sw_pairs_scored"] / d["sw_pairs"]) if d["sw_pairs"] else None
        if reset:
            _check(lib().rsk_path_counters_reset(self.h))
        return d

    def path_counters_reset(self):
        if hasattr(lib(), "rsk_path_counters_reset") and lib().rsk_path_counters_reset.argtypes is not None:
            _check(lib().rsk_path_counters_reset(self.h))

    def close(self):
        if self.h:
            lib().rsk_ctx_destroy(self.h)
            self.h = None

    # ---- D1 gapless -------------------------------------------------------------------------
    def mu_gapless_matrix_dev(self, q, t, self_triangle, d_scores_ptr, ldo):
        _check(lib().rsk_mu_gapless_matrix_dev(self.h, q.h, t.h, int(self_triangle), C.c_void_p(d_scores_ptr), ldo))

    def mu_gapless_hits_dev(self, q, t, self_triangle, min_score, rec_ptr, capacity, count_ptr, d_scores_ptr=0, ldo=0, q_base=0, t_base=0):
        """rsk_mu_gapless_hits_dev: device-side hit records {q, t, score} (uint32 x 3) for scores >= min_score; the dense
        matrix is optional (d_scores_ptr = 0: not written)."""
        _check(lib().rsk_mu_gapless_hits_dev(self.h, q.h, t.h, int(self_triangle), C.c_void_p(d_scores_ptr) if d_scores_ptr else None, ldo,
                                             int(min_score), int(q_base), int(t_base), C.c_void_p(rec_ptr), int(capacity), C.c_void_p(count_ptr)))

    def mu_gapless_shard_window(self, db, shard_index, shard_count):
        lo, hi = C.c_uint32(), C.c_uint32()
        _check(lib().rsk_mu_gapless_shard_window(db.h, int(shard_index), int(shard_count), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def mu_gapless_hits_window_dev(self, db, pos_lo, pos_hi, min_score, rec_ptr, capacity, count_ptr, d_scores_ptr=0, ldo=0, base=0):
        """rsk_mu_gapless_hits_window_dev: one rank's window of the self-search triangle (the whole set on every rank)"""
        _check(lib().rsk_mu_gapless_hits_window_dev(self.h, db.h, int(pos_lo), int(pos_hi), C.c_void_p(d_scores_ptr) if d_scores_ptr else None, ldo,
                                                    int(min_score), int(base), C.c_void_p(rec_ptr), int(capacity), C.c_void_p(count_ptr)))

    # ---- RCCL gather of hit records at the C-ABI (rsk_comm.hip) -----------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().rsk_comm_unique_id(buf))
        return buf.raw

    def comm_create(self, uid, rank, world):
        h = C.c_void_p()
        _check(lib().rsk_comm_create(self.h, uid, int(rank), int(world), C.byref(h)))
        return h

    @staticmethod
    def comm_destroy(comm):
        lib().rsk_comm_destroy(comm)

    @staticmethod
    def gather_hits(comm, d_local_ptr, n_local, rec_bytes, world):
        """-> (device pointer of the gathered records, total count, per-rank counts)"""
        d_all, n_all = C.c_void_p(), C.c_uint64()
        counts = (C.c_uint64 * world)()
        _check(lib().rsk_gather_hits(comm, C.c_void_p(d_local_ptr) if d_local_ptr else None, int(n_local), int(rec_bytes), C.byref(d_all), C.byref(n_all), counts))
        return d_all.value, n_all.value, list(counts)

    def mu_gapless_pairs(self, q, t, iq, it, positions=False):
        iq = np.ascontiguousarray(iq, np.uint32)
        it = np.ascontiguousarray(it, np.uint32)
        n = len(iq)
        sc = np.zeros(n, np.int32)
        bi = np.zeros(n, np.uint32) if positions else None
        bj = np.zeros(n, np.uint32) if positions else None
        _check(lib().rsk_mu_gapless_pairs(self.h, q.h, t.h, _p(iq, u32p), _p(it, u32p), n, _p(sc, i32p),
                                          _p(bi, u32p), _p(bj, u32p)))
        return (sc, bi, bj) if positions else sc

    # ---- P4 Mu SW filter ---------------------------------------------------------------------
    def mu_sw_matrix_dev(self, q, t, self_triangle, reverse_query, d_scores_ptr, ldo, gap_open=2, gap_ext=1):
        _check(lib().rsk_mu_sw_matrix_dev(self.h, q.h, t.h, int(self_triangle), int(reverse_query), gap_open, gap_ext,
                                          C.c_void_p(d_scores_ptr), ldo))

    def mu_filter_dev(self, q, t, self_triangle, omega, omega_fwd, d_fwd, ldo, d_pq, d_pt, d_pf, d_pr, capacity, d_n,
                      gap_open=2, gap_ext=1):
        _check(lib().rsk_mu_filter_dev(self.h, q.h, t.h, int(self_triangle), gap_open, gap_ext, omega, omega_fwd,
                                       C.c_void_p(d_fwd), ldo, C.c_void_p(d_pq), C.c_void_p(d_pt), C.c_void_p(d_pf),
                                       C.c_void_p(d_pr), capacity, C.c_void_p(d_n)))

    def mu_filter_window_dev(self, db, rank_lo, rank_hi, omega, omega_fwd, d_fwd, ldo, d_pq, d_pt, d_pf, d_pr, capacity, d_n, gap_open=2, gap_ext=1):
        """one shard of the self-search triangle: the pairs whose longer member stands at positions [rank_lo, rank_hi) of the length order"""
        _check(lib().rsk_mu_filter_window_dev(self.h, db.h, int(rank_lo), int(rank_hi), gap_open, gap_ext, omega, omega_fwd, C.c_void_p(d_fwd), ldo,
                                              C.c_void_p(d_pq), C.c_void_p(d_pt), C.c_void_p(d_pf), C.c_void_p(d_pr), capacity, C.c_void_p(d_n)))

    def len_rank(self, db):
        r = np.zeros(db.n if hasattr(db, "n") else len(db.lengths), np.uint32)
        _check(lib().rsk_len_rank(db.h, _p(r, u32p)))
        return r

    def pairs_sort_dev(self, d_major, d_minor, n, major_bound=0):
        """sorts the device pair list (two uint32 columns, device pointers) by (major, minor) in place"""
        _check(lib().rsk_pairs_sort_dev(self.h, C.c_void_p(d_major), C.c_void_p(d_minor), n, major_bound))

    def triples_sort_dev(self, d_q, d_t, d_score, n, d_keys):
        """prefilter triples (three uint32 device columns) -> n ascending uint64 keys query << 48 | target << 16 | score"""
        _check(lib().rsk_triples_sort_dev(self.h, C.c_void_p(d_q), C.c_void_p(d_t), C.c_void_p(d_score), n, C.c_void_p(d_keys)))

    def mu_filter_last_work(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().rsk_mu_filter_last_work(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- P5/P6/P7 main alignment ---------------------------------------------------------------
    def align_pairs(self, a, b, ia, ib, min_fwd_score=7.0, gap_open=GAP_OPEN, gap_ext=GAP_EXT, collect=True):
        """-> list of (Aln, path str); collect=False: run the call, return None (timing)"""
        ia = np.ascontiguousarray(ia, np.uint32)
        ib = np.ascontiguousarray(ib, np.uint32)
        n = len(ia)
        nbytes = lib().rsk_align_paths_bytes(a.h, b.h, _p(ia, u32p), _p(ib, u32p), n)
        buf = C.create_string_buffer(max(1, nbytes))
        out = (Aln * max(1, n))()
        _check(lib().rsk_align_pairs(self.h, a.h, b.h, _p(ia, u32p), _p(ib, u32p), n, gap_open, gap_ext, min_fwd_score,
                                     out, buf, nbytes))
        if not collect:
            return None
        raw = buf.raw
        return [(out[k], raw[out[k].path_off:out[k].path_off + out[k].path_len].decode()) for k in range(n)]

    def align_last_times(self):
        """-> (sw_ms, traceback_ms, stats_ms) of the last align_pairs call"""
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _check(lib().rsk_align_last_times(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def align_last_work(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().rsk_align_last_work(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- P1/P2/P13 search driver -------------------------------------------------------------------
    def search_rskdb(self, query, out_tsv, mode, db=None, columns=None, evalue=-1.0, noself=False):
        n = C.c_uint64()
        st = (C.c_uint64 * 8)()
        _check(lib().rsk_search_rskdb(self.h, query.encode(), db.encode() if db else None, mode.encode(),
                                      columns.encode() if columns else None, evalue, int(noself), out_tsv.encode(),
                                      C.byref(n), st))
        return n.value, list(st)

    @staticmethod
    def _opts(mode, kw):
        o = SearchOpts()
        o.struct_size = C.sizeof(SearchOpts)
        o.mode = mode.encode()
        for k, v in kw.items():
            if k in ("columns", "dbmu", "devices"):
                setattr(o, k, v.encode() if v else None)
            elif k in ("evalue", "mints", "pvalue"):
                setattr(o, k, float(v)); setattr(o, k + "_set", 1)
            else:
                setattr(o, k, int(v))
        return o

    def bca_to_rskdb(self, bca, out_rskdb, mode, shard_index=0, shard_count=1, query_flavour=False, **kw):
        """featurise shard shard_index / shard_count of a .bca file (chains by residues) into an RSKDB1 container -> chains written"""
        o = self._opts(mode, kw)
        n = C.c_uint64()
        _check(lib().rsk_bca_to_rskdb(self.h, bca.encode(), int(shard_index), int(shard_count), C.byref(o), int(bool(query_flavour)), out_rskdb.encode(), C.byref(n)))
        return n.value

    def fast_shard_open(self, query, db, **kw):
        """-search -fast -db, stage 1 on target shard shard_index of shard_count -> FastShard (local top-B candidates)"""
        o = self._opts("fast", kw)
        h = C.c_void_p()
        _check(lib().rsk_fast_shard_open(self.h, query.encode(), db.encode(), C.byref(o), C.byref(h)))
        return FastShard(h)

    def search(self, query, out_tsv, mode, db=None, **kw):
        """rsk_search with the options struct: columns, evalue, mints, pvalue, noself, selfrev0, idx_mode, rsb_size, dbmu, keeptmp,
        shard_index / shard_count (one process per GPU), devices ("0,1,...": one process, several devices)."""
        o = self._opts(mode, kw)
        n = C.c_uint64()
        st = (C.c_uint64 * 8)()
        _check(lib().rsk_search(self.h, query.encode(), db.encode() if db else None, C.byref(o), out_tsv.encode(), C.byref(n), st))
        return n.value, list(st)

    # ---- P10-P12 k-mer prefilter ---------------------------------------------------------------------
    def xdrop_pairs(self, a, b, ia, ib, lo_a, lo_b, X, gap_open, gap_ext):
        """rsk_xdrop_pairs -> list of (score_fwd, fwd_path, score_bwd, bwd_path)"""
        ia = np.ascontiguousarray(ia, np.uint32)
        ib = np.ascontiguousarray(ib, np.uint32)
        lo_a = np.ascontiguousarray(lo_a, np.uint32)
        lo_b = np.ascontiguousarray(lo_b, np.uint32)
        n = len(ia)
        la, lb = np.asarray(a.lengths), np.asarray(b.lengths)
        nbytes = int((la[ia].astype(np.int64) + lb[ib] + 4).sum()) + 16
        buf = C.create_string_buffer(nbytes)
        sf, sb = np.zeros(n, np.float32), np.zeros(n, np.float32)
        fo, bo = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        fl, bl = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        u64p = C.POINTER(C.c_uint64)
        _check(lib().rsk_xdrop_pairs(self.h, a.h, b.h, _p(ia, u32p), _p(ib, u32p), _p(lo_a, u32p), _p(lo_b, u32p), n, X, gap_open, gap_ext,
                                     _p(sf, f32p), _p(sb, f32p), buf, nbytes, _p(fo, u64p), _p(fl, u32p), _p(bo, u64p), _p(bl, u32p)))
        raw = buf.raw
        return [(float(sf[k]), raw[int(fo[k]):int(fo[k]) + int(fl[k])].decode(), float(sb[k]), raw[int(bo[k]):int(bo[k]) + int(bl[k])].decode())
                for k in range(n)]

    def mkf_align_pairs(self, a, b, ia, ib, hsp_first, hsp_lo_a, hsp_lo_b, hsp_len, x2=8.0, gap_open=GAP_OPEN, gap_ext=GAP_EXT,
                        min_mega_score=-4.0, min_fwd_score=7.0, hsp_score=None):
        """rsk_mkf_align_pairs -> (list of (Aln, path str), status uint8[n]); with hsp_score the lists are the UNCHAINED seed HSPs and
        rsk_mkf_chain_align_pairs chains them on the device first (status 3 = tied chain, left to the caller)"""
        ia = np.ascontiguousarray(ia, np.uint32)
        ib = np.ascontiguousarray(ib, np.uint32)
        hf = np.ascontiguousarray(hsp_first, np.uint32)
        hla, hlb, hl = (np.ascontiguousarray(x, np.int32) for x in (hsp_lo_a, hsp_lo_b, hsp_len))
        n = len(ia)
        la, lb = np.asarray(a.lengths), np.asarray(b.lengths)
        nbytes = int((la[ia].astype(np.int64) + lb[ib] + 1).sum()) + 16
        buf = C.create_string_buffer(nbytes)
        out = (Aln * max(1, n))()
        status = np.zeros(max(1, n), np.uint8)
        if hsp_score is not None:
            hs = np.ascontiguousarray(hsp_score, np.int32)
            _check(lib().rsk_mkf_chain_align_pairs(self.h, a.h, b.h, _p(ia, u32p), _p(ib, u32p), n, _p(hf, u32p), _p(hla, i32p), _p(hlb, i32p),
                                                   _p(hl, i32p), _p(hs, i32p), x2, gap_open, gap_ext, min_mega_score, min_fwd_score, out,
                                                   _p(status, u8p), buf, nbytes))
        else:
            _check(lib().rsk_mkf_align_pairs(self.h, a.h, b.h, _p(ia, u32p), _p(ib, u32p), n, _p(hf, u32p), _p(hla, i32p), _p(hlb, i32p), _p(hl, i32p),
                                             x2, gap_open, gap_ext, min_mega_score, min_fwd_score, out, _p(status, u8p), buf, nbytes))
        raw = buf.raw
        return [(out[k], raw[out[k].path_off:out[k].path_off + out[k].path_len].decode()) for k in range(n)], status[:n]

    def dss_densities(self, lens, x, y, z, W=50, w1=3, w2=8, radius=20.0, eps=1.0, nen_W=100, nen_w=12):
        """rsk_dss_densities -> dict of arrays over the concatenated residues: ss_fwd/ss_rev (uint8 chars), conf_fwd/conf_rev (uint8,
        255 = none), dens_fwd/sdens_fwd/dens_rev/sdens_rev (float64), nen_fwd/ren_fwd/nen_rev/ren_rev (uint32)"""
        lens = np.ascontiguousarray(lens, np.uint32)
        x, y, z = (np.ascontiguousarray(v, np.float32) for v in (x, y, z))
        tot = int(lens.sum())
        b8 = {k: np.zeros(max(tot, 1), np.uint8) for k in ("ss_fwd", "ss_rev", "conf_fwd", "conf_rev")}
        f64 = {k: np.zeros(max(tot, 1), np.float64) for k in ("dens_fwd", "sdens_fwd", "dens_rev", "sdens_rev")}
        u32 = {k: np.zeros(max(tot, 1), np.uint32) for k in ("nen_fwd", "ren_fwd", "nen_rev", "ren_rev")}
        f64p = C.POINTER(C.c_double)
        _check(lib().rsk_dss_densities(self.h, len(lens), _p(lens, u32p), _p(x, f32p), _p(y, f32p), _p(z, f32p),
                                       *[v.ctypes.data for v in b8.values()], W, w1, w2, radius, eps, *[_p(v, f64p) for v in f64.values()],
                                       nen_W, nen_w, *[_p(v, u32p) for v in u32.values()]))
        return {k: v[:tot] for d in (b8, f64, u32) for k, v in d.items()}

    def mkf_seed_pairs(self, q, t, iq, it, x1=8, min_hsp_score=50, cap=16, max_records=None):
        """-> (found uint8[n], {pair index: (nkept, kept int32 [min(nkept, cap), 4])})"""
        iq = np.ascontiguousarray(iq, np.uint32)
        it = np.ascontiguousarray(it, np.uint32)
        n = len(iq)
        mr = max_records if max_records is not None else max(n, 1)
        found = np.zeros(n, np.uint8)
        rp = np.zeros(mr, np.uint32)
        rn = np.zeros(mr, np.uint32)
        rk = np.zeros((mr, cap, 4), np.int32)
        nrec = C.c_size_t()
        _check(lib().rsk_mkf_seed_pairs(self.h, q.h, t.h, iq.ctypes.data_as(u32p), it.ctypes.data_as(u32p), n, x1, min_hsp_score, cap,
                                        found.ctypes.data_as(C.POINTER(C.c_uint8)), mr, C.byref(nrec), rp.ctypes.data_as(u32p),
                                        rn.ctypes.data_as(u32p), rk.ctypes.data_as(C.POINTER(C.c_int32))))
        assert nrec.value <= mr
        recs = {int(rp[r]): (int(rn[r]), rk[r, :min(int(rn[r]), cap)].copy()) for r in range(nrec.value)}
        return found, recs

    def mu_pinop_pairs(self, q, t, iq, it, open_=-2, ext=-1):
        iq = np.ascontiguousarray(iq, np.uint32)
        it = np.ascontiguousarray(it, np.uint32)
        out = np.zeros(len(iq), np.int32)
        _check(lib().rsk_mu_pinop_pairs(self.h, q.h, t.h, iq.ctypes.data_as(u32p), it.ctypes.data_as(u32p), len(iq), open_, ext,
                                        out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def mu_gapless_profb_pairs(self, q, t, iq, it):
        iq = np.ascontiguousarray(iq, np.uint32)
        it = np.ascontiguousarray(it, np.uint32)
        out = np.zeros(len(iq), np.float32)
        _check(lib().rsk_mu_gapless_profb_pairs(self.h, q.h, t.h, iq.ctypes.data_as(u32p), it.ctypes.data_as(u32p), len(iq),
                                                out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def gapless_float_pairs(self, a, b, ia, ib):
        ia = np.ascontiguousarray(ia, np.uint32)
        ib = np.ascontiguousarray(ib, np.uint32)
        sc = np.zeros(len(ia), np.float32)
        bi = np.zeros(len(ia), np.uint32)
        bj = np.zeros(len(ia), np.uint32)
        _check(lib().rsk_gapless_float_pairs(self.h, a.h, b.h, ia.ctypes.data_as(u32p), ib.ctypes.data_as(u32p), len(ia),
                                             sc.ctypes.data_as(C.POINTER(C.c_float)), bi.ctypes.data_as(u32p), bj.ctypes.data_as(u32p)))
        return sc, bi, bj

    def mu_filter_pairs(self, q, t, iq, it, omega, omega_fwd, gap_open=2, gap_ext=1):
        """-> (pass uint8[n], fwd int32[n], rev int32[n]) for host pair lists."""
        iq = np.ascontiguousarray(iq, np.uint32)
        it = np.ascontiguousarray(it, np.uint32)
        n = len(iq)
        ok = np.zeros(n, np.uint8)
        fwd = np.zeros(n, np.int32)
        rev = np.zeros(n, np.int32)
        _check(lib().rsk_mu_filter_pairs(self.h, q.h, t.h, iq.ctypes.data_as(u32p), it.ctypes.data_as(u32p), n, gap_open, gap_ext,
                                         omega, omega_fwd, ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                         fwd.ctypes.data_as(C.POINTER(C.c_int32)), rev.ctypes.data_as(C.POINTER(C.c_int32))))
        return ok, fwd, rev

    def mu_prefilter_dev(self, q, t, d_q, d_t, d_score, capacity, d_n, neighbourhood=0):
        _check(lib().rsk_mu_prefilter_dev(self.h, q.h, t.h, neighbourhood, C.c_void_p(d_q), C.c_void_p(d_t), C.c_void_p(d_score),
                                          capacity, C.c_void_p(d_n)))

    def mu_prefilter_last_work(self):
        """(seed items, index postings, two-hit diagonals, diagonal cells scored) of the last mu_prefilter_dev call"""
        v = [C.c_uint64() for _ in range(4)]
        _check(lib().rsk_mu_prefilter_last_work(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def mu_gapless_last_work(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().rsk_mu_gapless_last_work(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


class FastShard:
    """Handle of rsk_fast_shard_open.  Exact exchange: triples() -> int32 [n, 3] (query, global target, score) of the whole
    shard, finish(all ranks' rows, exact=True).  Top-B exchange: candidates() / finish(rows) (own tie rule at the cut)."""

    def __init__(self, h):
        self.h = h

    def _rows(self, fn):
        q, t, s = u32p(), u32p(), u32p()
        n = C.c_size_t()
        _check(fn(self.h, C.byref(q), C.byref(t), C.byref(s), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 3), np.int32)
        f = lambda p: np.ctypeslib.as_array(p, shape=(n.value,)).astype(np.int32)      # noqa: E731
        return np.stack([f(q), f(t), f(s)], axis=1)

    def candidates(self):
        return self._rows(lib().rsk_fast_shard_candidates)

    def triples(self):
        return self._rows(lib().rsk_fast_shard_triples)

    def finish(self, rows, out_tsv, tmp_tsv=None, exact=False):
        rows = np.ascontiguousarray(rows, np.int32).reshape(-1, 3)
        q = np.ascontiguousarray(rows[:, 0], np.uint32)
        t = np.ascontiguousarray(rows[:, 1], np.uint32)
        s = np.ascontiguousarray(rows[:, 2], np.uint32)
        nh = C.c_uint64()
        st = (C.c_uint64 * 8)()
        fn = lib().rsk_fast_shard_finish_exact if exact else lib().rsk_fast_shard_finish
        _check(fn(self.h, _p(q, u32p), _p(t, u32p), _p(s, u32p), len(q), out_tsv.encode(), tmp_tsv.encode() if tmp_tsv else None, C.byref(nh), st))
        return nh.value, list(st)

    def close(self):
        if self.h:
            lib().rsk_fast_shard_close(self.h)
            self.h = None


def rsb_merge(rows, nqueries, rsb_size):
    """rsk_rsb_merge: per query the rsb_size best (score desc, target asc) of int32 [n, 3] (query, target, score) rows"""
    rows = np.ascontiguousarray(rows, np.int32).reshape(-1, 3)
    q = np.ascontiguousarray(rows[:, 0], np.uint32)
    t = np.ascontiguousarray(rows[:, 1], np.uint32)
    s = np.ascontiguousarray(rows[:, 2], np.uint32)
    oq, ot, os_ = np.zeros(len(q), np.uint32), np.zeros(len(q), np.uint32), np.zeros(len(q), np.uint32)
    n = C.c_size_t()
    _check(lib().rsk_rsb_merge(_p(q, u32p), _p(t, u32p), _p(s, u32p), len(q), nqueries, rsb_size, _p(oq, u32p), _p(ot, u32p), _p(os_, u32p),
                               C.byref(n)))
    return np.stack([oq[:n.value], ot[:n.value], os_[:n.value]], axis=1).astype(np.int32)


class Db:
    """A chain set uploaded to HBM (rsk_db)."""

    def __init__(self, ctx, lengths, mu=None, prof=None, xyz=None, selfrev=None):
        lengths = np.ascontiguousarray(lengths, np.uint32)
        mu = None if mu is None else np.ascontiguousarray(mu, np.uint8)
        prof = None if prof is None else np.ascontiguousarray(prof, np.uint8)
        x = y = z = None
        if xyz is not None:
            x, y, z = (np.ascontiguousarray(a, np.float32) for a in xyz)
        selfrev = None if selfrev is None else np.ascontiguousarray(selfrev, np.float32)
        h = C.c_void_p()
        _check(lib().rsk_db_create(ctx.h, len(lengths), _p(lengths, u32p), _p(mu, u8p), _p(prof, u8p),
                                   _p(x, f32p), _p(y, f32p), _p(z, f32p), _p(selfrev, f32p), C.byref(h)))
        self.h = h
        self.ctx = ctx
        self.lengths = lengths

    @classmethod
    def from_mu_seqs(cls, ctx, seqs):
        lengths = np.array([len(s) for s in seqs], np.uint32)
        return cls(ctx, lengths, mu=np.concatenate(seqs).astype(np.uint8))

    @classmethod
    def from_chains(cls, ctx, chains):
        """chains: objects with .mu, .prof [8,L], .x/.y/.z, .selfrev (tests/fixtures.py Chain)."""
        lengths = np.array([len(c.mu) for c in chains], np.uint32)
        mu = np.concatenate([c.mu for c in chains])
        prof = np.concatenate([np.ascontiguousarray(c.prof).reshape(-1) for c in chains])
        xyz = tuple(np.concatenate([getattr(c, k) for c in chains]) for k in "xyz")
        selfrev = np.array([c.selfrev for c in chains], np.float32)
        return cls(ctx, lengths, mu=mu, prof=prof, xyz=xyz, selfrev=selfrev)

    @property
    def n(self):
        return lib().rsk_db_nchains(self.h)

    def close(self):
        if self.h:
            lib().rsk_db_destroy(self.h)
            self.h = None


def rsb_select_keys(keys, nqueries, rsb_size=1500, tmp_tsv_path=None):
    """rsk_rsb_select_keys: keys = query << 48 | target << 16 | score, ascending (rsk_triples_sort_dev) -> (q, t, score) kept"""
    keys = np.ascontiguousarray(keys, np.uint64)
    n = len(keys)
    oq, ot, os_ = (np.zeros(max(n, 1), np.uint32) for _ in range(3))
    nout = C.c_size_t()
    _check(lib().rsk_rsb_select_keys(keys.ctypes.data_as(C.POINTER(C.c_uint64)), n, nqueries, rsb_size, _p(oq, u32p), _p(ot, u32p),
                                     _p(os_, u32p), C.byref(nout), tmp_tsv_path.encode() if tmp_tsv_path else None))
    m = nout.value
    return oq[:m], ot[:m], os_[:m]


def rsb_select(q, t, score, nqueries, rsb_size=1500, tmp_tsv_path=None):
    """RankedScoresBag on host triples -> (q, t, score) survivors (host C++ in librsk, not Python)."""
    q = np.ascontiguousarray(q, np.uint32)
    t = np.ascontiguousarray(t, np.uint32)
    score = np.ascontiguousarray(score, np.uint32)
    n = len(q)
    oq = np.zeros(max(n, 1), np.uint32)
    ot = np.zeros(max(n, 1), np.uint32)
    os_ = np.zeros(max(n, 1), np.uint32)
    nout = C.c_size_t()
    _check(lib().rsk_rsb_select(_p(q, u32p), _p(t, u32p), _p(score, u32p), n, nqueries, rsb_size, _p(oq, u32p), _p(ot, u32p),
                                _p(os_, u32p), C.byref(nout), tmp_tsv_path.encode() if tmp_tsv_path else None))
    m = nout.value
    return oq[:m], ot[:m], os_[:m]


def _xdrop(fn, ctx, S, X, gap_open, gap_ext, a, b):
    S = np.ascontiguousarray(S, np.float32)
    LA, LB = S.shape
    buf = C.create_string_buffer(LA + LB + 2)
    score, n = C.c_float(), C.c_uint32()
    _check(fn(ctx.h, _p(S, f32p), LA, LB, X, gap_open, gap_ext, a, b, C.byref(score), buf, len(buf), C.byref(n)))
    return score.value, buf.value.decode()


def xdrop_fwd(ctx, S, X, gap_open, gap_ext, lo_a, lo_b):
    """XDropFwd (xdropfwd.cpp:71) on an explicit score matrix S[LA, LB] -> (score, path): one extension on the device
    (k_xdrop_wave with the matrix in place of the profile tables)."""
    return _xdrop(lib().rsk_xdrop_fwd, ctx, S, X, gap_open, gap_ext, lo_a, lo_b)


def xdrop_bwd(ctx, S, X, gap_open, gap_ext, hi_a, hi_b):
    """XDropBwd (xdropbwd.cpp:28) -> (score, path), on the device."""
    return _xdrop(lib().rsk_xdrop_bwd, ctx, S, X, gap_open, gap_ext, hi_a, hi_b)


def merge_fwd_bwd(LA, LB, fwd_lo_a, fwd_lo_b, fwd_path, bwd_hi_a, bwd_hi_b, bwd_path):
    """MergeFwdBwd (mergefwdback.cpp:6) -> (lo_a, lo_b, hi_a, hi_b, path)."""
    v = [C.c_uint32() for _ in range(4)]
    buf = C.create_string_buffer(len(fwd_path) + len(bwd_path) + 2)
    n = C.c_uint32()
    _check(lib().rsk_merge_fwd_bwd(LA, LB, fwd_lo_a, fwd_lo_b, fwd_path.encode(), bwd_hi_a, bwd_hi_b, bwd_path.encode(),
                                   C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(v[3]), buf, len(buf), C.byref(n)))
    return v[0].value, v[1].value, v[2].value, v[3].value, buf.value.decode()


def dss_densities_host(x, y, z):
    """DSS::GetDensity / GetSSDensity(Pos, 's') of one chain with the host's libm exp -> (dens, sdens) float64 [L]"""
    x, y, z = (np.ascontiguousarray(v, np.float32) for v in (x, y, z))
    L = len(x)
    d, s = np.zeros(L, np.float64), np.zeros(L, np.float64)
    f64p = C.POINTER(C.c_double)
    _check(lib().rsk_dss_densities_host(_p(x, f32p), _p(y, f32p), _p(z, f32p), L, _p(d, f64p), _p(s, f64p)))
    return d, s


def shard_range(kind, lengths, index, count):
    """rsk_shard_range: kind 0 = self-search targets balanced by DP cells, 1 = chain range balanced by residues -> (lo, hi)"""
    L = np.ascontiguousarray(lengths, np.uint32)
    lo, hi = C.c_uint64(), C.c_uint64()
    _check(lib().rsk_shard_range(kind, _p(L, u32p), len(L), index, count, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def dss_featurize(seq, x, y, z):
    """DSS::GetProfile + GetMuLetters of one chain -> (prof uint8 [8, L], mu uint8 [L]).  Host code."""
    L = len(seq)
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    z = np.ascontiguousarray(z, np.float32)
    prof = np.zeros((8, L), np.uint8)
    mu = np.zeros(L, np.uint8)
    s = seq.encode() if isinstance(seq, str) else bytes(seq)
    _check(lib().rsk_dss_featurize(s, _p(x, f32p), _p(y, f32p), _p(z, f32p), L, _p(prof, u8p), _p(mu, u8p)))
    return prof, mu


def dss_featurize_reversed(seq, x, y, z):
    """Profile of the reversed chain (target side of GetSelfRevScore) given the un-reversed chain -> uint8 [8, L]."""
    L = len(seq)
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    z = np.ascontiguousarray(z, np.float32)
    prof = np.zeros((8, L), np.uint8)
    s = seq.encode() if isinstance(seq, str) else bytes(seq)
    _check(lib().rsk_dss_featurize_reversed(s, _p(x, f32p), _p(y, f32p), _p(z, f32p), L, _p(prof, u8p)))
    return prof


def bca_info(path):
    n, r = C.c_uint64(), C.c_uint64()
    ml, mlab = C.c_uint32(), C.c_uint32()
    _check(lib().rsk_bca_info(path.encode(), C.byref(n), C.byref(r), C.byref(ml), C.byref(mlab)))
    return n.value, r.value, ml.value, mlab.value


def bca_read_chain(path, idx, cap=70000):
    """-> (label, seq, x, y, z) of chain idx of a .bca file."""
    lab = C.create_string_buffer(1024)
    seq = C.create_string_buffer(cap + 1)
    x = np.zeros(cap, np.float32)
    y = np.zeros(cap, np.float32)
    z = np.zeros(cap, np.float32)
    L = C.c_uint32()
    _check(lib().rsk_bca_read_chain(path.encode(), idx, lab, 1024, seq, _p(x, f32p), _p(y, f32p), _p(z, f32p), cap, C.byref(L)))
    n = L.value
    return lab.value.decode(), seq.raw[:n].split(b"\0")[0].decode(), x[:n].copy(), y[:n].copy(), z[:n].copy()


def bca_copy(src, dst):
    _check(lib().rsk_bca_copy(src.encode(), dst.encode()))


def bca_to_mu_fasta(src, dst):
    _check(lib().rsk_bca_to_mu_fasta(src.encode(), dst.encode()))

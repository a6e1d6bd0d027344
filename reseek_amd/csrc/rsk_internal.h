// rsk_internal.h -- shared host-side definitions of librsk.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/reseek_amd.h"

#define RSK_MU_NULL 36          // pad letter: its profile row resets every diagonal (score -32768)
#define RSK_CHAIN_PAD 16        // chains are padded to a multiple of 16 residues in HBM

void rsk_set_error(const char *fmt, ...);
int rsk_hip_fail(hipError_t e, const char *what, const char *file, int line);

#define RSK_HIP(call)                                                          \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return rsk_hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

struct rsk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_wait = nullptr;   // blocking-sync event of rsk_stream_wait (created on first use)
    hipStream_t aux = nullptr;      // side stream of the gapless launch (the per-pair kernel of chains beyond a ring runs beside the ring kernels)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // rsk_align_pairs' own events (created together on first use): before / after the SW kernels, after the traceback kernel, after
    // the statistics kernels; al_times_valid = all four were recorded by the context's last rsk_align_pairs call
    hipEvent_t ev_al0 = nullptr, ev_al1 = nullptr, ev_tb = nullptr, ev_st = nullptr;
    bool al_times_valid = false;
    uint64_t al_steps = 0, al_col_tests = 0;       // last rsk_align_pairs call with statistics: path characters walked, LDDT column-pair tests
    float last_ms = -1.0f;
    // accounting of the last gapless matrix call
    uint64_t gl_pairs = 0, gl_cells = 0, gl_slots = 0;
    uint64_t mf_pairs = 0, mf_candidates = 0;   // last Mu filter call
    uint64_t al_pairs = 0, al_cells = 0, al_tb_bytes = 0;   // last rsk_align_pairs call
    uint64_t pf_hits = 0, pf_postings = 0, pf_twohit = 0, pf_cells = 0;   // last rsk_mu_prefilter_dev call: seed items, index size, two-hit diagonals, cells scored
    int num_cus = 0;
    // caching device allocator (hipMalloc/hipFree cost ~0.1-1 ms each and synchronise; the batch
    // entry points need a dozen temporaries per call): blocks are kept in size classes until the
    // context is destroyed.
    std::map<void *, size_t> pool_live;
    std::multimap<size_t, void *> pool_free;
    uint64_t pool_bytes = 0;
    // grow-only pinned host staging buffers (slot 0: host->device blob, 1: device->host results)
    void *pin[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t pin_bytes[4] = { 0, 0, 0, 0 };
};

// One-time per-DEVICE initialisation (constant-table uploads, hipFuncSetAttribute: both belong to the device's copy of the
// code object), safe against the host threads that drive several contexts at once: `flags` is a zero-initialised static
// array of 64 atomics; returns f()'s code.
#include <atomic>
#include <mutex>
template <class F>
static inline int rsk_once_per_device(std::atomic<int> *flags, int device, F f)
{
    std::atomic<int> &fl = flags[device & 63];
    if (fl.load(std::memory_order_acquire)) return RSK_OK;
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if (fl.load(std::memory_order_relaxed)) return RSK_OK;
    const int rc = f();
    if (rc == RSK_OK) fl.store(1, std::memory_order_release);
    return rc;
}

// device temporaries of one call, returned to the context's pool on every exit path
struct rsk_scratch {
    rsk_ctx *ctx;
    std::vector<void *> all;
    explicit rsk_scratch(rsk_ctx *c) : ctx(c) {}
    rsk_scratch(const rsk_scratch &) = delete;
    rsk_scratch &operator=(const rsk_scratch &) = delete;
    ~rsk_scratch();
    int alloc(void **p, size_t bytes);
    template <class T> int alloc(T **p, size_t count) { return alloc((void **) p, count * sizeof(T)); }
};

int rsk_pool_alloc(rsk_ctx *ctx, void **p, size_t bytes);
// hipMalloc on the current device with the out-of-memory ladder every device allocation of the library goes through:
// on failure (1) the cached blocks of `ctx` (may be NULL) are released, (2) the host layer's hook runs -- it destroys the
// idle helper contexts parked for this device, whose pools can hold tens of GB of scratch (host/dbsearcher.cpp
// SecondaryCtx) -- and the allocation is tried again after each step.  Sets the error text and returns RSK_E_NOMEM.
int rsk_dev_malloc(rsk_ctx *ctx, void **p, size_t bytes);
void rsk_set_oom_hook(void (*release_idle)(int device));
// Makes `device` current for the calling thread and puts the previous one back on scope exit: library code that must
// allocate on a particular device (a chain set's arrays, a helper context) does not leave the caller's thread on it.
struct rsk_device_guard {
    int prev = -1, cur = -1;
    explicit rsk_device_guard(int device) : cur(device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void) hipSetDevice(device);
    }
    rsk_device_guard(const rsk_device_guard &) = delete;
    rsk_device_guard &operator=(const rsk_device_guard &) = delete;
    ~rsk_device_guard() { if (prev >= 0 && prev != cur) (void) hipSetDevice(prev); }
};
// An array that belongs to a chain set (lazily built caches, rsk_db_set_seq): allocated on the SET's device whatever the
// calling thread's current device is (the kernels that read it run on db->ctx->device), through the same OOM ladder.
// `caller` = the context of the calling thread when it is on that device (its pool may be released), else NULL.
int rsk_db_malloc(const struct rsk_db *db, rsk_ctx *caller, void **p, size_t bytes);
// Waits for the context's stream without spinning: the long waits (tens of ms of kernels) of several host threads would
// otherwise each burn a core of a CPU-quota'd container.
int rsk_stream_wait(rsk_ctx *ctx);
int rsk_db_update_selfrev(rsk_db *db, const float *selfrev);   // completes a set uploaded before its self-rev scores were known
void rsk_pool_free(rsk_ctx *ctx, void *p);
void rsk_pool_release(rsk_ctx *ctx);
int rsk_pinned(rsk_ctx *ctx, int slot, size_t bytes, void **p);

// Process-wide counters behind rsk_path_counters_read (search_api.cpp); the k_sw_qp clock words live in device memory, two
// uint64 per device (k_sw_float.hip: rsk_swqp_clock_words).
struct rsk_host_counters {
    std::atomic<uint64_t> sw_pairs{0}, sw_pairs_scored{0}, sw_pairs_rescored{0}, upload_copies{0}, upload_bytes{0}, db_batches{0};
    std::atomic<uint64_t> loader_ns{0}, featurise_ns{0}, upload_ns{0};
};
extern rsk_host_counters g_rsk_counters;
unsigned long long *rsk_swqp_clock_words(int device);      // device pointer to {cycles, ticks}; nullptr if the allocation failed

// rsk_db_create's internal form: the chains arrive through copy functions (chain i's Mu letters / feature row f / coordinate axis
// -> dst[length of chain i]); a null function = the set has no such array.  Called from several host threads at once.
struct rsk_chain_source {
    std::function<void(uint32_t, uint8_t *)> mu;
    std::function<void(uint32_t, int, uint8_t *)> prof;
    std::function<void(uint32_t, int, float *)> xyz;
};
int rsk_db_create_from(rsk_ctx *ctx, uint32_t n, const uint32_t *lengths, const rsk_chain_source &from, const float *selfrev, struct rsk_db **out);

// One "ring" of the gapless kernel: several query chains laid out on a circular array of
// 128*D diagonal slots (see k_mu_gapless.hip).
struct rsk_ring {
    uint32_t D;            // dwords of diagonal state per lane (ring = 128*D slots)
    uint32_t nq;           // queries in the ring
    uint32_t min_q;        // smallest query index (triangle skipping)
    uint32_t letters_off;  // byte offset into d_ring_letters (128*D bytes; 0xFF = separator row)
    uint32_t laneq_off;    // offset into d_ring_laneq ((D/4)*64 bytes: local query of each lane group; 0xFF none)
    uint32_t qid_off;      // offset into d_ring_qid (nq uint32: global query index)
};

#define RSK_GL_WORK_ENTRIES 16

struct rsk_db {
    rsk_ctx *ctx = nullptr;
    uint64_t uid = 0;          // unique per rsk_db_create (cache keys)
    uint32_t n = 0;
    uint64_t nres = 0;         // sum of lengths
    uint64_t npad = 0;         // sum of padded lengths
    std::vector<uint32_t> len; // host copies
    std::vector<uint32_t> off; // n+1 padded residue offsets
    std::vector<uint8_t> h_mu; // padded host copy (ring construction)
    uint32_t *d_len = nullptr;
    uint32_t *d_off = nullptr;
    uint8_t *d_mu = nullptr;   // npad bytes, pad = RSK_MU_NULL
    uint8_t *d_prof = nullptr; // [RSK_NFEAT][npad]
    uint16_t *d_prof_cb = nullptr; // [npad + 64][8]: per residue, letter*4 of each feature (float SW column offsets)
    uint16_t *d_prof_ra = nullptr; // [npad + 64][8]: letter*alphabet*4 (table row offsets, transposed float SW)
    std::vector<float> h_selfrev;
    float *d_x = nullptr, *d_y = nullptr, *d_z = nullptr;
    float *d_selfrev = nullptr;
    uint8_t *d_seq = nullptr;  // npad bytes: residue characters (optional, rsk_db_set_seq), chains at off[] like d_mu
    uint64_t hbm_bytes = 0;
    // gapless ring cache (built lazily when the chain set is used as the query side)
    bool rings_built = false;
    std::vector<rsk_ring> rings;        // sorted by D
    std::vector<uint32_t> ring_first;   // first ring index per D class
    rsk_ring *d_ring_tab = nullptr;
    uint8_t *d_ring_letters = nullptr;
    uint8_t *d_ring_laneq = nullptr;
    uint32_t *d_ring_qid = nullptr;
    uint32_t *d_ring_perm = nullptr;    // processing order of the chains in self-triangle mode (ring members, then long chains)
    std::vector<uint32_t> h_ring_perm;
    // claim orders of the gapless kernel, one per target-block size `tb` the set has been used with (the block size depends
    // on the OTHER operand of a call): per tb-position block of ring_perm (tri) / of the natural chain order (nat), the
    // positions by decreasing chain length.  Built under claim_mutex, never freed while the set lives -- two contexts that use
    // the set with different partners cannot free an array the other one has handed to a launch.
    std::map<uint32_t, uint32_t *> tri_claims, nat_claims;
    std::mutex claim_mutex;
    // gapless work list cache: a few entries keyed by (target set, triangle flag, block size, window) -- a rank's window, or the
    // N windows a one-GPU prediction walks through, come back call after call.  Looked up, rebuilt and handed to the launches
    // under claim_mutex (two contexts sharing a set with different partners / windows must not free each other's
    // list between the check and the launch; an evicted entry is hipFree'd = after the kernels reading it).
    struct gl_work {
        uint64_t work_for = 0;          // uid of the target set the list was built for (0 = entry invalid)
        int work_tri = -1;
        uint32_t work_tb = 0;           // targets per work item
        uint32_t win_lo = 0, win_hi = 0;
        void *d_work = nullptr;         // uint2 (ring, first target) entries, D = 4 class first
        uint32_t work_count[2] = { 0, 0 };
        uint32_t *d_long_iq = nullptr, *d_long_it = nullptr;   // (long query, target) pairs of the per-pair kernel
        uint32_t long_pairs = 0;
        uint64_t last_use = 0;
    };
    std::vector<gl_work> work_cache;    // <= RSK_GL_WORK_ENTRIES
    uint64_t work_clock = 0;
    // chains by increasing length (Mu SW filter: the targets a wave works on at once have similar lengths)
    uint32_t *d_len_perm = nullptr;     // perm[k] = chain index of rank k
    uint32_t *d_len_rank = nullptr;     // rank[chain]
    std::vector<uint32_t> h_len_rank;
    // k-mer prefilter index (built lazily when the chain set is the query side)
    bool mudex_built = false;
    int mudex_mode = -1;                // 0 exact k-mers, 1 idxq, 2 idxt-equivalent (k_prefilter.hip)
    void *d_pf_table = nullptr;         // uint2 [36^5] (start, count)
    uint32_t *d_pf_postings = nullptr;  // q << 16 | pos
    size_t pf_postings = 0;
    std::vector<uint32_t> long_q;       // queries too long for a ring (handled by the per-pair kernel)
    uint64_t ring_slots_total = 0;      // sum of 128*D over rings
};

// kernels (k_mu_gapless.hip)
int rsk_launch_gapless_rings(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle,
                             uint16_t *d_scores, size_t ldo, uint32_t min_score, uint32_t q_base, uint32_t t_base, uint32_t *d_rec,
                             uint32_t capacity, uint32_t *d_count, uint32_t win_lo = 0, uint32_t win_hi = 0xFFFFFFFFu);
int rsk_launch_gapless_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *d_iq,
                             const uint32_t *d_it, size_t npairs, int32_t *d_scores,
                             uint32_t *d_besti, uint32_t *d_bestj);
int rsk_build_rings(rsk_db *db);
int rsk_build_mudex(rsk_ctx *ctx, rsk_db *db, int mode);
int rsk_build_len_perm(rsk_db *db);
// k_pairs_sort.hip: stable radix sort of (64-bit key, 32-bit payload) pairs on bits [0, end_bit) of the key, queued on the
// context's stream (temporaries from its pool)
int rsk_sort_pairs_u64_u32(rsk_ctx *ctx, const unsigned long long *d_keys_in, unsigned long long *d_keys_out, const uint32_t *d_vals_in,
                           uint32_t *d_vals_out, size_t n, int end_bit);

// k_sw_float.hip: CalcEvalue + path packing for alignments whose paths already sit on the device (see the definition)
int rsk_paths_stats_pack(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, size_t npairs, const uint32_t *ia, const uint32_t *ib,
                         const uint32_t *d_ia, const uint32_t *d_ib, const char *d_paths, const uint64_t *d_pstart, const uint32_t *d_plen,
                         const uint32_t *d_loa, const uint32_t *d_lob, const float *d_score, float min_fwd_score, rsk_aln *out, char *paths,
                         size_t paths_bytes);

// host worker threads: min(hardware threads, cgroup CPU quota, cap) -- defined in host/dbsearcher.cpp
namespace reseek_amd { unsigned HostThreads(unsigned cap); }

// run body(lo, hi) over [0, n) on the host worker threads (contiguous slices)
template <class F>
static inline void rsk_parallel_for(size_t n, size_t min_per_thread, F body)
{
    const unsigned T = (unsigned) std::min<size_t>(reseek_amd::HostThreads(64), n / (min_per_thread ? min_per_thread : 1) + 1);
    if (T < 2) { body((size_t) 0, n); return; }
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < T; ++t) ts.emplace_back([&, t]() { body(n * t / T, n * (t + 1) / T); });
    for (auto &t : ts) t.join();
}

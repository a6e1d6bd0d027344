// k_prefilter.hip -- Mu k-mer prefilter on gfx950 (P12 in host/prefilter.cpp).
//
// Reference semantics (exact k-mers = what `reseek -prefilter_mu` runs, cmd_prefiltermu.cpp:52):
//   P10 MuDex::FromSeqDB mudex.cpp:386 / GetKmers :517: spaced 5-of-7 k-mers (offsets 0,1,2,5,6, base 36),
//       k-mers whose self score under Mu_S_ij_i8 is < 36 are masked (mermx.cpp:153, prefiltermuparams.h:22)
//   P11 PrefilterMu::Search prefiltermu.cpp:382: per target, every (TPos, posting) pair gives
//       (QIdx, diag = (uint16)(QL + TPos - QPos - 1)), dropped if diag > 16383 (:254); pairs seen >= 2 times
//       are two-hit diagonals (twohitdiag.cpp:368-398); each is scored with FindHSP (:12-48, Kadane over the
//       whole diagonal on Mu_S_ij_i8); per query the best score > 0, clamped to 65534 (:288-313).
//
// MI355X design (HBM/latency bound gather + small sorts; SCOP40 x SCOP40 is 94 M postings hits):
//   * index: postings (q << 16 | pos) sorted by k-mer, plus a direct 36^5-entry (start,count) table in
//     HBM (484 MB of 288 GB) -> one O(1) lookup per target k-mer instead of the reference's 3 x 242 MB arrays.
//   * one workgroup per target: k-mers and per-position row sizes go to LDS, hits are expanded in
//     parallel into 30-bit keys (q << 14 | diag) in LDS, sorted there (bitonic), adjacent equal keys =
//     two-hit diagonals, one thread per diagonal runs the Kadane scan (target letters from LDS), runs of
//     equal q are max-reduced and (q, t, score) is appended to the output with one atomic per run.
//   * targets with more hits than fit in LDS are processed in query-range chunks.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "rsk_internal.h"
#include "rsk_tables_data.h"

#define PF_THREADS 512
#define PF_CAP 4096               // keys per chunk (hash set of 2 * PF_CAP slots in LDS)
#define PF_QSPAN 4096             // queries per chunk (64 buckets of 64)
#define PF_DICT 60466176u         // 36^5
#define PF_LONGROW 64             // index rows from this length on are walked by a whole wave
#define PF_MINSELF 36             // MIN_KMER_PAIR_SCORE prefiltermuparams.h:22

static __device__ __constant__ signed char c_mu_s8[36 * 36];   // Mu_S_ij_i8 (mumx_data.cpp:81)

static int pf_upload_tables(rsk_ctx *ctx)
{
    static std::atomic<int> done[64];
    return rsk_once_per_device(done, ctx->device, [&]() -> int {
        RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_mu_s8), rsk_mu_s8, sizeof(rsk_mu_s8)));
        return RSK_OK;
    });
}

// ---------------------------------------------------------------------------------------------
// P10 index build on the device.  mode 0: exact k-mers (MuDex::FromSeqDB without neighbourhood);
// mode 1 ("idxq", mudex.cpp:158-176,201-219): the posting (q, pos) of k-mer K goes to row K AND to the
// row of every 5-mer K' with pair score S(K,K') >= 36 -- K itself included, so an exact match is listed
// twice (and is a "two-hit" diagonal on its own: reference behaviour, kept); mode 2 ("idxt",
// prefiltermu.cpp:174-199): the reference enumerates the neighbourhood of each TARGET k-mer against the
// plain index; S is symmetric, so listing (q, pos) in the rows of all neighbours of K (once each, K
// included) gives the same (TPos, QIdx, QPos) items -- with 288 GB of HBM the expanded index
// (~660 postings per query position) replaces the per-target enumeration.
// One workgroup per chain, one thread per k-mer position: branch-and-bound over the per-letter score
// lists sorted by decreasing score (the reference's MerMx::GetHighScoring5mers mermx.cpp:484 bounds
// AB|CD|E the same way; only the resulting SET matters).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf_hood(const uint8_t *mu, const uint32_t *off, const uint32_t *len, int mode, int pass,
                                                 uint32_t *cnt, const uint2 *table, uint32_t *postings,
                                                 unsigned long long *total)
{
    __shared__ signed char ss[36][36];     // scores of letter l's partners, descending
    __shared__ uint8_t sl[36][36];         // the partner letters in that order
    const int tid = threadIdx.x;
    if (tid < 36) {
        signed char sc[36];
        uint8_t lt[36];
        for (int b = 0; b < 36; ++b) { sc[b] = c_mu_s8[tid * 36 + b]; lt[b] = (uint8_t) b; }
        for (int i = 1; i < 36; ++i) {                     // insertion sort, descending, stable
            const signed char v = sc[i]; const uint8_t l = lt[i];
            int j = i - 1;
            while (j >= 0 && sc[j] < v) { sc[j + 1] = sc[j]; lt[j + 1] = lt[j]; --j; }
            sc[j + 1] = v; lt[j + 1] = l;
        }
        for (int b = 0; b < 36; ++b) { ss[tid][b] = sc[b]; sl[tid][b] = lt[b]; }
    }
    __syncthreads();
    const uint32_t q = blockIdx.x;
    const uint32_t L = len[q];
    const uint8_t *s = mu + off[q];
    unsigned long long mine = 0;
    auto emit = [&](uint32_t code, uint32_t p) {
        if (pass == 0) { atomicAdd(&cnt[code], 1u); ++mine; }
        else { const uint32_t slot = atomicAdd(&cnt[code], 1u); postings[table[code].x + slot] = (q << 16) | p; }
    };
    for (uint32_t p = tid; p + 7 <= L; p += blockDim.x) {
        const uint32_t a0 = s[p], a1 = s[p + 1], a2 = s[p + 2], a3 = s[p + 5], a4 = s[p + 6];
        const int self = c_mu_s8[a0 * 37] + c_mu_s8[a1 * 37] + c_mu_s8[a2 * 37] + c_mu_s8[a3 * 37] + c_mu_s8[a4 * 37];
        if (self < PF_MINSELF) continue;
        if (mode != 2) emit((((a0 * 36 + a1) * 36 + a2) * 36 + a3) * 36 + a4, p);
        if (mode == 0) continue;
        const int r4 = ss[a4][0], r3 = r4 + ss[a3][0], r2 = r3 + ss[a2][0], r1 = r2 + ss[a1][0];
        for (int i0 = 0; i0 < 36; ++i0) {
            const int s0 = ss[a0][i0];
            if (s0 + r1 < PF_MINSELF) break;
            const uint32_t c0 = sl[a0][i0];
            for (int i1 = 0; i1 < 36; ++i1) {
                const int s1 = s0 + ss[a1][i1];
                if (s1 + r2 < PF_MINSELF) break;
                const uint32_t c1 = c0 * 36 + sl[a1][i1];
                for (int i2 = 0; i2 < 36; ++i2) {
                    const int s2 = s1 + ss[a2][i2];
                    if (s2 + r3 < PF_MINSELF) break;
                    const uint32_t c2 = c1 * 36 + sl[a2][i2];
                    for (int i3 = 0; i3 < 36; ++i3) {
                        const int s3 = s2 + ss[a3][i3];
                        if (s3 + r4 < PF_MINSELF) break;
                        const uint32_t c3 = c2 * 36 + sl[a3][i3];
                        for (int i4 = 0; i4 < 36; ++i4) {
                            if (s3 + ss[a4][i4] < PF_MINSELF) break;
                            emit(c3 * 36 + sl[a4][i4], p);
                        }
                    }
                }
            }
        }
    }
    if (pass == 0 && mine) atomicAdd(total, mine);
}

__global__ void k_pf_make_table(const uint32_t *start, uint32_t *cnt, uint2 *table)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= PF_DICT) return;
    table[k] = make_uint2(start[k], cnt[k]);
    cnt[k] = 0;                    // becomes the fill cursor
}

struct pf_args {
    const uint2 *table;            // [36^5] (start, count) into postings
    const uint32_t *postings;      // q << 16 | pos, grouped by k-mer row (unordered within a row)
    const uint8_t *q_mu; const uint32_t *q_off; const uint32_t *q_len;
    const uint8_t *t_mu; const uint32_t *t_off; const uint32_t *t_len;
    uint32_t nt, nq;
    uint32_t *out_q, *out_t, *out_score;
    uint32_t capacity;
    uint32_t *out_n;
    uint32_t *overflow;            // set if a single 64-query bucket exceeds PF_CAP hits for one target
    unsigned long long *hits;      // statistics: (TPos, posting) items seen over all targets
    uint32_t t_base;               // first target of this launch (targets are batched by scratch size)
    const uint64_t *koff;          // per target of the launch: offset of its key region in kscratch
    uint32_t tl_cap;               // target letters staged in LDS up to this length
    unsigned long long *stat;      // RSK_TRACE: chunks, overflowing buckets, query runs, dense queries, two-hit diagonals, clock cycles of the chunk phase
    uint32_t dbg;                  // RSK_PF_DEBUG: 1 = stop after the counting pass, 2 = after the scatter pass (timing experiments)
    uint32_t *kscratch;            // keys (q << 14 | diag) of a target, grouped by 64-query bucket (written once, read once)
};

__device__ __forceinline__ uint32_t pf_kmer(const uint8_t *s, int &self)
{
    const int o[5] = { 0, 1, 2, 5, 6 };
    uint32_t k = 0;
    self = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint32_t l = s[o[i]];
        k = k * 36 + l;
        self += c_mu_s8[l * 36 + l];
    }
    return k;
}

// upper bound of the keys of each target (sum of the index row sizes of its unmasked k-mers) -> sizes the scratch
__global__ __launch_bounds__(256) void k_pf_rowsum(const uint2 *table, const uint8_t *t_mu, const uint32_t *t_off, const uint32_t *t_len,
                                                   uint32_t nt, uint64_t *sums)
{
    const uint32_t t = blockIdx.x;
    if (t >= nt) return;
    __shared__ unsigned long long acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const uint32_t TL = t_len[t];
    const uint8_t *T = t_mu + t_off[t];
    unsigned long long mine = 0;
    for (uint32_t p = threadIdx.x; p + 7 <= TL; p += blockDim.x) {
        int self;
        const uint32_t k = pf_kmer(T + p, self);
        if (self < PF_MINSELF) continue;
        mine += table[k].y;
    }
    if (mine) atomicAdd(&acc, mine);
    __syncthreads();
    if (threadIdx.x == 0) sums[t] = acc;
}

__global__ __launch_bounds__(PF_THREADS) void k_prefilter(pf_args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *keys = (uint32_t *) smem;                                  // 2 * PF_CAP: hash set of a chunk's keys / dense histogram
    uint32_t *list2 = keys + 2 * PF_CAP;                                 // PF_CAP / 2: the chunk's two-hit (query, diagonal) keys
    uint32_t *qmax = list2 + PF_CAP / 2;                                 // PF_QSPAN: best diagonal score per query of the chunk
    uint32_t *bucket = qmax + PF_QSPAN;                                  // 1025: hits per 64-query bucket
    uint32_t *boff = bucket + 1032;                                      // 1025: start of each bucket in this target's key region
    uint32_t *cursor = boff + 1032;                                      // 1024: scatter cursors
    uint32_t *sv = cursor + 1024;                                        // 8 scalars shared by the workgroup
    signed char *mat = (signed char *) (sv + 8);                         // 1296
    uint8_t *tl_lds = (uint8_t *) (mat + 1312);                          // target letters (when they fit: a.tl_cap)
    uint32_t &s_n = sv[0], &s_total = sv[1], &s_chunk_lo = sv[2], &s_chunk_hi = sv[3], &s_more = sv[4];

    const int tid = threadIdx.x;
    for (int i = tid; i < 1296; i += PF_THREADS) mat[i] = c_mu_s8[i];
    const uint32_t t = a.t_base + blockIdx.x;
    uint32_t *kscr = a.kscratch + a.koff[blockIdx.x];
    const uint32_t TL = a.t_len[t];
    const uint8_t *T = a.t_mu + a.t_off[t];
    const uint8_t *tl = T;                                               // chains longer than the LDS staging are read in place
    if (TL <= a.tl_cap) {
        for (uint32_t i = tid; i < TL; i += PF_THREADS) tl_lds[i] = T[i];
        tl = tl_lds;
    }
    __syncthreads();
    if (TL < 7) return;
    const uint32_t NK = TL - 6;

    // ---- walk of the index rows of this target's k-mers, used twice (count, then scatter).  Row lengths span four
    // orders of magnitude with neighbourhood indexes; a thread that walked a 10^4-posting row alone kept its wave
    // busy for milliseconds of dependent latency.  Short rows stay with the thread that owns the position, rows of
    // >= PF_LONGROW postings are queued in LDS and split over the lanes of a wave (coalesced posting loads).
    uint32_t *rowq = keys;                                               // (position, start, count) of the queued rows; keys[] is free here
    const int lane = tid & 63, wid = tid >> 6;
    auto walk = [&](auto &&item) {
        for (uint32_t base = 0; base < NK; base += PF_THREADS) {
            const uint32_t p = base + tid;
            if (tid == 0) sv[7] = 0;
            __syncthreads();
            if (p < NK) {
                int self;
                const uint32_t k = pf_kmer(tl + p, self);
                if (self >= PF_MINSELF) {
                    const uint2 r = a.table[k];
                    if (r.y >= PF_LONGROW) {
                        const uint32_t e = atomicAdd(&sv[7], 1u);
                        rowq[3 * e] = p; rowq[3 * e + 1] = r.x; rowq[3 * e + 2] = r.y;
                    } else
                        for (uint32_t c = 0; c < r.y; ++c) item(p, a.postings[r.x + c]);
                }
            }
            __syncthreads();
            const uint32_t nrow = sv[7];
            for (uint32_t e = wid; e < nrow; e += PF_THREADS / 64) {
                const uint32_t rp = rowq[3 * e], rx = rowq[3 * e + 1], ry = rowq[3 * e + 2];
                for (uint32_t c = lane; c < ry; c += 64) item(rp, a.postings[rx + c]);
            }
            __syncthreads();
        }
    };
    // ---- hits per 64-query bucket over the whole target -> chunks of <= PF_CAP keys
    for (int i = tid; i < 1025; i += PF_THREADS) bucket[i] = 0;
    __syncthreads();
    uint32_t my_total = 0;
    walk([&](uint32_t p, uint32_t post) {
        const uint32_t q = post >> 16, qp = post & 0xFFFFu;
        const uint32_t d = (a.q_len[q] + p - qp - 1) & 0xFFFFu;
        if (d > 16383u) return;
        atomicAdd(&bucket[q >> 6], 1u);
        ++my_total;
    });
    if (tid == 0) s_total = 0;
    __syncthreads();
    if (my_total) atomicAdd(&s_total, my_total);
    __syncthreads();
    if (tid == 0 && a.hits) atomicAdd(a.hits, (unsigned long long) s_total);
    if (s_total < 2) return;
    if (a.dbg == 1) return;
    // ---- every key goes ONCE to this target's region of the HBM scratch, grouped by bucket (counting sort);
    // the chunks below are then contiguous ranges of it (re-walking the index rows per chunk was quadratic
    // in the hits of a target -- neighbourhood indexes have ~100x the hits of exact k-mers)
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 1024; ++b) { boff[b] = run; run += bucket[b]; }
        boff[1024] = run;
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += PF_THREADS) cursor[i] = boff[i];
    __syncthreads();
    walk([&](uint32_t p, uint32_t post) {
        const uint32_t q = post >> 16, qp = post & 0xFFFFu;
        const uint32_t d = (a.q_len[q] + p - qp - 1) & 0xFFFFu;
        if (d > 16383u) return;
        const uint32_t pos = atomicAdd(&cursor[q >> 6], 1u);
        kscr[pos] = (q << 14) | d;        // read back by this workgroup only, after __threadfence + barrier; never read before
    });
    __threadfence();
    __syncthreads();
    if (a.dbg == 2) return;

    // Kadane over the whole diagonal d of query q (FindHSP prefiltermu.cpp:12, diag.h:51-92), score clamped to u16
    auto diag_score = [&](uint32_t q, int d) -> uint32_t {
        const int QL = (int) a.q_len[q];
        const uint8_t *Q = a.q_mu + a.q_off[q];
        int i0 = QL - d - 1; if (i0 < 0) i0 = 0;
        int j0 = d + 1 - QL; if (j0 < 0) j0 = 0;
        int hi = QL - 1; if (QL + (int) TL - d - 2 < hi) hi = QL + (int) TL - d - 2;
        int len = hi - i0 + 1;
        int F = 0, Bst = 0;
        auto step = [&](uint32_t ql, uint32_t tlet) {
            F += mat[ql * 36 + tlet];
            if (F > Bst) Bst = F;
            else if (F < 0) F = 0;
        };
        const uint8_t *qp = Q + i0, *tp = tl + j0;
        while (len > 0 && ((uintptr_t) qp & 3)) { step(*qp++, *tp++); --len; }
        if (len >= 4) {
            // four residues per iteration: one aligned dword of query letters, the target letters through a sliding pair
            // of aligned dwords (chains and the LDS staging are padded, reading up to 3 bytes past the end is safe)
            const uint32_t tsh = (uint32_t) ((uintptr_t) tp & 3);
            const uint32_t *tw = (const uint32_t *) (tp - tsh);
            uint32_t ta = *tw++;
            for (; len >= 4; len -= 4, qp += 4, tp += 4) {
                const uint32_t qw = *(const uint32_t *) qp;
                const uint32_t tb = *tw++;
                const uint32_t t4 = __builtin_amdgcn_alignbyte(tb, ta, tsh);
                ta = tb;
                step(qw & 0xFFu, t4 & 0xFFu);
                step((qw >> 8) & 0xFFu, (t4 >> 8) & 0xFFu);
                step((qw >> 16) & 0xFFu, (t4 >> 16) & 0xFFu);
                step(qw >> 24, t4 >> 24);
            }
        }
        while (len > 0) { step(*qp++, *tp++); --len; }
        return (uint32_t) (Bst > 0 ? (Bst >= 65535 ? 65534 : Bst) : 0);
    };
    auto emit = [&](uint32_t q, uint32_t best) {
        const uint32_t pos = atomicAdd(a.out_n, 1u);
        if (pos < a.capacity) { a.out_q[pos] = q; a.out_t[pos] = t; a.out_score[pos] = best; }
    };
    // The keys [k0, k1) of the scratch whose query lies in [qa, qb) (at most PF_CAP of them, qb - qa <= PF_QSPAN):
    // two-hit diagonals = keys that occur at least twice (twohitdiag.cpp:368-389).  Every key goes into an LDS hash set
    // (linear probing, load <= 0.5); the FIRST repeat of a key marks its slot and appends the key to list2; then one
    // thread per two-hit diagonal (dense lanes) scores it, and each query emits its best.  (A bitonic sort of the chunk
    // did this before: 91 barrier-separated passes over 8192 keys.)  Called by the whole workgroup.
    auto process_range = [&](uint32_t k0, uint32_t k1, uint32_t qa, uint32_t qb, bool filter) {
        const uint32_t nqc = qb - qa;
        for (uint32_t i = tid; i < 2 * PF_CAP; i += PF_THREADS) keys[i] = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < nqc; i += PF_THREADS) qmax[i] = 0;
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (uint32_t idx = k0 + tid; idx < k1; idx += PF_THREADS) {
            const uint32_t key = kscr[idx];
            if (filter && ((key >> 14) < qa || (key >> 14) >= qb)) continue;
            uint32_t h = (key * 2654435761u) >> 19;                              // 13 bits: 2 * PF_CAP slots
            for (;;) {
                const uint32_t old = atomicCAS(&keys[h], 0xFFFFFFFFu, key);
                if (old == 0xFFFFFFFFu) break;                                   // first occurrence
                if ((old & 0x7FFFFFFFu) == key) {
                    if (!(old & 0x80000000u) && !(atomicOr(&keys[h], 0x80000000u) & 0x80000000u)) list2[atomicAdd(&s_n, 1u)] = key;
                    break;
                }
                h = (h + 1) & (2 * PF_CAP - 1);
            }
        }
        __syncthreads();
        const uint32_t n2 = s_n;
        if (a.stat && tid == 0) atomicAdd(a.stat + 4, (unsigned long long) n2);
        for (uint32_t i = tid; i < n2; i += PF_THREADS) {
            const uint32_t key = list2[i];
            const uint32_t sc = diag_score(key >> 14, (int) (key & 16383u));
            if (sc > 0) atomicMax(&qmax[(key >> 14) - qa], sc);
        }
        __syncthreads();
        for (uint32_t i = tid; i < nqc; i += PF_THREADS)
            if (qmax[i] > 0) emit(qa + i, qmax[i]);
        __syncthreads();
    };
    // One query whose keys alone exceed PF_CAP for this target (low-complexity chains): two 16384-bit LDS bitmaps
    // (diagonal seen once / seen twice) instead of the hash set.
    auto process_dense_query = [&](uint32_t k0, uint32_t k1, uint32_t q) {
        uint32_t *seen1 = keys, *seen2 = keys + 512;
        for (uint32_t i = tid; i < 1024; i += PF_THREADS) seen1[i] = 0;
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (uint32_t idx = k0 + tid; idx < k1; idx += PF_THREADS) {
            const uint32_t key = kscr[idx];
            if ((key >> 14) != q) continue;
            const uint32_t d = key & 16383u, bit = 1u << (d & 31);
            if (atomicOr(&seen1[d >> 5], bit) & bit) atomicOr(&seen2[d >> 5], bit);
        }
        __syncthreads();
        uint32_t best = 0;
        for (uint32_t d = tid; d < 16384u; d += PF_THREADS)
            if ((seen2[d >> 5] >> (d & 31)) & 1u) best = max(best, diag_score(q, (int) d));
        if (best) atomicMax(&s_n, best);
        __syncthreads();
        if (tid == 0 && s_n > 0) emit(q, s_n);
        __syncthreads();
    };

    const long long t_chunk0 = a.stat ? (long long) clock64() : 0;
    uint32_t chunk_lo = 0;                    // first bucket of the current chunk
    for (;;) {
        // thread 0 picks the chunk [chunk_lo, chunk_hi) of buckets with <= PF_CAP keys
        if (tid == 0) {
            uint32_t sum = 0, hi = chunk_lo;
            while (hi < 1024 && hi - chunk_lo < PF_QSPAN / 64 && sum + bucket[hi] <= PF_CAP) { sum += bucket[hi]; ++hi; }
            sv[5] = 0;
            if (hi == chunk_lo) { sv[5] = 1; hi = chunk_lo + 1; sum = bucket[chunk_lo]; }   // one bucket alone exceeds the hash set
            s_chunk_lo = chunk_lo; s_chunk_hi = hi; s_more = hi < 1024 ? 1u : 0u;
            s_total = sum;
        }
        __syncthreads();
        const uint32_t qlo = s_chunk_lo << 6, qhi = min(s_chunk_hi << 6, a.nq);
        const uint32_t expect = s_total, over = sv[5];
        const uint32_t k0 = boff[s_chunk_lo], k1 = boff[s_chunk_hi];
        __syncthreads();
        if (a.stat && tid == 0) { atomicAdd(a.stat + 0, 1ull); if (over) atomicAdd(a.stat + 1, 1ull); }
        if (!over) {
            if (expect >= 2) process_range(k0, k1, qlo, qhi, false);
        } else {
            // ---- the bucket is cut into runs of queries with <= PF_CAP keys each (its keys are scanned once per run);
            // a query that exceeds PF_CAP on its own takes the bitmap path
            uint32_t *qcnt = qmax;                                               // 64 counters (qmax is cleared by process_range)
            for (uint32_t i = tid; i < 64; i += PF_THREADS) qcnt[i] = 0;
            __syncthreads();
            for (uint32_t idx = k0 + tid; idx < k1; idx += PF_THREADS) atomicAdd(&qcnt[(kscr[idx] >> 14) - qlo], 1u);
            __syncthreads();
            uint32_t *run_lo = cursor, *run_hi = cursor + 64, *run_dense = cursor + 128;   // the scatter cursors are free now
            if (tid == 0) {
                uint32_t nr = 0, q = 0;
                const uint32_t nqb = qhi - qlo;
                while (q < nqb) {
                    if (qcnt[q] > PF_CAP) { run_lo[nr] = q; run_hi[nr] = q + 1; run_dense[nr] = 1; ++nr; ++q; continue; }
                    uint32_t sum = 0, e = q;
                    while (e < nqb && qcnt[e] <= PF_CAP && sum + qcnt[e] <= PF_CAP) { sum += qcnt[e]; ++e; }
                    run_lo[nr] = q; run_hi[nr] = e; run_dense[nr] = sum < 2 ? 2 : 0; ++nr;       // 2 = nothing to do
                    q = e;
                }
                sv[6] = nr;
            }
            __syncthreads();
            const uint32_t nr = sv[6];
            for (uint32_t r = 0; r < nr; ++r) {
                const uint32_t qa = qlo + run_lo[r], qb = qlo + run_hi[r], kind = run_dense[r];
                __syncthreads();
                if (a.stat && tid == 0) atomicAdd(a.stat + (kind == 1 ? 3 : 2), 1ull);
                if (kind == 1) process_dense_query(k0, k1, qa);
                else if (kind == 0) process_range(k0, k1, qa, qb, true);
            }
        }
        __syncthreads();
        if (!s_more) break;
        chunk_lo = s_chunk_hi;
        __syncthreads();
    }
    if (a.stat && tid == 0) atomicAdd(a.stat + 5, (unsigned long long) ((long long) clock64() - t_chunk0));
}

// ---------------------------------------------------------------------------------------------
// host: index build (P10) + launch
// ---------------------------------------------------------------------------------------------
int rsk_build_mudex(rsk_ctx *ctx, rsk_db *db, int mode)
{
    if (db->mudex_built && db->mudex_mode == mode) return RSK_OK;
    if (db->n > 65535) { rsk_set_error("k-mer prefilter: at most 65535 query chains (uint16 query index, prefiltermu.cpp:296)"); return RSK_E_RANGE; }
    for (uint32_t L : db->len)
        if (L > 65535) { rsk_set_error("k-mer prefilter: query longer than 65535 (uint16 position)"); return RSK_E_RANGE; }
    if (db->d_pf_postings) { (void) hipFree(db->d_pf_postings); db->d_pf_postings = nullptr; db->hbm_bytes -= db->pf_postings * 4; }
    // temporaries through the CALLING context's pool (returned on every exit path); everything on that context's stream
    // (the context that created the set may be another thread's, e.g. the -db loader's)
    rsk_scratch ws(ctx);
    uint32_t *d_cnt, *d_start;
    unsigned long long *d_total, total = 0;
    void *d_tmp;
    int rc;
    if (!db->d_pf_table) {
        { const int rc_ = rsk_dev_malloc(nullptr, (void **) &db->d_pf_table, (size_t) PF_DICT * sizeof(uint2)); if (rc_ != RSK_OK) return rc_; }
        db->hbm_bytes += (size_t) PF_DICT * sizeof(uint2);
    }
    if ((rc = ws.alloc(&d_cnt, (size_t) PF_DICT)) || (rc = ws.alloc(&d_start, (size_t) PF_DICT)) || (rc = ws.alloc(&d_total, 1))) return rc;
    RSK_HIP(hipMemsetAsync(d_cnt, 0, (size_t) PF_DICT * 4, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_total, 0, 8, ctx->stream));
    if (db->n) hipLaunchKernelGGL(k_pf_hood, dim3(db->n), dim3(256), 0, ctx->stream, db->d_mu, db->d_off, db->d_len, mode, 0, d_cnt,
                                  (const uint2 *) nullptr, (uint32_t *) nullptr, d_total);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    if (total > 0xFFFFFFF0ull) { rsk_set_error("k-mer prefilter: %llu index postings exceed 2^32; split the query set", total); return RSK_E_RANGE; }
    size_t tmp_bytes = 0;
    RSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_start, (int) PF_DICT, ctx->stream));
    if ((rc = ws.alloc(&d_tmp, std::max<size_t>(tmp_bytes, 16))) != RSK_OK) return rc;
    RSK_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cnt, d_start, (int) PF_DICT, ctx->stream));
    hipLaunchKernelGGL(k_pf_make_table, dim3((PF_DICT + 255) / 256), dim3(256), 0, ctx->stream, d_start, d_cnt, (uint2 *) db->d_pf_table);
    { const int rc_ = rsk_dev_malloc(nullptr, (void **) &db->d_pf_postings, std::max<size_t>((size_t) total, 1) * 4); if (rc_ != RSK_OK) return rc_; }
    if (db->n) hipLaunchKernelGGL(k_pf_hood, dim3(db->n), dim3(256), 0, ctx->stream, db->d_mu, db->d_off, db->d_len, mode, 1, d_cnt,
                                  (const uint2 *) db->d_pf_table, db->d_pf_postings, d_total);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    db->pf_postings = (size_t) total;
    db->hbm_bytes += (size_t) total * 4;
    db->mudex_built = true;
    db->mudex_mode = mode;
    return RSK_OK;
}

extern "C" int rsk_mu_prefilter_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int neighbourhood, uint32_t *d_out_q,
                                    uint32_t *d_out_t, uint32_t *d_out_score, size_t capacity, uint32_t *d_n)
{
    if (!ctx || !q || !t || !d_out_q || !d_out_t || !d_out_score || !d_n) { rsk_set_error("rsk_mu_prefilter_dev: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_prefilter_dev: chain set has no Mu letters"); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = pf_upload_tables(ctx);
    if (rc != RSK_OK) return rc;
    if (neighbourhood < -1 || neighbourhood > 2) { rsk_set_error("rsk_mu_prefilter_dev: neighbourhood must be -1, 0, 1 or 2"); return RSK_E_INVALID; }
    if (neighbourhood == -1) neighbourhood = q->n <= 100 ? 1 : 2;      // MAX_QUERY_CHAINS_FOR_QUERY_NEIGHBORHOOD muprefilter.cpp:78-87
    if ((rc = rsk_build_mudex(ctx, const_cast<rsk_db *>(q), neighbourhood)) != RSK_OK) return rc;
    for (uint32_t L : t->len)
        if (L > 65534) { rsk_set_error("rsk_mu_prefilter_dev: target longer than 65534"); return RSK_E_RANGE; }
    rsk_scratch ws(ctx);                       // every temporary goes back to the pool on every exit path
    uint32_t *d_over;
    if ((rc = ws.alloc((void **) &d_over, 16 + 64)) != RSK_OK) return rc;
    RSK_HIP(hipMemsetAsync(d_over, 0, 16 + 64, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_n, 0, 4, ctx->stream));
    pf_args a = {};
    a.table = (const uint2 *) q->d_pf_table; a.postings = q->d_pf_postings;
    a.q_mu = q->d_mu; a.q_off = q->d_off; a.q_len = q->d_len;
    a.t_mu = t->d_mu; a.t_off = t->d_off; a.t_len = t->d_len; a.nt = t->n; a.nq = q->n;
    a.out_q = d_out_q; a.out_t = d_out_t; a.out_score = d_out_score;
    a.capacity = (uint32_t) std::min<size_t>(capacity, 0xFFFFFFFFu);
    a.out_n = d_n; a.overflow = d_over; a.hits = (unsigned long long *) (d_over + 2);
    a.stat = getenv("RSK_TRACE") ? (unsigned long long *) (d_over + 4) : nullptr;
    uint32_t maxTL = 0;
    for (uint32_t L : t->len) maxTL = std::max(maxTL, L);
    // hash set + two-hit list + per-query maxima + bucket bookkeeping + matrix + target letters (as many as fit; longer
    // targets are read from HBM in place)
    const size_t lds_fixed = (size_t) PF_CAP * 8 + (size_t) PF_CAP / 2 * 4 + (size_t) PF_QSPAN * 4 + (1032 + 1032 + 1024 + 8) * 4 + 1312;
    // two workgroups per CU (their chunk loops are latency-bound): each may use half of the 160 KB
    const uint32_t tl_cap = (uint32_t) std::min<size_t>(maxTL, (81000 - lds_fixed - 32) & ~(size_t) 15);
    a.tl_cap = tl_cap;
    a.dbg = getenv("RSK_PF_DEBUG") ? (uint32_t) atoi(getenv("RSK_PF_DEBUG")) : 0;
    const size_t lds = lds_fixed + (((size_t) tl_cap + 31) & ~(size_t) 15);
    RSK_HIP(hipFuncSetAttribute((const void *) k_prefilter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));   // depends on the call's targets
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (t->n) {
        // key scratch: per target an upper bound of its keys; targets go in batches whose scratch fits the budget
        uint64_t *d_sums;
        if ((rc = ws.alloc(&d_sums, (size_t) t->n)) != RSK_OK) return rc;
        hipLaunchKernelGGL(k_pf_rowsum, dim3(t->n), dim3(256), 0, ctx->stream, (const uint2 *) q->d_pf_table, t->d_mu, t->d_off, t->d_len, t->n, d_sums);
        std::vector<uint64_t> sums(t->n);
        RSK_HIP(hipMemcpyAsync(sums.data(), d_sums, (size_t) t->n * 8, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipStreamSynchronize(ctx->stream));
        const uint64_t budget_keys = 12ull << 30;                       // 48 GB of 4-byte keys per batch
        // batches: the largest key count and target count size the two scratch blocks once
        uint64_t max_keys = 0;
        uint32_t max_nt = 0;
        for (uint32_t t0 = 0; t0 < t->n;) {
            uint32_t t1 = t0;
            uint64_t keys = 0;
            while (t1 < t->n && (t1 == t0 || keys + sums[t1] <= budget_keys)) { keys += sums[t1]; ++t1; }
            max_keys = std::max(max_keys, keys); max_nt = std::max(max_nt, t1 - t0);
            t0 = t1;
        }
        uint32_t *d_scr;
        uint64_t *d_koff;
        if ((rc = ws.alloc(&d_scr, (size_t) max_keys + 16)) != RSK_OK) { rsk_set_error("rsk_mu_prefilter_dev: out of device memory for %llu seed keys", (unsigned long long) max_keys); return rc; }
        if ((rc = ws.alloc(&d_koff, (size_t) max_nt)) != RSK_OK) return rc;
        std::vector<uint64_t> koff;
        for (uint32_t t0 = 0; t0 < t->n;) {
            uint32_t t1 = t0;
            uint64_t keys = 0;
            koff.clear();
            while (t1 < t->n && (t1 == t0 || keys + sums[t1] <= budget_keys)) { koff.push_back(keys); keys += sums[t1]; ++t1; }
            RSK_HIP(hipMemcpyAsync(d_koff, koff.data(), koff.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            a.t_base = t0; a.koff = d_koff; a.kscratch = d_scr;
            hipLaunchKernelGGL(k_prefilter, dim3(t1 - t0), dim3(PF_THREADS), lds, ctx->stream, a);
            RSK_HIP(hipGetLastError());
            RSK_HIP(hipStreamSynchronize(ctx->stream));                 // koff (host vector) and the scratch are reused by the next batch
            t0 = t1;
        }
    }
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    uint32_t over = 0;
    unsigned long long hits = 0;
    RSK_HIP(hipMemcpyAsync(&over, d_over, 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(&hits, d_over + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long stat[8] = { 0 };
    RSK_HIP(hipMemcpyAsync(stat, d_over + 4, 64, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->pf_hits = hits;
    ctx->pf_postings = q->pf_postings;
    if (getenv("RSK_TRACE"))
        fprintf(stderr, "[prefilter] index postings %zu, seed items %llu; chunks %llu, overflowing buckets %llu (query runs %llu, dense queries %llu), "
                        "two-hit diagonals %llu, chunk-phase cycles (100 MHz clock, summed over targets) %llu\n",
                q->pf_postings, hits, stat[0], stat[1], stat[2], stat[3], stat[4], stat[5]);
    if (over) { rsk_set_error("rsk_mu_prefilter_dev: more than %d k-mer hits between one target and 64 consecutive queries", PF_CAP); return RSK_E_RANGE; }
    return RSK_OK;
}

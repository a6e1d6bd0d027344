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
#include <vector>

#include "rsk_internal.h"
#include "rsk_tables_data.h"

#define PF_THREADS 512
#define PF_CAP 8192               // keys per chunk held in LDS
#define PF_DICT 60466176u         // 36^5
#define PF_MINSELF 36             // MIN_KMER_PAIR_SCORE prefiltermuparams.h:22

static __device__ __constant__ signed char c_mu_s8[36 * 36];   // Mu_S_ij_i8 (mumx_data.cpp:81)

static int pf_upload_tables(rsk_ctx *ctx)
{
    static bool done[64] = { false };
    if (ctx->device < 64 && done[ctx->device]) return RSK_OK;
    RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_mu_s8), rsk_mu_s8, sizeof(rsk_mu_s8)));
    if (ctx->device < 64) done[ctx->device] = true;
    return RSK_OK;
}

__global__ void k_pf_fill_table(const uint32_t *ukmer, const uint32_t *ustart, uint32_t nu, uint2 *table)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nu) table[ukmer[i]] = make_uint2(ustart[i], ustart[i + 1] - ustart[i]);
}

struct pf_args {
    const uint2 *table;            // [36^5] (start, count) into postings
    const uint32_t *postings;      // q << 16 | pos, sorted by (kmer, q, pos)
    const uint8_t *q_mu; const uint32_t *q_off; const uint32_t *q_len;
    const uint8_t *t_mu; const uint32_t *t_off; const uint32_t *t_len;
    uint32_t nt, nq;
    uint32_t *out_q, *out_t, *out_score;
    uint32_t capacity;
    uint32_t *out_n;
    uint32_t *overflow;            // set if a single 64-query bucket exceeds PF_CAP hits for one target
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

__global__ __launch_bounds__(PF_THREADS) void k_prefilter(pf_args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *keys = (uint32_t *) smem;                                  // PF_CAP
    uint32_t *score = keys + PF_CAP;                                     // PF_CAP (score of the run head candidates)
    uint32_t *bucket = score + PF_CAP;                                   // 1025: hits per 64-query bucket
    uint32_t *sv = bucket + 1032;                                        // 8 scalars shared by the workgroup
    signed char *mat = (signed char *) (sv + 8);                         // 1296
    uint8_t *tl = (uint8_t *) (mat + 1312);                              // target letters, up to 65536 + 16
    uint32_t &s_n = sv[0], &s_total = sv[1], &s_chunk_lo = sv[2], &s_chunk_hi = sv[3], &s_more = sv[4];

    const int tid = threadIdx.x;
    for (int i = tid; i < 1296; i += PF_THREADS) mat[i] = c_mu_s8[i];
    const uint32_t t = blockIdx.x;
    const uint32_t TL = a.t_len[t];
    const uint8_t *T = a.t_mu + a.t_off[t];
    for (uint32_t i = tid; i < TL; i += PF_THREADS) tl[i] = T[i];
    __syncthreads();
    if (TL < 7) return;
    const uint32_t NK = TL - 6;

    // ---- hits per 64-query bucket over the whole target -> chunks of <= PF_CAP keys
    for (int i = tid; i < 1025; i += PF_THREADS) bucket[i] = 0;
    __syncthreads();
    uint32_t my_total = 0;
    for (uint32_t p = tid; p < NK; p += PF_THREADS) {
        int self;
        const uint32_t k = pf_kmer(tl + p, self);
        if (self < PF_MINSELF) continue;
        const uint2 r = a.table[k];
        for (uint32_t c = 0; c < r.y; ++c) {
            const uint32_t post = a.postings[r.x + c];
            const uint32_t q = post >> 16, qp = post & 0xFFFFu;
            const uint32_t d = (a.q_len[q] + p - qp - 1) & 0xFFFFu;
            if (d > 16383u) continue;
            atomicAdd(&bucket[q >> 6], 1u);
            ++my_total;
        }
    }
    if (tid == 0) s_total = 0;
    __syncthreads();
    if (my_total) atomicAdd(&s_total, my_total);
    __syncthreads();
    if (s_total < 2) return;

    uint32_t chunk_lo = 0;                    // first bucket of the current chunk
    for (;;) {
        // thread 0 picks the chunk [chunk_lo, chunk_hi) of buckets with <= PF_CAP hits
        if (tid == 0) {
            uint32_t sum = 0, hi = chunk_lo;
            while (hi < 1024 && sum + bucket[hi] <= PF_CAP) { sum += bucket[hi]; ++hi; }
            sv[5] = 0;
            if (hi == chunk_lo) { sv[5] = 1; hi = chunk_lo + 1; sum = 0; }    // one bucket alone overflows the key array: dense path
            s_chunk_lo = chunk_lo; s_chunk_hi = hi; s_more = hi < 1024 ? 1u : 0u; s_n = 0;
            s_total = sum;
        }
        __syncthreads();
        const uint32_t qlo = s_chunk_lo << 6, qhi = s_chunk_hi << 6;
        const uint32_t expect = s_total;
        if (sv[5]) {
            // ---- dense path (low-complexity chains): one query at a time, hits counted per diagonal in a
            // 16384-entry LDS histogram (keys[] and score[] together), no sort.
            uint32_t *hist = keys;                         // 2 * PF_CAP = 16384 counters
            for (uint32_t q = qlo; q < min(qhi, a.nq); ++q) {
                for (uint32_t i = tid; i < 2 * PF_CAP; i += PF_THREADS) hist[i] = 0;
                if (tid == 0) s_n = 0;
                __syncthreads();
                const uint32_t QL = a.q_len[q];
                for (uint32_t p = tid; p < NK; p += PF_THREADS) {
                    int self;
                    const uint32_t k = pf_kmer(tl + p, self);
                    if (self < PF_MINSELF) continue;
                    const uint2 r = a.table[k];
                    // postings of a row are sorted by (q, pos): binary search the sub-range of this query
                    uint32_t lo = 0, hi2 = r.y;
                    while (lo < hi2) { const uint32_t mid = (lo + hi2) >> 1; if ((a.postings[r.x + mid] >> 16) < q) lo = mid + 1; else hi2 = mid; }
                    for (uint32_t c = lo; c < r.y; ++c) {
                        const uint32_t post = a.postings[r.x + c];
                        if ((post >> 16) != q) break;
                        const uint32_t d = (QL + p - (post & 0xFFFFu) - 1) & 0xFFFFu;
                        if (d > 16383u) continue;
                        atomicAdd(&hist[d], 1u);
                    }
                }
                __syncthreads();
                uint32_t best = 0;
                const uint8_t *Q = a.q_mu + a.q_off[q];
                for (uint32_t d = tid; d < 2 * PF_CAP; d += PF_THREADS) {
                    if (hist[d] < 2) continue;
                    int i0 = (int) QL - (int) d - 1; if (i0 < 0) i0 = 0;
                    int j0 = (int) d + 1 - (int) QL; if (j0 < 0) j0 = 0;
                    int hi3 = (int) QL - 1; if ((int) QL + (int) TL - (int) d - 2 < hi3) hi3 = (int) QL + (int) TL - (int) d - 2;
                    const int len = hi3 - i0 + 1;
                    int F = 0, Bst = 0;
                    for (int k = 0; k < len; ++k) {
                        F += mat[Q[i0 + k] * 36 + tl[j0 + k]];
                        if (F > Bst) Bst = F;
                        else if (F < 0) F = 0;
                    }
                    if (Bst > 0) best = max(best, (uint32_t) (Bst >= 65535 ? 65534 : Bst));
                }
                if (best) atomicMax(&s_n, best);
                __syncthreads();
                if (tid == 0 && s_n > 0) {
                    const uint32_t pos = atomicAdd(a.out_n, 1u);
                    if (pos < a.capacity) { a.out_q[pos] = q; a.out_t[pos] = t; a.out_score[pos] = s_n; }
                }
                __syncthreads();
            }
        } else if (expect >= 2) {
            // ---- expand the hits of this chunk into LDS keys
            for (uint32_t p = tid; p < NK; p += PF_THREADS) {
                int self;
                const uint32_t k = pf_kmer(tl + p, self);
                if (self < PF_MINSELF) continue;
                const uint2 r = a.table[k];
                for (uint32_t c = 0; c < r.y; ++c) {
                    const uint32_t post = a.postings[r.x + c];
                    const uint32_t q = post >> 16, qp = post & 0xFFFFu;
                    if (q < qlo || q >= qhi) continue;
                    const uint32_t d = (a.q_len[q] + p - qp - 1) & 0xFFFFu;
                    if (d > 16383u) continue;
                    const uint32_t pos = atomicAdd(&s_n, 1u);
                    if (pos < PF_CAP) keys[pos] = (q << 14) | d;
                }
            }
            __syncthreads();
            const uint32_t n = min(s_n, (uint32_t) PF_CAP);
            uint32_t np2 = 2;
            while (np2 < n) np2 <<= 1;
            for (uint32_t i = n + tid; i < np2; i += PF_THREADS) keys[i] = 0xFFFFFFFFu;
            __syncthreads();
            // ---- bitonic sort of keys[0..np2)
            for (uint32_t k2 = 2; k2 <= np2; k2 <<= 1) {
                for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = tid; i < np2; i += PF_THREADS) {
                        const uint32_t ixj = i ^ j;
                        if (ixj > i) {
                            const uint32_t x = keys[i], y = keys[ixj];
                            const bool up = (i & k2) == 0;
                            if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                        }
                    }
                    __syncthreads();
                }
            }
            // ---- two-hit diagonals: first element of every run of >= 2 equal keys; Kadane on the diagonal
            for (uint32_t i = tid; i < n; i += PF_THREADS) {
                uint32_t sc = 0;
                const uint32_t key = keys[i];
                if (i + 1 < n && keys[i + 1] == key && (i == 0 || keys[i - 1] != key)) {
                    const uint32_t q = key >> 14;
                    const int d = (int) (key & 16383u);
                    const int QL = (int) a.q_len[q];
                    const uint8_t *Q = a.q_mu + a.q_off[q];
                    int i0 = QL - d - 1; if (i0 < 0) i0 = 0;                    // diag.h:51-92
                    int j0 = d + 1 - QL; if (j0 < 0) j0 = 0;
                    int hi = QL - 1; if (QL + (int) TL - d - 2 < hi) hi = QL + (int) TL - d - 2;
                    const int len = hi - i0 + 1;
                    int F = 0, Bst = 0;
                    for (int k = 0; k < len; ++k) {
                        F += mat[Q[i0 + k] * 36 + tl[j0 + k]];
                        if (F > Bst) Bst = F;
                        else if (F < 0) F = 0;
                    }
                    sc = (uint32_t) (Bst > 0 ? (Bst >= 65535 ? 65534 : Bst) : 0);
                }
                score[i] = sc;
            }
            __syncthreads();
            // ---- per query: the head of each run of equal q takes the maximum and emits
            for (uint32_t i = tid; i < n; i += PF_THREADS) {
                const uint32_t q = keys[i] >> 14;
                if (i > 0 && (keys[i - 1] >> 14) == q) continue;
                uint32_t best = 0;
                for (uint32_t k = i; k < n && (keys[k] >> 14) == q; ++k) best = max(best, score[k]);
                if (best > 0) {
                    const uint32_t pos = atomicAdd(a.out_n, 1u);
                    if (pos < a.capacity) { a.out_q[pos] = q; a.out_t[pos] = t; a.out_score[pos] = best; }
                }
            }
        }
        __syncthreads();
        if (!s_more) break;
        chunk_lo = s_chunk_hi;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// host: index build (P10) + launch
// ---------------------------------------------------------------------------------------------
int rsk_build_mudex(rsk_db *db)
{
    if (db->mudex_built) return RSK_OK;
    if (db->n > 65535) { rsk_set_error("k-mer prefilter: at most 65535 query chains (uint16 query index, prefiltermu.cpp:296)"); return RSK_E_RANGE; }
    std::vector<uint64_t> ent;
    static const int o[5] = { 0, 1, 2, 5, 6 };
    for (uint32_t q = 0; q < db->n; ++q) {
        const uint8_t *s = &db->h_mu[db->off[q]];
        const uint32_t L = db->len[q];
        for (uint32_t p = 0; p + 7 <= L; ++p) {
            uint32_t k = 0;
            int self = 0;
            for (int i = 0; i < 5; ++i) { const uint32_t l = s[p + o[i]]; k = k * 36 + l; self += rsk_mu_s8[l * 36 + l]; }
            if (self < PF_MINSELF) continue;
            ent.push_back(((uint64_t) k << 32) | ((uint64_t) q << 16) | p);
        }
    }
    std::sort(ent.begin(), ent.end());
    std::vector<uint32_t> postings(ent.size()), ukmer, ustart;
    for (size_t i = 0; i < ent.size(); ++i) {
        postings[i] = (uint32_t) (ent[i] & 0xFFFFFFFFu);
        const uint32_t k = (uint32_t) (ent[i] >> 32);
        if (ukmer.empty() || ukmer.back() != k) { ukmer.push_back(k); ustart.push_back((uint32_t) i); }
    }
    ustart.push_back((uint32_t) ent.size());
    uint32_t *d_uk = nullptr, *d_us = nullptr;
    RSK_HIP(hipMalloc((void **) &db->d_pf_table, (size_t) PF_DICT * sizeof(uint2)));
    RSK_HIP(hipMemset(db->d_pf_table, 0, (size_t) PF_DICT * sizeof(uint2)));
    RSK_HIP(hipMalloc((void **) &db->d_pf_postings, std::max<size_t>(postings.size(), 1) * 4));
    RSK_HIP(hipMemcpy(db->d_pf_postings, postings.data(), postings.size() * 4, hipMemcpyHostToDevice));
    if (!ukmer.empty()) {
        RSK_HIP(hipMalloc((void **) &d_uk, ukmer.size() * 4));
        RSK_HIP(hipMalloc((void **) &d_us, ustart.size() * 4));
        RSK_HIP(hipMemcpy(d_uk, ukmer.data(), ukmer.size() * 4, hipMemcpyHostToDevice));
        RSK_HIP(hipMemcpy(d_us, ustart.data(), ustart.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_pf_fill_table, dim3((unsigned) ((ukmer.size() + 255) / 256)), dim3(256), 0, 0, d_uk, d_us,
                           (uint32_t) ukmer.size(), (uint2 *) db->d_pf_table);
        RSK_HIP(hipGetLastError());
        RSK_HIP(hipDeviceSynchronize());
        (void) hipFree(d_uk);
        (void) hipFree(d_us);
    }
    db->pf_postings = postings.size();
    db->hbm_bytes += (size_t) PF_DICT * sizeof(uint2) + postings.size() * 4;
    db->mudex_built = true;
    return RSK_OK;
}

extern "C" int rsk_mu_prefilter_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, uint32_t *d_out_q, uint32_t *d_out_t,
                                    uint32_t *d_out_score, size_t capacity, uint32_t *d_n)
{
    if (!ctx || !q || !t || !d_out_q || !d_out_t || !d_out_score || !d_n) { rsk_set_error("rsk_mu_prefilter_dev: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_prefilter_dev: chain set has no Mu letters"); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = pf_upload_tables(ctx);
    if (rc != RSK_OK) return rc;
    if ((rc = rsk_build_mudex(const_cast<rsk_db *>(q))) != RSK_OK) return rc;
    for (uint32_t L : t->len)
        if (L > 65534) { rsk_set_error("rsk_mu_prefilter_dev: target longer than 65534"); return RSK_E_RANGE; }
    uint32_t *d_over = nullptr;
    RSK_HIP(hipMalloc((void **) &d_over, 4));
    RSK_HIP(hipMemsetAsync(d_over, 0, 4, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_n, 0, 4, ctx->stream));
    pf_args a = {};
    a.table = (const uint2 *) q->d_pf_table; a.postings = q->d_pf_postings;
    a.q_mu = q->d_mu; a.q_off = q->d_off; a.q_len = q->d_len;
    a.t_mu = t->d_mu; a.t_off = t->d_off; a.t_len = t->d_len; a.nt = t->n; a.nq = q->n;
    a.out_q = d_out_q; a.out_t = d_out_t; a.out_score = d_out_score;
    a.capacity = (uint32_t) std::min<size_t>(capacity, 0xFFFFFFFFu);
    a.out_n = d_n; a.overflow = d_over;
    uint32_t maxTL = 0;
    for (uint32_t L : t->len) maxTL = std::max(maxTL, L);
    const size_t lds = (size_t) PF_CAP * 8 + 1040 * 4 + 1312 + (((size_t) maxTL + 31) & ~15u);
    if (lds > 163000) { rsk_set_error("rsk_mu_prefilter_dev: target of %u residues does not fit the LDS staging", maxTL); (void) hipFree(d_over); return RSK_E_RANGE; }
    RSK_HIP(hipFuncSetAttribute((const void *) k_prefilter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (t->n) hipLaunchKernelGGL(k_prefilter, dim3(t->n), dim3(PF_THREADS), lds, ctx->stream, a);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    uint32_t over = 0;
    RSK_HIP(hipMemcpyAsync(&over, d_over, 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    (void) hipFree(d_over);
    if (over) { rsk_set_error("rsk_mu_prefilter_dev: more than %d k-mer hits between one target and 64 consecutive queries", PF_CAP); return RSK_E_RANGE; }
    return RSK_OK;
}

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
// MI355X design (r04).  The seed items of one target against a query set range from ~20 per (query, target) pair on
// real SCOP40 letters to ~430 on low-complexity sets (BASELINE configs[2]'s synthetic chains: 5.9e10 items, a two-hit
// diagonal on 99 % of the pairs), so nothing here is sized by "hits are rare":
//   * index: postings (q << 16 | pos) grouped by k-mer row and SORTED inside a row (one 64-bit radix sort of
//     (k-mer, posting) at build time), plus a direct 36^5-entry (start, end) table in HBM (484 MB of 288 GB).
//   * scan: one workgroup per target walks the query set in SPANS of consecutive queries whose diagonals fit two LDS
//     bitmaps ("seen once" / "seen twice": one bit per (query, diagonal), QL + TL - 1 bits per query).  Because the
//     rows are sorted, the postings of a span are a contiguous piece of every index row (two binary searches), so every
//     seed item is read exactly ONCE, turned into a bit address and ORed into the bitmaps -- no key list in HBM, no
//     sort, no hash, no special case for dense queries (r01-r03 scattered 4-byte keys into an HBM scratch -- 236 GB on
//     the synthetic set, in five launches -- and re-read a 64-query bucket once per 4096 keys).
//   * the set bits of "seen twice" are the two-hit diagonals: they are compacted into an LDS list (dense lanes whatever
//     the density), one thread per diagonal runs the Kadane scan four residues per iteration with ONE LDS byte gather
//     per cell (a 36 x 256 table addressed by the (query letter, target letter) byte pair that v_perm_b32 puts together),
//     per query the best is kept in LDS and appended with one atomic per wave.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "rsk_internal.h"
#include "rsk_tables_data.h"

#ifndef PF_THREADS
#define PF_THREADS 512
#endif
#define PF_QSPAN 2048             // queries per span at most (LDS: info + best score per query)
#ifndef PF_BW
#define PF_BW 4096
#endif
#ifndef PF_LDS_BUDGET
#define PF_LDS_BUDGET 81000   // bytes of LDS a workgroup may use (two workgroups per CU)
#endif
//      PF_BW                     // words of each diagonal bitmap (131,072 diagonals per span)
#define PF_LIST 4096              // two-hit diagonals scored per round
#define PF_DICT 60466176u         // 36^5
#define PF_LONGROW 64             // index row pieces from this length on are walked by a whole wave
#define PF_MINSELF 36             // MIN_KMER_PAIR_SCORE prefiltermuparams.h:22
#define PF_MAXDIAG 16384u         // diagonals above 16383 are dropped (prefiltermu.cpp:254)

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
// Pass 0 counts the rows a position goes to (poscnt, indexed by the padded residue offset); after an exclusive scan
// pass 1 writes the position's (k-mer << 32 | posting) keys to its own range of the key array -- no atomics; the radix
// sort then groups the keys by k-mer and orders every row by (query, position).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf_hood(const uint8_t *mu, const uint32_t *off, const uint32_t *len, int mode, int pass,
                                                 uint64_t *poscnt, const uint64_t *posoff, unsigned long long *keys)
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
    const uint32_t o = off[q];
    const uint8_t *s = mu + o;
    for (uint32_t p = tid; p + 7 <= L; p += blockDim.x) {
        uint32_t mine = 0;
        unsigned long long *out = pass ? keys + posoff[o + p] : nullptr;
        const unsigned long long post = ((unsigned long long) q << 16) | p;
        auto emit = [&](uint32_t code) {
            if (pass) out[mine] = ((unsigned long long) code << 32) | post;
            ++mine;
        };
        const uint32_t a0 = s[p], a1 = s[p + 1], a2 = s[p + 2], a3 = s[p + 5], a4 = s[p + 6];
        const int self = c_mu_s8[a0 * 37] + c_mu_s8[a1 * 37] + c_mu_s8[a2 * 37] + c_mu_s8[a3 * 37] + c_mu_s8[a4 * 37];
        if (self >= PF_MINSELF) {
            if (mode != 2) emit((((a0 * 36 + a1) * 36 + a2) * 36 + a3) * 36 + a4);
            if (mode != 0) {
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
                                    emit(c3 * 36 + sl[a4][i4]);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (!pass) poscnt[o + p] = mine;
    }
}

// sorted keys -> postings + the (start, end) of every k-mer row (the table was zeroed: an absent k-mer reads (0, 0))
__global__ void k_pf_rows(const unsigned long long *keys, size_t n, uint32_t *postings, uint2 *table)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const uint32_t code = (uint32_t) (k >> 32);
    postings[i] = (uint32_t) k;
    if (i == 0 || (uint32_t) (keys[i - 1] >> 32) != code) table[code].x = (uint32_t) i;
    if (i + 1 == n || (uint32_t) (keys[i + 1] >> 32) != code) table[code].y = (uint32_t) (i + 1);
}

struct pf_args {
    const uint2 *table;            // [36^5] (start, end) into postings
    const uint32_t *postings;      // q << 16 | pos, grouped by k-mer row, ascending inside a row
    const uint8_t *q_mu; const uint32_t *q_off; const uint32_t *q_len;
    const uint8_t *t_mu; const uint32_t *t_off; const uint32_t *t_len;
    const uint32_t *t_order;       // the targets of this launch, longest first (workgroup b scans target t_order[b])
    uint32_t nt, nq;
    uint32_t *out_q, *out_t, *out_score;
    uint32_t capacity;
    uint32_t *out_n;
    unsigned long long *stat;      // [0] seed items, [1] two-hit diagonals, [2] spans, [3] scoring rounds, [4] diagonal cells scored
    uint32_t tl_cap;               // target letters staged in LDS up to this length
    uint32_t dbg;                  // RSK_PF_DEBUG: 1 = stop after the seed walk of every span (timing experiments)
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

// first index in [lo, hi) of the ascending postings whose value is >= key
__device__ __forceinline__ uint32_t pf_lower_bound(const uint32_t *post, uint32_t lo, uint32_t hi, uint32_t key)
{
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (post[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// LDS of one workgroup (bytes): info + qmax + two bitmaps + list + Kadane table + scalars + target letters
#define PF_LDS_FIXED ((size_t) PF_QSPAN * 8 + (size_t) PF_BW * 8 + (size_t) PF_LIST * 4 + 36 * 256 + 64 * 4)

__global__ __launch_bounds__(PF_THREADS) void k_prefilter(pf_args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *info = (uint32_t *) smem;                                  // PF_QSPAN: (first bitmap word << 16) | QL of the span's queries
    uint32_t *qmax = info + PF_QSPAN;                                    // PF_QSPAN: best diagonal score per query of the span
    uint32_t *seen1 = qmax + PF_QSPAN;                                   // PF_BW: (query, diagonal) seen at least once
    uint32_t *seen2 = seen1 + PF_BW;                                     // PF_BW: ... at least twice = two-hit diagonals
    uint32_t *list = seen2 + PF_BW;                                      // PF_LIST: two-hit diagonals of a round ((query - qa) << 14 | diag); row queue during the walk
    signed char *tab = (signed char *) (list + PF_LIST);                 // 36 x 256: Mu_S_ij_i8[q][t] at q * 256 + t
    uint32_t *sv = (uint32_t *) (tab + 36 * 256);                        // 64 scalars / scan scratch
    uint8_t *tl_lds = (uint8_t *) (sv + 64);                             // target letters (when they fit: a.tl_cap)
    uint32_t &s_rows = sv[0], &s_end = sv[1], &s_words = sv[2];
    uint32_t *wsum = sv + 16;                                            // per-wave partial sums of the block scans

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t t = a.t_order[blockIdx.x];
    for (int i = tid; i < 36 * 256; i += PF_THREADS) tab[i] = (i & 255) < 36 ? c_mu_s8[(i >> 8) * 36 + (i & 255)] : (signed char) 0;
    const uint32_t TL = a.t_len[t];
    const uint8_t *T = a.t_mu + a.t_off[t];
    const uint8_t *tl = T;                                               // chains longer than the LDS staging are read in place
    if (TL <= a.tl_cap) {
        for (uint32_t i = tid; i < TL + 8; i += PF_THREADS) tl_lds[i] = i < TL ? T[i] : (uint8_t) 0;
        tl = tl_lds;
    }
    __syncthreads();
    if (TL < 7) return;
    const uint32_t NK = TL - 6;

    // exclusive scan of one value per thread over the workgroup; returns the thread's offset, `total` for everyone
    auto block_scan = [&](uint32_t v, uint32_t &total) -> uint32_t {
        uint32_t inc = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = __shfl_up(inc, s, 64);
            if (lane >= s) inc += o;
        }
        __syncthreads();                                                 // wsum of the previous scan has been read by everyone
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < PF_THREADS / 64; ++w) { const uint32_t x = wsum[w]; if (w < wid) base += x; tot += x; }
        total = tot;
        return base + inc - v;
    };

    unsigned long long my_cells = 0;                                     // statistics: cells the diagonal scans visit
    // Kadane over the whole diagonal d of query q (FindHSP prefiltermu.cpp:12, diag.h:51-92), score clamped to u16.
    // `F += s; if (F > B) B = F; else if (F < 0) F = 0` is B = max(B, F); F = max(F, 0): F > B >= 0 leaves F alone.
    auto diag_score = [&](uint32_t q, int d) -> uint32_t {
        const int QL = (int) a.q_len[q];
        const uint8_t *Q = a.q_mu + a.q_off[q];
        int i0 = QL - d - 1; if (i0 < 0) i0 = 0;
        int j0 = d + 1 - QL; if (j0 < 0) j0 = 0;
        int hi = QL - 1; if (QL + (int) TL - d - 2 < hi) hi = QL + (int) TL - d - 2;
        int len = hi - i0 + 1;
        if (len > 0) my_cells += (unsigned) len;
        int F = 0, Bst = 0;
        auto step = [&](uint32_t pair) {                                 // pair = query letter << 8 | target letter
            F += tab[pair];
            Bst = max(Bst, F);
            F = max(F, 0);
        };
        const uint8_t *qp = Q + i0, *tp = tl + j0;
        while (len > 0 && ((uintptr_t) qp & 3)) { step(((uint32_t) *qp++ << 8) | *tp++); --len; }
        if (len >= 4) {
            // four residues per iteration: one aligned dword of query letters, the target letters through a sliding pair
            // of aligned dwords (chains and the LDS staging are padded, reading up to 3 bytes past the end is safe);
            // v_perm_b32 interleaves them into two (q, t) byte pairs per dword = two table addresses
            const uint32_t tsh = (uint32_t) ((uintptr_t) tp & 3);
            const uint32_t *tw = (const uint32_t *) (tp - tsh);
            uint32_t ta = *tw++;
            // (Measured and not kept, r06: sixteen residues per iteration with the NEXT iteration's query dwords requested before
            // this one's steps -- real SCOP40 letters 93.0 vs 91.6 ms, synthetic 650 vs 626 ms: the scans do not wait for the
            // query letters, a diagonal is a chain of 3 dependent operations per cell and the lanes of a wave end with its
            // longest diagonal.  Phase clocks of the workgroups (RSK_TRACE), real letters: span plan 9 %, seed walk 23-27 %,
            // compaction + scans 62-67 %; 1,024 threads per workgroup with the same or doubled bitmaps: 126 / 108 ms.)
            for (; len >= 4; len -= 4, qp += 4, tp += 4) {
                const uint32_t qw = *(const uint32_t *) qp;
                const uint32_t tb = *tw++;
                const uint32_t t4 = __builtin_amdgcn_alignbyte(tb, ta, tsh);
                ta = tb;
                const uint32_t p01 = __builtin_amdgcn_perm(qw, t4, 0x05010400u);     // [q1 t1 q0 t0]
                const uint32_t p23 = __builtin_amdgcn_perm(qw, t4, 0x07030602u);     // [q3 t3 q2 t2]
                step(p01 & 0xFFFFu);
                step(p01 >> 16);
                step(p23 & 0xFFFFu);
                step(p23 >> 16);
            }
        }
        while (len > 0) { step(((uint32_t) *qp++ << 8) | *tp++); --len; }
        return (uint32_t) (Bst > 0 ? (Bst >= 65535 ? 65534 : Bst) : 0);
    };

    unsigned long long my_items = 0, my_two = 0;
    uint32_t nspans = 0, nrounds = 0;
    // phase clocks of the workgroup (shader cycles seen by its first wave between the barriers that end the phases): plan, clear,
    // seed walk, compaction + scoring, triples -- RSK_TRACE prints their shares
    unsigned long long ph[5] = { 0, 0, 0, 0, 0 }, ph_t = __builtin_amdgcn_s_memtime();
    auto phase = [&](int k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph[k] += now - ph_t; ph_t = now; };
    for (uint32_t qa = 0; qa < a.nq;) {
        // ---- plan the span [qa, qb): consecutive queries whose diagonals (QL + TL - 1 bits each, rounded up to whole
        // words so that a word belongs to one query) fit the bitmaps
        constexpr int PER = PF_QSPAN / PF_THREADS;
        uint32_t w[PER], ql[PER], mine = 0;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const uint32_t q = qa + (uint32_t) tid * PER + r;
            ql[r] = q < a.nq ? a.q_len[q] : 0;
            const uint32_t nb = min(ql[r] + TL - 1, PF_MAXDIAG);
            w[r] = q < a.nq ? (nb + 31) >> 5 : 0;
            mine += w[r];
        }
        if (tid == 0) { s_end = PF_QSPAN; s_rows = 0; }
        uint32_t total;
        uint32_t offw = block_scan(mine, total);                         // (barriers inside: s_end is visible below)
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const uint32_t idx = (uint32_t) tid * PER + r;
            if (offw + w[r] > PF_BW) atomicMin(&s_end, idx);             // the first query that does not fit ends the span
            info[idx] = (min(offw, 0xFFFFu) << 16) | ql[r];
            qmax[idx] = 0;
            offw += w[r];
        }
        __syncthreads();
        const uint32_t nqs = min(min(s_end, (uint32_t) PF_QSPAN), a.nq - qa);       // >= 1: one query needs <= 512 words
        const uint32_t qb = qa + nqs;
        if (tid == 0) s_words = (info[nqs - 1] >> 16) + ((min((info[nqs - 1] & 0xFFFFu) + TL - 1, PF_MAXDIAG) + 31) >> 5);
        __syncthreads();
        const uint32_t nwords = s_words;
        phase(0);
        for (uint32_t i = tid; i < nwords; i += PF_THREADS) { seen1[i] = 0; seen2[i] = 0; }
        __syncthreads();
        ++nspans;
        phase(1);

        // ---- seed walk: every posting of the span's queries in the rows of this target's k-mers, once.  Row pieces of
        // >= PF_LONGROW postings are queued in LDS and split over the lanes of a wave (coalesced posting loads); shorter
        // ones stay with the thread that owns the k-mer position.
        auto item = [&](uint32_t p, uint32_t post) {
            const uint32_t in = info[(post >> 16) - qa];
            const uint32_t d = ((in & 0xFFFFu) + p - (post & 0xFFFFu) - 1) & 0xFFFFu;
            if (d >= PF_MAXDIAG) return;
            const uint32_t bit = ((in >> 16) << 5) + d, m = 1u << (bit & 31);
            if (atomicOr(&seen1[bit >> 5], m) & m) atomicOr(&seen2[bit >> 5], m);
        };
        uint32_t *rowq = list;                                           // (position, start, end) of the queued row pieces
        const uint32_t klo = qa << 16, khi = qb << 16;                   // qb <= 65535 (rsk_build_mudex)
        for (uint32_t base = 0; base < NK; base += PF_THREADS) {
            const uint32_t p = base + tid;
            if (tid == 0) s_rows = 0;
            __syncthreads();
            if (p < NK) {
                int self;
                const uint32_t k = pf_kmer(tl + p, self);
                if (self >= PF_MINSELF) {
                    const uint2 r = a.table[k];
                    if (r.y > r.x) {
                        const uint32_t lo = qa ? pf_lower_bound(a.postings, r.x, r.y, klo) : r.x;
                        const uint32_t hi = qb < a.nq ? pf_lower_bound(a.postings, lo, r.y, khi) : r.y;
                        my_items += hi - lo;
                        if (hi - lo >= PF_LONGROW) {
                            const uint32_t e = atomicAdd(&s_rows, 1u);
                            rowq[3 * e] = p; rowq[3 * e + 1] = lo; rowq[3 * e + 2] = hi;
                        } else {
#ifndef PF_SHORTROW_SERIAL
                            // short row pieces stay with the thread that owns the k-mer position: four postings requested before
                            // the first is used (r04 walked them one dependent load at a time)
                            uint32_t c = lo;
                            for (; c + 4 <= hi; c += 4) {
                                const uint32_t p0 = a.postings[c], p1 = a.postings[c + 1], p2 = a.postings[c + 2], p3 = a.postings[c + 3];
                                item(p, p0); item(p, p1); item(p, p2); item(p, p3);
                            }
                            if (c < hi) {
                                const uint32_t n = hi - c;
                                const uint32_t p0 = a.postings[c], p1 = a.postings[n > 1 ? c + 1 : c], p2 = a.postings[n > 2 ? c + 2 : c];
                                item(p, p0);
                                if (n > 1) item(p, p1);
                                if (n > 2) item(p, p2);
                            }
#else
                            for (uint32_t c = lo; c < hi; ++c) item(p, a.postings[c]);
#endif
                        }
                    }
                }
            }
            __syncthreads();
            const uint32_t nrow = s_rows;
            for (uint32_t e = wid; e < nrow; e += PF_THREADS / 64) {
                const uint32_t rp = rowq[3 * e], rlo = rowq[3 * e + 1], rhi = rowq[3 * e + 2];
                uint32_t c = rlo + lane;
                for (; c + 192 < rhi; c += 256) {                        // four loads in flight per lane
                    const uint32_t p0 = a.postings[c], p1 = a.postings[c + 64], p2 = a.postings[c + 128], p3 = a.postings[c + 192];
                    item(rp, p0); item(rp, p1); item(rp, p2); item(rp, p3);
                }
                for (; c < rhi; c += 64) item(rp, a.postings[c]);
            }
            __syncthreads();
        }
        phase(2);
        if (a.dbg == 1) { qa = qb; continue; }

        // ---- two-hit diagonals = set bits of seen2, compacted into the list in rounds of <= PF_LIST and scored
        // (the waves claim 64 entries at a time: diagonal lengths run from 7 to the chain length along the list, and with a
        // fixed share per wave the round ended when the wave holding the long ones did)
        auto score_list = [&](uint32_t n) {
            if (tid == 0) sv[4] = 0;
            __syncthreads();
            for (;;) {
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(&sv[4], 64u);
                c = (uint32_t) __builtin_amdgcn_readfirstlane((int) c);
                if (c >= n) break;
                const uint32_t i = c + (uint32_t) lane;
                if (i < n) {
                    const uint32_t e = list[i];
                    const uint32_t sc = diag_score(qa + (e >> 14), (int) (e & 16383u));
                    if (sc > 0) atomicMax(&qmax[e >> 14], sc);
                }
            }
            ++nrounds;
            __syncthreads();
        };
        // appends the set bits of `word` (bitmap word index wi) at list[pos...]
        auto expand = [&](uint32_t word, uint32_t wi, uint32_t pos) {
            // the query this word belongs to: last index whose first word is <= wi
            uint32_t lo = 0, hi = nqs - 1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if ((info[mid] >> 16) <= wi) lo = mid; else hi = mid - 1;
            }
            const uint32_t d0 = (wi - (info[lo] >> 16)) << 5;
            while (word) {
                const uint32_t b = (uint32_t) __builtin_ctz(word);
                word &= word - 1;
                list[pos++] = (lo << 14) | (d0 + b);
            }
        };
        uint32_t filled = 0;
        for (uint32_t base = 0; base < nwords; base += PF_THREADS) {
            const uint32_t wi = base + tid;
            const uint32_t word = wi < nwords ? seen2[wi] : 0;
            const uint32_t c = (uint32_t) __builtin_popcount(word);
            uint32_t tot;
            const uint32_t ex = block_scan(c, tot);
            if (tot == 0) continue;
            my_two += c;
            if (tot > PF_LIST) {
                // more than a list of diagonals in 512 words: quarter blocks (128 words hold <= 4096 bits)
                if (filled) { score_list(filled); filled = 0; }
                for (int sub = 0; sub < PF_THREADS / 128; ++sub) {
                    const bool in = (tid >> 7) == sub;
                    uint32_t st;
                    const uint32_t ex2 = block_scan(in ? c : 0, st);
                    if (st == 0) continue;
                    if (in && word) expand(word, wi, ex2);
                    score_list(st);
                }
                continue;
            }
            if (filled + tot > PF_LIST) { score_list(filled); filled = 0; }
            if (word) expand(word, wi, filled + ex);
            filled += tot;
        }
        if (filled) score_list(filled);
        __syncthreads();
        phase(3);

        // ---- one (query, target, score) triple per query with a two-hit diagonal scoring > 0, one atomic per wave
        for (uint32_t i0 = 0; i0 < nqs; i0 += PF_THREADS) {
            const uint32_t i = i0 + tid;
            const uint32_t best = i < nqs ? qmax[i] : 0;
            const unsigned long long m = __ballot(best > 0);
            if (m) {
                uint32_t basepos = 0;
                if (lane == 0) basepos = atomicAdd(a.out_n, (uint32_t) __popcll(m));
                basepos = (uint32_t) __builtin_amdgcn_readfirstlane((int) basepos);
                if (best > 0) {
                    const uint32_t pos = basepos + (uint32_t) __popcll(m & ((1ull << lane) - 1));
                    if (pos < a.capacity) { a.out_q[pos] = qa + i; a.out_t[pos] = t; a.out_score[pos] = best; }
                }
            }
        }
        __syncthreads();
        phase(4);
        qa = qb;
    }
    if (a.stat) {
        if (tid == 0)
            for (int k = 0; k < 5; ++k) atomicAdd(a.stat + 8 + k, ph[k]);
        // wave-reduced statistics (RSK_TRACE / the work counters of the context)
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { my_items += __shfl_xor(my_items, s, 64); my_two += __shfl_xor(my_two, s, 64); my_cells += __shfl_xor(my_cells, s, 64); }
        if (lane == 0) { atomicAdd(a.stat + 0, my_items); atomicAdd(a.stat + 1, my_two); atomicAdd(a.stat + 4, my_cells); }
        if (tid == 0) { atomicAdd(a.stat + 2, (unsigned long long) nspans); atomicAdd(a.stat + 3, (unsigned long long) nrounds); }
    }
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
    if (db->d_pf_postings) { (void) hipFree(db->d_pf_postings); db->d_pf_postings = nullptr; db->hbm_bytes -= db->pf_postings * 4; db->pf_postings = 0; }
    db->mudex_built = false;
    // temporaries through the CALLING context's pool (returned on every exit path); everything on that context's stream
    // (the context that created the set may be another thread's, e.g. the -db loader's)
    rsk_scratch ws(ctx);
    const size_t npos = (size_t) db->npad + 1;
    uint64_t *d_poscnt, *d_posoff;              // 64-bit counts: the scan accumulates in its input type
    void *d_tmp;
    int rc;
    if (!db->d_pf_table) {
        if ((rc = rsk_db_malloc(db, ctx, (void **) &db->d_pf_table, (size_t) PF_DICT * sizeof(uint2))) != RSK_OK) return rc;
        db->hbm_bytes += (size_t) PF_DICT * sizeof(uint2);
    }
    if ((rc = ws.alloc(&d_poscnt, npos)) || (rc = ws.alloc(&d_posoff, npos))) return rc;
    RSK_HIP(hipMemsetAsync(d_poscnt, 0, npos * 8, ctx->stream));
    RSK_HIP(hipMemsetAsync(db->d_pf_table, 0, (size_t) PF_DICT * sizeof(uint2), ctx->stream));
    if (db->n) hipLaunchKernelGGL(k_pf_hood, dim3(db->n), dim3(256), 0, ctx->stream, db->d_mu, db->d_off, db->d_len, mode, 0, d_poscnt,
                                  (const uint64_t *) nullptr, (unsigned long long *) nullptr);
    RSK_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    RSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_poscnt, d_posoff, (int) npos, ctx->stream));
    if ((rc = ws.alloc(&d_tmp, std::max<size_t>(tmp_bytes, 16))) != RSK_OK) return rc;
    RSK_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_poscnt, d_posoff, (int) npos, ctx->stream));
    uint64_t total = 0;
    RSK_HIP(hipMemcpyAsync(&total, d_posoff + (npos - 1), 8, hipMemcpyDeviceToHost, ctx->stream));      // the last entry counts nothing itself
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    if (total > 0xFFFFFFF0ull) { rsk_set_error("k-mer prefilter: %llu index postings exceed 2^32; split the query set", (unsigned long long) total); return RSK_E_RANGE; }
    if ((rc = rsk_db_malloc(db, ctx, (void **) &db->d_pf_postings, std::max<size_t>((size_t) total, 1) * 4)) != RSK_OK) return rc;
    db->hbm_bytes += (size_t) total * 4;
    db->pf_postings = (size_t) total;
    if (total) {
        unsigned long long *d_keys, *d_sorted;
        void *d_sort_tmp;
        if ((rc = ws.alloc(&d_keys, (size_t) total)) || (rc = ws.alloc(&d_sorted, (size_t) total))) return rc;
        hipLaunchKernelGGL(k_pf_hood, dim3(db->n), dim3(256), 0, ctx->stream, db->d_mu, db->d_off, db->d_len, mode, 1, d_poscnt,
                           (const uint64_t *) d_posoff, d_keys);
        RSK_HIP(hipGetLastError());
        // rows by k-mer (bits 32..57: 36^5 < 2^26), inside a row by posting (query << 16 | position)
        size_t sort_bytes = 0;
        RSK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, d_keys, d_sorted, (size_t) total, 0, 58, ctx->stream));
        if ((rc = ws.alloc(&d_sort_tmp, std::max<size_t>(sort_bytes, 16))) != RSK_OK) return rc;
        RSK_HIP(hipcub::DeviceRadixSort::SortKeys(d_sort_tmp, sort_bytes, d_keys, d_sorted, (size_t) total, 0, 58, ctx->stream));
        hipLaunchKernelGGL(k_pf_rows, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, ctx->stream, d_sorted, (size_t) total, db->d_pf_postings,
                           (uint2 *) db->d_pf_table);
        RSK_HIP(hipGetLastError());
    }
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    db->mudex_built = true;
    db->mudex_mode = mode;
    return RSK_OK;
}

extern "C" int rsk_mu_prefilter_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int neighbourhood, uint32_t *d_out_q,
                                    uint32_t *d_out_t, uint32_t *d_out_score, size_t capacity, uint32_t *d_n)
{
    return rsk_mu_prefilter_range_dev(ctx, q, t, neighbourhood, 0, t ? t->n : 0, d_out_q, d_out_t, d_out_score, capacity, d_n);
}

extern "C" int rsk_mu_prefilter_range_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int neighbourhood, uint32_t t_lo, uint32_t t_hi,
                                          uint32_t *d_out_q, uint32_t *d_out_t, uint32_t *d_out_score, size_t capacity, uint32_t *d_n)
{
    if (!ctx || !q || !t || !d_out_q || !d_out_t || !d_out_score || !d_n) { rsk_set_error("rsk_mu_prefilter_dev: NULL argument"); return RSK_E_INVALID; }
    if (t_lo > t_hi || t_hi > t->n) { rsk_set_error("rsk_mu_prefilter_range_dev: target range [%u, %u) outside the set of %u", t_lo, t_hi, t->n); return RSK_E_INVALID; }
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
    // the targets of the range, longest first (the grid's order: a long target's workgroup must not start last)
    const uint32_t ntr = t_hi - t_lo;
    std::vector<uint32_t> order(ntr);
    for (uint32_t k = 0; k < ntr; ++k) order[k] = t_lo + k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return t->len[x] > t->len[y]; });
    uint32_t *d_order;
    if ((rc = ws.alloc(&d_order, (size_t) std::max<uint32_t>(ntr, 1))) != RSK_OK) return rc;
    if (ntr) RSK_HIP(hipMemcpyAsync(d_order, order.data(), (size_t) ntr * 4, hipMemcpyHostToDevice, ctx->stream));
    unsigned long long *d_stat;
    if ((rc = ws.alloc(&d_stat, 16)) != RSK_OK) return rc;
    RSK_HIP(hipMemsetAsync(d_stat, 0, 128, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_n, 0, 4, ctx->stream));
    pf_args a = {};
    a.table = (const uint2 *) q->d_pf_table; a.postings = q->d_pf_postings;
    a.q_mu = q->d_mu; a.q_off = q->d_off; a.q_len = q->d_len;
    a.t_mu = t->d_mu; a.t_off = t->d_off; a.t_len = t->d_len; a.t_order = d_order; a.nt = ntr; a.nq = q->n;
    a.out_q = d_out_q; a.out_t = d_out_t; a.out_score = d_out_score;
    a.capacity = (uint32_t) std::min<size_t>(capacity, 0xFFFFFFFFu);
    a.out_n = d_n;
    a.stat = d_stat;
    uint32_t maxTL = 0;
    for (uint32_t k = t_lo; k < t_hi; ++k) maxTL = std::max(maxTL, t->len[k]);
    // two workgroups per CU (the seed walk is latency-bound): each may use half of the 160 KB; target letters as far as
    // they fit (longer targets are read from HBM in place)
    const uint32_t tl_cap = (uint32_t) std::min<size_t>(maxTL, ((size_t) PF_LDS_BUDGET - PF_LDS_FIXED - 48) & ~(size_t) 15);
    a.tl_cap = tl_cap;
    a.dbg = getenv("RSK_PF_DEBUG") ? (uint32_t) atoi(getenv("RSK_PF_DEBUG")) : 0;
    const size_t lds = PF_LDS_FIXED + (((size_t) tl_cap + 8 + 31) & ~(size_t) 15);
    RSK_HIP(hipFuncSetAttribute((const void *) k_prefilter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));   // depends on the call's targets
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (ntr && q->n) {
        hipLaunchKernelGGL(k_prefilter, dim3(ntr), dim3(PF_THREADS), lds, ctx->stream, a);
        RSK_HIP(hipGetLastError());
    }
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    unsigned long long stat[16] = { 0 };
    RSK_HIP(hipMemcpyAsync(stat, d_stat, 128, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->pf_hits = stat[0];
    ctx->pf_postings = q->pf_postings;
    ctx->pf_twohit = stat[1];
    ctx->pf_cells = stat[4];
    if (getenv("RSK_TRACE"))
        fprintf(stderr, "[prefilter] index postings %zu, seed items %llu, two-hit diagonals %llu (%llu cells); query spans %llu, scoring rounds %llu (targets %u .. %u)\n",
                q->pf_postings, stat[0], stat[1], stat[4], stat[2], stat[3], t_lo, t_hi);
    if (getenv("RSK_TRACE")) {
        const double tot = (double) (stat[8] + stat[9] + stat[10] + stat[11] + stat[12]) + 1e-9;
        fprintf(stderr, "[prefilter] workgroup cycles: span plan %.1f %%, bitmap clear %.1f %%, seed walk %.1f %%, compaction + diagonal scans %.1f %%, triples %.1f %% (%.3g cycles per span)\n",
                100.0 * stat[8] / tot, 100.0 * stat[9] / tot, 100.0 * stat[10] / tot, 100.0 * stat[11] / tot, 100.0 * stat[12] / tot, tot / (double) std::max<unsigned long long>(stat[2], 1));
    }
    return RSK_OK;
}

extern "C" int rsk_mu_prefilter_last_work(rsk_ctx *ctx, uint64_t *seed_items, uint64_t *index_postings, uint64_t *twohit_diagonals, uint64_t *diagonal_cells)
{
    if (!ctx) { rsk_set_error("rsk_mu_prefilter_last_work: NULL context"); return RSK_E_INVALID; }
    if (seed_items) *seed_items = ctx->pf_hits;
    if (index_postings) *index_postings = ctx->pf_postings;
    if (twohit_diagonals) *twohit_diagonals = ctx->pf_twohit;
    if (diagonal_cells) *diagonal_cells = ctx->pf_cells;
    return RSK_OK;
}

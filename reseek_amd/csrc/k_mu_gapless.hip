// k_mu_gapless.hip -- gapless integer Mu-letter local score on gfx950.
//
// Reference semantics (bit-exact): SWFastGapless_Int swgaplessint.cpp:7 == SWFastPinopGapless
// swfastpinopgapless.cpp:6:  x(i,j) = max(0, x(i-1,j-1)) + IntScoreMx_Mu[a_i][b_j], best = max x.
// Because best starts at 0 and only x > 0 matters, H = max(0, H_prev + s) gives the same best.
//
// MI355X design ("ring" kernel; the path is VALU/LDS bound, there is no GEMM in it) -- the r04 form, DESIGN.md 4.1:
//   * Every diagonal of a gapless DP is independent.  Several query chains, each preceded by one separator row, are laid
//     out on a circular array.  A ring is TWO sub-rings of 64*D slots (D = 16: 1024 each; D = 8 for the short tail): sub-ring
//     A lives in the low halves and sub-ring B in the high halves of 64*D ring dwords, and lane l owns the D CONSECUTIVE
//     dwords D*l .. D*l + D - 1 (D VGPRs of running diagonal values, D of per-slot bests).
//   * One wave walks ONE target chain; the target letter is wave-uniform, so the score row of that letter comes from the
//     LDS-resident query profile (one copy, 2048 slots of a D = 16 ring) with D/4 contiguous, conflict-free ds_read_b128;
//     the two LDS row addresses of a pair of letters are ONE v_pk_add_f32 on the address bit patterns (denormals add like
//     the integers they spell; the code object runs with fp32 denormals on).
//   * Values are packed HALF floats scaled by 2^-11 (score n = n / 2048: every integer 0..2048 is exact, and so is every
//     sum the recurrence forms).  v_pk_add_f16 ... clamp is both the add and the max(0, .) floor (the clamp of a float op is
//     [0, 1]); separator rows / pad letters hold -1.0 and reset a diagonal.  The best per ring slot over the two letters of
//     a pair-step is ONE v_pk_maximum3_f16 (gfx950) => 0.75 packed VALU op per cell.
//     The clamp is also a CEILING at 2048: a pair whose best reads 2048 (two chains of >= 512 residues; in practice self
//     pairs of long chains) is scored again, exactly, in int32 by the wave that found it before anything is reported.
//   * A diagonal moves one row per target letter.  Because the two halves of a dword belong to DIFFERENT sub-rings, that is
//     a move by one whole dword: the add of the next letter reads its neighbour's register (no instruction), and only the
//     dword that crosses to the next lane costs one v_mov_b32_dpp wave_ror:1 per letter (lane 63 -> lane 0 closes both
//     sub-rings).  Per pair of letters and lane: 3*D packed add / max3 + 2 DPP + 1 address op.
//     (r01-r04a kept CONSECUTIVE rows in the two halves of a dword -- a value then moves half a dword per letter, which took a
//     second, row-shifted copy of the profile for the odd letters: twice the LDS per slot, half the D.)
//   * Sub-rings are bin-packed (best fit decreasing, <= 64 members) and paired fullest first; a work item = (ring, block of
//     <= 1024 targets claimed longest first); per target the per-slot bests are reduced per query (in-lane, then LDS atomic
//     max); hit records {query, target, score} above a threshold are appended by the kernel, the dense uint16 matrix
//     out[query][target] is optional.
// A simple per-pair kernel covers chains too long for a ring and returns best-cell positions.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <map>

#include "rsk_dev_tables.h"

typedef short v2s __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define GL_SCALE 2048               // scores are stored as n / GL_SCALE in half floats
#define GL_CAP_BITS 0x3C00          // 1.0 = the clamp's ceiling = score 2048
#define GL_RESET_BITS 0xBC00        // -1.0: pad letters and separator rows
#define RING_TB_MAX 1024     // targets per work item (a workgroup builds one LDS profile per item), see ring_target_block
#define RING_SUB 1024        // slots of one sub-ring of a D = 16 ring: the largest query block (1 + L rounded up to 4)

// ---------------------------------------------------------------------------------------------
// host: pack queries onto rings
// ---------------------------------------------------------------------------------------------
static inline uint32_t qblock(uint32_t L) { return (1 + L + 3) / 4 * 4; }

int rsk_build_rings(rsk_db *db)
{
    const uint32_t n = db->n;
    std::vector<uint8_t> letters, laneq;
    std::vector<uint32_t> qids;
    std::vector<rsk_ring> rings;
    db->long_q.clear();
    // Best-fit-decreasing bin packing of the query blocks into 1024-slot sub-rings (capacity quantum, not order, is
    // what a ring loses: consecutive chains of a length-sorted set leave 1024 mod block unused); two bins make one ring
    // (sub-ring A in the low halves of the ring dwords, B in the high halves).  Any grouping is valid because the chain set
    // is then PROCESSED in ring order: position p of the permuted order `ring_perm` is the p-th ring member; in
    // self-triangle mode a ring takes the targets at positions >= its first member, so every unordered pair is scored at
    // least once (the score is symmetric) and stored at out[min][max].  Chains too long for a ring come last in the
    // order (per-pair kernel).
    std::vector<uint32_t> items;
    for (uint32_t i = 0; i < n; ++i) {
        if (qblock(db->len[i]) > RING_SUB) db->long_q.push_back(i);
        else items.push_back(i);
    }
    std::stable_sort(items.begin(), items.end(), [&](uint32_t x, uint32_t y) { return qblock(db->len[x]) > qblock(db->len[y]); });
    struct bin { uint32_t used = 0; std::vector<uint32_t> members; };
    std::vector<bin> bins;
    std::multimap<uint32_t, uint32_t> by_free;          // free slots -> bin index
    const uint32_t CAP = RING_SUB, NQCAP = 64;          // 128 members per ring: the per-wave result words in LDS
    for (uint32_t q : items) {
        const uint32_t b = qblock(db->len[q]);
        auto it = by_free.lower_bound(b);               // tightest bin that still takes the block
        while (it != by_free.end() && bins[it->second].members.size() >= NQCAP) ++it;
        uint32_t bi;
        if (it == by_free.end()) { bi = (uint32_t) bins.size(); bins.emplace_back(); }
        else { bi = it->second; by_free.erase(it); }
        bins[bi].used += b;
        bins[bi].members.push_back(q);
        if (bins[bi].used < CAP) by_free.insert({ CAP - bins[bi].used, bi });
    }
    // fullest bins first, so that the two bins of a ring are about equally full and only the last rings are small
    std::stable_sort(bins.begin(), bins.end(), [](const bin &x, const bin &y) { return x.used > y.used; });
    for (size_t b0 = 0; b0 < bins.size(); b0 += 2) {
        const bin *sub[2] = { &bins[b0], b0 + 1 < bins.size() ? &bins[b0 + 1] : nullptr };
        const uint32_t bestD = sub[0]->used <= 512 ? 8 : 16;         // sub[0] is the fuller one
        rsk_ring r;
        r.D = bestD;
        r.nq = (uint32_t) (sub[0]->members.size() + (sub[1] ? sub[1]->members.size() : 0));
        r.min_q = 0;                                    // position in the permuted order, set below
        r.letters_off = (uint32_t) letters.size();
        r.laneq_off = (uint32_t) laneq.size();
        r.qid_off = (uint32_t) qids.size();
        const uint32_t SR = 64 * bestD;                 // slots of a sub-ring
        letters.resize(letters.size() + 2 * SR, 0xFF);
        laneq.resize(laneq.size() + 2 * (SR / 4), 0xFF);
        uint32_t k = 0;
        for (int h = 0; h < 2; ++h) {
            if (!sub[h]) break;
            uint8_t *rl = &letters[r.letters_off + h * SR];
            uint8_t *lq = &laneq[r.laneq_off + h * (SR / 4)];
            uint32_t s = 0;
            for (uint32_t q : sub[h]->members) {
                const uint32_t L = db->len[q], b = qblock(L);
                memcpy(rl + s + 1, &db->h_mu[db->off[q]], L);     // slot s = separator, then the L rows
                for (uint32_t g = s / 4; g < (s + b) / 4; ++g) lq[g] = (uint8_t) k;   // granule g = slots [4g, 4g+4)
                qids.push_back(q);
                s += b;
                ++k;
            }
        }
        rings.push_back(r);
    }
    // sort by D so each class is one launch
    std::stable_sort(rings.begin(), rings.end(), [](const rsk_ring &a, const rsk_ring &b) { return a.D < b.D; });
    // processing order: ring members in launch order, then the long chains
    std::vector<uint32_t> perm;
    perm.reserve(n);
    for (rsk_ring &r : rings) {
        r.min_q = (uint32_t) perm.size();
        for (uint32_t k = 0; k < r.nq; ++k) perm.push_back(qids[r.qid_off + k]);
    }
    for (uint32_t q : db->long_q) perm.push_back(q);
    db->h_ring_perm = perm;
    db->rings = rings;
    db->ring_slots_total = 0;
    for (auto &r : rings) db->ring_slots_total += 128ull * r.D;      // two sub-rings of 64 * D slots
    auto up = [&](void **d, const void *h, size_t bytes) -> int {
        *d = nullptr;
        if (!bytes) return RSK_OK;
        { const int rc_ = rsk_db_malloc(db, nullptr, d, bytes); if (rc_ != RSK_OK) return rc_; }
        RSK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        db->hbm_bytes += bytes;
        return RSK_OK;
    };
    int rc;
    if ((rc = up((void **) &db->d_ring_tab, rings.data(), rings.size() * sizeof(rsk_ring))) != RSK_OK) return rc;
    if ((rc = up((void **) &db->d_ring_letters, letters.data(), letters.size())) != RSK_OK) return rc;
    if ((rc = up((void **) &db->d_ring_laneq, laneq.data(), laneq.size())) != RSK_OK) return rc;
    if ((rc = up((void **) &db->d_ring_qid, qids.data(), qids.size() * 4)) != RSK_OK) return rc;
    if ((rc = up((void **) &db->d_ring_perm, perm.data(), perm.size() * 4)) != RSK_OK) return rc;
    db->rings_built = true;
    return RSK_OK;
}

// ---------------------------------------------------------------------------------------------
// ring kernel
// ---------------------------------------------------------------------------------------------
// Hit emission (optional): every scored pair whose score reaches min_score is appended as a record {q_base + query,
// t_base + target, score} (self triangle: {min, max} of the two chain ids, each unordered pair once) -- what a search keeps
// of the pair space, a few records per million pairs, instead of an n x n matrix the host would have to scan.
// *count is the number of hits found; records beyond cap are not stored (the caller enlarges and repeats).
struct gl_hits {
    uint32_t *rec;           // [cap][3], NULL = no emission
    uint32_t *count;
    uint32_t cap, min_score, q_base, t_base;
};
__device__ __forceinline__ void gl_emit(const gl_hits &h, uint32_t a, uint32_t b, uint32_t score)
{
    const uint32_t k = atomicAdd(h.count, 1u);
    if (k < h.cap) { h.rec[3 * (size_t) k] = a; h.rec[3 * (size_t) k + 1] = b; h.rec[3 * (size_t) k + 2] = score; }
}

template <int D> struct RingGeom {
    static constexpr int SR = 64 * D;          // slots of one sub-ring (A = low halves, B = high halves of the ring dwords)
    static constexpr int RSB = 4 * SR;         // bytes per profile row: one dword per ring dword
    static constexpr int NROWS = 37;           // 36 letters + the pad letter (all -1.0)
    static constexpr int PROF_BYTES = NROWS * RSB;
    static constexpr int NQMAX = 128;          // members of a ring (64 per sub-ring, rsk_build_rings)
    static constexpr int M = D / 4;            // b128 groups per lane
};

template <int D, int NW> constexpr size_t ring_lds_bytes()
{
    return (size_t) RingGeom<D>::PROF_BYTES + (size_t) NW * RingGeom<D>::NQMAX * 4 + 1312 + 2 * RingGeom<D>::SR + 16;
}

// (a + b) clamped to [0, 1] per half
__device__ __forceinline__ int pk_add_clamp01(int a, int b)
{
    int r;
    asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int pk_max3_f16(int a, int b, int c)
{
    int r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// maximum of two packed values that are known to be >= 0: non-negative half floats order like their bit patterns
__device__ __forceinline__ int pk_max(int a, int b)
{
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b)));
}
__device__ __forceinline__ unsigned short gl_half_bits(int score)      // score / 2048 as a half float; |score| <= 2048
{
    return __builtin_bit_cast(unsigned short, (_Float16) ((float) score * (1.0f / GL_SCALE)));
}
__device__ __forceinline__ uint32_t gl_score_of_bits(int bits)         // inverse, bits of a half float in [0, 1]
{
    return (uint32_t) ((float) __builtin_bit_cast(_Float16, (unsigned short) bits) * (float) GL_SCALE);
}
__device__ __forceinline__ int dpp_wave_ror1(int x)
{
    return __builtin_amdgcn_mov_dpp(x, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);      // no tied operand: the source stays live
}

// One pair of target letters (c0, c1).  A lane owns D CONSECUTIVE ring dwords (ring dword D*lane + k, k = 0..D-1); a
// dword holds slot D*lane + k of sub-ring A in its low half and the same slot of sub-ring B in its high half, so "every
// value moves up one row per letter" is a move by one whole dword: the add of the next letter simply takes its
// neighbour's register as the source (no instruction), and only the dword that crosses to the next lane costs ONE
// v_mov_b32_dpp wave_ror:1 (lane 63 -> lane 0 closes both sub-rings).  G[k] = the value standing at dword k before the
// letter's score is added.  The LDS profile keeps the b128 blocks of all lanes contiguous (block m of lane l at
// m*1024 + 16*l: conflict-free), only the builder knows the slot permutation.
template <int D>
__device__ __forceinline__ void ring_pairstep(int (&G)[D], int (&E)[D], v2f lane_pp, unsigned c0, unsigned c1)
{
    constexpr int M = D / 4;
    constexpr int RSB = RingGeom<D>::RSB;
    v4i S[M], T[M];
    // row offsets are wave-uniform: multiply on the scalar unit, then ONE VALU op for the two addresses of the step --
    // v_pk_add_f32 on the BIT PATTERNS: an LDS byte address (< 2^23) read as a float is a denormal, and denormals add like
    // the integers they spell (this code object runs with fp32 denormals enabled: .amdhsa_float_denorm_mode_32 3)
    const unsigned o0 = (unsigned) __builtin_amdgcn_readfirstlane((int) (c0 * (unsigned) RSB));
    const unsigned o1 = (unsigned) __builtin_amdgcn_readfirstlane((int) (c1 * (unsigned) RSB));
    const unsigned long long oo = ((unsigned long long) o1 << 32) | o0;
    unsigned long long ad;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(ad) : "v"(lane_pp), "s"(oo));
    const unsigned a0 = (unsigned) ad, a1 = (unsigned) (ad >> 32);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        S[m] = *(const v4i __attribute__((address_space(3))) *) (uintptr_t) (a0 + m * 1024);
        T[m] = *(const v4i __attribute__((address_space(3))) *) (uintptr_t) (a1 + m * 1024);
    }
    int g1[D], g2[D];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) g1[k] = pk_add_clamp01(G[k], S[k >> 2][k & 3]);
    const int x1 = dpp_wave_ror1(g1[D - 1]);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        g2[k] = pk_add_clamp01(k ? g1[k - 1] : x1, T[k >> 2][k & 3]);
        E[k] = pk_max3_f16(E[k], g1[k], g2[k]);
    }
    G[0] = dpp_wave_ror1(g2[D - 1]);
#pragma unroll
    for (int k = 1; k < D; ++k) G[k] = g2[k - 1];
}

template <int D>
__device__ __forceinline__ void ring_8letters(int (&G)[D], int (&E)[D], v2f lane_p, uint2 Lc)
{
    ring_pairstep<D>(G, E, lane_p, Lc.x & 0xFF, (Lc.x >> 8) & 0xFF);
    ring_pairstep<D>(G, E, lane_p, (Lc.x >> 16) & 0xFF, Lc.x >> 24);
    ring_pairstep<D>(G, E, lane_p, Lc.y & 0xFF, (Lc.y >> 8) & 0xFF);
    ring_pairstep<D>(G, E, lane_p, (Lc.y >> 16) & 0xFF, Lc.y >> 24);
}

template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void k_gapless_ring(const rsk_ring *__restrict__ rings,
                                                          const uint2 *__restrict__ work,   // (ring index, first target)
                                                          const uint8_t *__restrict__ ring_letters,
                                                          const uint8_t *__restrict__ ring_laneq,
                                                          const uint32_t *__restrict__ ring_qid,
                                                          const uint8_t *__restrict__ t_mu,
                                                          const uint32_t *__restrict__ t_off,
                                                          const uint32_t *__restrict__ t_len, uint32_t nt,
                                                          const uint32_t *__restrict__ t_perm,   // self triangle: processing order
                                                          const uint32_t *__restrict__ t_claim,  // positions of each aligned block, longest first
                                                          uint32_t tb_size, int self_triangle, uint32_t win_lo, uint32_t win_hi,
                                                          uint16_t *__restrict__ out, size_t ldo, gl_hits hits,
                                                          const uint8_t *__restrict__ q_mu, const uint32_t *__restrict__ q_off,
                                                          const uint32_t *__restrict__ q_len)
{
    typedef RingGeom<D> Gm;
    constexpr int SR = Gm::SR, NQMAX = Gm::NQMAX, M = Gm::M;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *prof = (int *) smem;
    int *res = (int *) (smem + Gm::PROF_BYTES);
    signed char *mat = (signed char *) (res + NW * NQMAX);
    unsigned char *rl = (unsigned char *) (mat + 1312);
    uint32_t &next_t = *(uint32_t *) (rl + 2 * SR);   // all LDS lives in the dynamic region (keeps it 16-byte aligned)

    const int tid = threadIdx.x;
    const int nthreads = 64 * NW;
    const uint2 wk = work[blockIdx.x];
    const rsk_ring rg = rings[wk.x];
    const uint32_t t0 = wk.y;
    const uint32_t t1 = min(nt, wk.y + tb_size);
    if (tid == 0) next_t = t0;

    for (int i = tid; i < 1296; i += nthreads) mat[i] = c_mu_int[i];
    for (int i = tid; i < 2 * SR; i += nthreads) rl[i] = ring_letters[rg.letters_off + i];
    for (int i = tid; i < NW * NQMAX; i += nthreads) res[i] = 0;
    __syncthreads();
    // build the profile, 4 ring dwords (16 bytes: 4 slots of A and of B) per store
    for (int idx = tid; idx < Gm::NROWS * (SR / 4); idx += nthreads) {
        const int c = idx / (SR / 4), g = idx - c * (SR / 4);
        // LDS block g = (b128 block m = g / 64 of lane g % 64) holds ring granule M * lane + m
        const int gr = M * (g & 63) + (g >> 6);
        v4i v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned la = rl[4 * gr + k], lb = rl[SR + 4 * gr + k];
            const unsigned lo = (c == 36 || la == 0xFF) ? GL_RESET_BITS : gl_half_bits(mat[c * 36 + la]);
            const unsigned hi = (c == 36 || lb == 0xFF) ? GL_RESET_BITS : gl_half_bits(mat[c * 36 + lb]);
            v[k] = (int) (lo | (hi << 16));
        }
        *(v4i *) (prof + (size_t) c * SR + 4 * g) = v;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    // this lane's first b128 block of a profile row, as an LDS byte address, twice (see ring_pairstep)
    const unsigned lane_a = (unsigned) (uintptr_t) (const __attribute__((address_space(3))) char *) prof + lane * 16;
    const v2f lane_p = { __builtin_bit_cast(float, lane_a), __builtin_bit_cast(float, lane_a) };
    int *wres = res + wave * NQMAX;
    int lqa[M], lqb[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        lqa[m] = ring_laneq[rg.laneq_off + M * lane + m];
        lqb[m] = ring_laneq[rg.laneq_off + SR / 4 + M * lane + m];
    }
    // chain ids of the ring's members (at most 128): lane l reports members l and 64 + l
    uint32_t qid2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) qid2[h] = (uint32_t) (64 * h + lane) < rg.nq ? ring_qid[rg.qid_off + 64 * h + lane] : 0u;

    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&next_t, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= t1) break;
        t = __builtin_amdgcn_readfirstlane(t_claim[t]);                       // longest targets of the block first (LPT)
        const uint32_t tpos = t;                                              // position in the processing order
        if (self_triangle) {
            if (t < rg.min_q || t < win_lo || t >= win_hi) continue;          // positions before the ring's first member / outside the shard's window
            t = __builtin_amdgcn_readfirstlane(t_perm[t]);                    // position -> chain
        }
        const uint32_t toff = __builtin_amdgcn_readfirstlane(t_off[t]);
        const uint32_t tlen = __builtin_amdgcn_readfirstlane(t_len[t]);
        const uint2 *lp = (const uint2 *) (t_mu + toff);
        const uint32_t nfull = tlen >> 3;                                     // whole 8-letter words
        int G[D], E[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { G[k] = 0; E[k] = 0; }
        uint2 L0 = lp[0], L1 = lp[1];
        uint32_t ch = 0;
        for (; ch + 2 <= nfull; ch += 2) {
            const uint2 Lc0 = L0, Lc1 = L1;
            L0 = lp[ch + 2]; L1 = lp[ch + 3];   // prefetch (the chain set has >= 64 bytes of tail padding)
            ring_8letters<D>(G, E, lane_p, Lc0);
            ring_8letters<D>(G, E, lane_p, Lc1);
        }
        if (ch < nfull) { ring_8letters<D>(G, E, lane_p, L0); L0 = L1; }
        // the last tlen % 8 letters, two at a time (chains are padded with the pad letter, whose row resets)
        {
            unsigned long long w = ((unsigned long long) L0.y << 32) | L0.x;
            for (uint32_t np = ((tlen & 7) + 1) >> 1; np; --np) {
                ring_pairstep<D>(G, E, lane_p, (unsigned) w & 0xFF, (unsigned) (w >> 8) & 0xFF);
                w >>= 16;
            }
        }
        // per-query reduction: the granules of a lane mostly belong to one query -- one LDS atomic per run
        {
            int va[M], vb[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int b = pk_max(pk_max(E[4 * m], E[4 * m + 1]), pk_max(E[4 * m + 2], E[4 * m + 3]));
                va[m] = b & 0xFFFF;                                           // half-float bits, >= 0: ordered like integers
                vb[m] = (int) ((unsigned) b >> 16);
            }
            int ca = lqa[0], xa = va[0], cb = lqb[0], xb = vb[0];
#pragma unroll
            for (int m = 1; m < M; ++m) {
                if (lqa[m] == ca) xa = max(xa, va[m]);
                else { if (ca != 0xFF) atomicMax(&wres[ca], xa); ca = lqa[m]; xa = va[m]; }
                if (lqb[m] == cb) xb = max(xb, vb[m]);
                else { if (cb != 0xFF) atomicMax(&wres[cb], xb); cb = lqb[m]; xb = vb[m]; }
            }
            if (ca != 0xFF) atomicMax(&wres[ca], xa);
            if (cb != 0xFF) atomicMax(&wres[cb], xb);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (uint32_t kb = 0; kb < rg.nq; kb += 64) {
            const uint32_t k = kb + lane;
            const bool have = k < rg.nq;
            int v = 0;
            uint32_t qid = 0;
            if (have) { v = wres[k]; wres[k] = 0; qid = kb ? qid2[1] : qid2[0]; }
            uint32_t sc = gl_score_of_bits(v);
            // the clamp's ceiling: score these pairs again, exactly -- the wave walks the diagonals of one pair at a time
            // (x = max(0, x) + s in integers; a lane per diagonal)
            unsigned long long capped = __ballot(have && v >= GL_CAP_BITS);
            while (capped) {
                const int l = __builtin_ctzll(capped);
                capped &= capped - 1;
                const uint32_t cq = (uint32_t) __builtin_amdgcn_readlane((int) qid, l);
                const uint8_t *A = q_mu + q_off[cq];
                const uint8_t *B = t_mu + toff;
                const int LA = (int) q_len[cq], LB = (int) tlen;
                int best = 0;
                for (int d = lane; d < LA + LB - 1; d += 64) {
                    int i = d < LB ? 0 : d - LB + 1;
                    int j = d < LB ? LB - 1 - d : 0;
                    int x = 0;
                    for (; i < LA && j < LB; ++i, ++j) {
                        x = max(x, 0) + mat[A[i] * 36 + B[j]];
                        best = max(best, x);
                    }
                }
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) best = max(best, __shfl_xor(best, sft, 64));
                if (lane == l) sc = (uint32_t) min(best, 65535);
            }
            if (!have) continue;
            if (out) {
                const size_t o = self_triangle ? (size_t) min(qid, t) * ldo + max(qid, t) : (size_t) qid * ldo + t;
                out[o] = (uint16_t) sc;
            }
            // a ring sees every target from its first member's position on, so the pairs of two of its own members are
            // scored twice: the one whose target comes later in the order reports
            if (hits.rec && sc >= hits.min_score && (!self_triangle || tpos >= rg.min_q + k))
                gl_emit(hits, hits.q_base + (self_triangle ? min(qid, t) : qid), hits.t_base + (self_triangle ? max(qid, t) : t), sc);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------
// per-pair kernel: one wave per pair, lanes stride over the LA+LB-1 diagonals.  Any length; also
// returns the first strict maximum in row-major order (Besti/Bestj of SWFastGapless_Int).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_gapless_pairs(const uint8_t *__restrict__ q_mu, const uint32_t *__restrict__ q_off,
                                                        const uint32_t *__restrict__ q_len,
                                                        const uint8_t *__restrict__ t_mu, const uint32_t *__restrict__ t_off,
                                                        const uint32_t *__restrict__ t_len,
                                                        const uint32_t *__restrict__ iq, const uint32_t *__restrict__ it,
                                                        uint32_t npairs, int32_t *__restrict__ scores,
                                                        uint32_t *__restrict__ besti, uint32_t *__restrict__ bestj,
                                                        uint16_t *__restrict__ out16, size_t ldo, gl_hits hits)
{
    // one workgroup (64..1024 threads) per pair
    __shared__ signed char mat[1296];
    __shared__ unsigned long long wbest[16];
    for (int i = threadIdx.x; i < 1296; i += blockDim.x) mat[i] = c_mu_int[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t p = blockIdx.x;
    const uint32_t a = iq[p], b = it[p];
    const uint8_t *A = q_mu + q_off[a];
    const uint8_t *B = t_mu + t_off[b];
    const int LA = (int) q_len[a], LB = (int) t_len[b];
    unsigned long long best = 0;   // (score << 48) | (0xFFFFFF - i) << 24 | (0xFFFFFF - j)
    const int ndiag = LA + LB - 1;
    for (int d = threadIdx.x; d < ndiag; d += blockDim.x) {
        int i = d < LB ? 0 : d - LB + 1;
        int j = d < LB ? LB - 1 - d : 0;
        int x = 0;
        for (; i < LA && j < LB; ++i, ++j) {
            x = max(x, 0) + mat[A[i] * 36 + B[j]];
            if (x > 0) {
                const unsigned long long key = ((unsigned long long) x << 48) | ((unsigned long long) (0xFFFFFF - i) << 24) |
                                               (unsigned long long) (0xFFFFFF - j);
                if (key > best) best = key;
            }
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(best, s, 64);
        if (o > best) best = o;
    }
    if (lane == 0) wbest[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < (blockDim.x >> 6); ++w)
            if (wbest[w] > best) best = wbest[w];
        const int sc = (int) (best >> 48);
        if (scores) scores[p] = sc;
        if (besti) besti[p] = sc ? (uint32_t) (0xFFFFFF - ((best >> 24) & 0xFFFFFF)) : RSK_NO_POS;
        if (bestj) bestj[p] = sc ? (uint32_t) (0xFFFFFF - (best & 0xFFFFFF)) : RSK_NO_POS;
        if (out16) out16[(size_t) a * ldo + b] = (uint16_t) min(sc, 65535);
        if (hits.rec && (uint32_t) sc >= hits.min_score) gl_emit(hits, hits.q_base + a, hits.t_base + b, (uint32_t) min(sc, 65535));
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int D, int NW>
static int launch_ring_class(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint2 *d_work, uint32_t nwork, const uint32_t *d_claim,
                             int self_triangle, uint32_t win_lo, uint32_t win_hi, uint16_t *d_scores, size_t ldo, uint32_t tb_size, const gl_hits &hits)
{
    if (nwork == 0) return RSK_OK;
    constexpr size_t lds = ring_lds_bytes<D, NW>();
    static std::atomic<int> attr_set[64];      // one per template instance and device
    const int arc = rsk_once_per_device(attr_set, ctx->device, [&]() -> int {
        RSK_HIP(hipFuncSetAttribute((const void *) k_gapless_ring<D, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        return RSK_OK;
    });
    if (arc != RSK_OK) return arc;
    hipLaunchKernelGGL((k_gapless_ring<D, NW>), dim3(nwork), dim3(64 * NW), lds, ctx->stream, q->d_ring_tab, d_work,
                       q->d_ring_letters, q->d_ring_laneq, q->d_ring_qid, t->d_mu, t->d_off, t->d_len, t->n, q->d_ring_perm,
                       d_claim, tb_size, self_triangle, win_lo, win_hi, d_scores, ldo, hits, q->d_mu, q->d_off, q->d_len);
    RSK_HIP(hipGetLastError());
    return RSK_OK;
}

// Targets per work item.  An item costs one profile build (37 rows of the whole LDS, nothing else runs on the CU meanwhile),
// so the blocks are as large as the launch allows while it still has several items per CU: measured on the 11,211-chain
// triangle (935 rings), 128 / 256 / 512 / 1024 / 2048 targets: 45.1 / 43.6 / 42.9 / 42.6 / 43.0 ms.  (r05 tried a cost model --
// rounds of items x (build + targets per wave + half a target of end-of-item spread) -- in its place: it picked 256-target
// blocks for thin shards and was slower on 5 of 8 shards of the bench triangle.  Exchanging the roles of the two sets for a thin
// target set -- the score is symmetric -- was 2.5 x slower still: long chains make poor ring members, a 1,024-slot sub-ring
// holds one 700-residue chain and 320 idle slots, and chains beyond a ring go to the per-pair kernel.)
static uint32_t ring_target_block(size_t nrings, uint32_t nt)
{
    uint32_t tb = RING_TB_MAX;
    while (tb > 64 && nrings * ((nt + tb - 1) / tb) < 2048) tb /= 2;
    return tb;
}

// positions of every aligned block of `tb` targets ordered by decreasing target length: the waves of a workgroup claim the
// long targets first, so the block ends without one wave still walking a long chain (LPT scheduling)
static int build_claim_order(const rsk_db *db, const uint32_t *perm, std::map<uint32_t, uint32_t *> &cache, std::mutex &m, uint32_t tb,
                             const uint32_t **d_out)
{
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find(tb);
    if (it != cache.end()) { *d_out = it->second; return RSK_OK; }
    std::vector<uint32_t> claim(db->n);
    for (uint32_t i = 0; i < db->n; ++i) claim[i] = i;
    for (uint32_t b = 0; b < db->n; b += tb) {
        const uint32_t e = std::min(db->n, b + tb);
        std::stable_sort(claim.begin() + b, claim.begin() + e, [&](uint32_t x, uint32_t y) {
            return db->len[perm ? perm[x] : x] > db->len[perm ? perm[y] : y];
        });
    }
    uint32_t *d = nullptr;
    { const int rc_ = rsk_db_malloc(db, nullptr, (void **) &d, std::max<size_t>(db->n, 1) * 4); if (rc_ != RSK_OK) return rc_; }
    if (hipMemcpy(d, claim.data(), (size_t) db->n * 4, hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(d); RSK_HIP(hipErrorUnknown); }
    cache[tb] = d;
    *d_out = d;
    return RSK_OK;
}

// win_lo / win_hi (self triangle only; 0 / n = everything): the launch scores the pairs whose LATER member in the processing
// order (ring_perm) stands at a position in [win_lo, win_hi) -- one rank's share of the triangle (rsk_mu_gapless_shard_window).
// Every rank keeps the whole set and runs the same rings against its window of targets: one launch of the same shape as the
// whole triangle, instead of a rectangle plus a small triangle of its own chains (r04; measured r05 on the 8 shards of the bench
// set: the small triangles ran at 4 - 37 T cells/s against 41 - 44 for the rectangles, the last rank took 1.29 x its cells).
int rsk_launch_gapless_rings(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle,
                             uint16_t *d_scores, size_t ldo, uint32_t min_score, uint32_t q_base, uint32_t t_base, uint32_t *d_rec,
                             uint32_t capacity, uint32_t *d_count, uint32_t win_lo, uint32_t win_hi)
{
    if (!self_triangle) { win_lo = 0; win_hi = t->n; }
    win_hi = std::min(win_hi, t->n);
    win_lo = std::min(win_lo, win_hi);
    gl_hits hits = {};
    hits.rec = d_rec; hits.count = d_count; hits.cap = capacity; hits.min_score = min_score; hits.q_base = q_base; hits.t_base = t_base;
    int rc = rsk_upload_mu_tables(ctx);
    if (rc != RSK_OK) return rc;
    const uint32_t tb = ring_target_block(q->rings.size(), win_hi - win_lo);
    const uint32_t *d_claim = nullptr;
    if (self_triangle) rc = build_claim_order(q, q->h_ring_perm.data(), const_cast<rsk_db *>(q)->tri_claims, const_cast<rsk_db *>(q)->claim_mutex, tb, &d_claim);
    else rc = build_claim_order(t, nullptr, const_cast<rsk_db *>(t)->nat_claims, const_cast<rsk_db *>(t)->claim_mutex, tb, &d_claim);
    if (rc != RSK_OK) return rc;
    // work accounting (host side, O(rings))
    // targets in processing order: the ring permutation of the (same) chain set in self-triangle mode
    std::vector<uint64_t> pre_len(t->n + 1, 0), pre_slots(t->n + 1, 0);
    for (uint32_t i = 0; i < t->n; ++i) {
        const uint32_t L = t->len[self_triangle ? q->h_ring_perm[i] : i];
        pre_len[i + 1] = pre_len[i] + L;
        pre_slots[i + 1] = pre_slots[i] + (uint64_t) ((L + 1) / 2 * 2);        // letters go in pairs
    }
    uint64_t pairs = 0, cells = 0, slots = 0;
    if (self_triangle) {                                  // the pair space of the contract: every unordered pair once
        // (a pair belongs to the window its later member's position lies in: position p closes p + 1 pairs)
        for (uint32_t p = win_lo; p < win_hi; ++p) {
            pairs += (uint64_t) p + 1;
            cells += (uint64_t) t->len[q->h_ring_perm[p]] * pre_len[p + 1];
        }
    } else {
        pairs = (uint64_t) q->n * t->n;
        cells = pre_len[t->n] * [&]() { uint64_t z = 0; for (uint32_t L : q->len) z += L; return z; }();
    }
    uint32_t nD4 = 0, nD8 = 0;
    for (auto &r : q->rings) {
        (r.D == 8 ? nD4 : nD8)++;
        const uint32_t ts = std::min(win_hi, std::max(win_lo, self_triangle ? r.min_q : 0u));
        slots += 128ull * r.D * (pre_slots[win_hi] - pre_slots[ts]);
    }
    {
        uint32_t pos = (uint32_t) (q->n - q->long_q.size());
        for (uint32_t lqi : q->long_q) {
            const uint32_t ts = std::min(win_hi, std::max(win_lo, self_triangle ? pos : 0u));
            slots += (uint64_t) q->len[lqi] * (pre_len[win_hi] - pre_len[ts]);
            ++pos;
        }
    }
    ctx->gl_pairs = pairs; ctx->gl_cells = cells; ctx->gl_slots = slots;

    // work list: one workgroup per (ring, block of targets); only blocks that contain work, the expensive ones first.  Cached in
    // the query chain set per (target set, triangle flag, block size, window); the lookup, a rebuild and the launches that read the
    // entry all happen under the set's mutex (launches are asynchronous: the lock is held for microseconds), and an entry that is
    // evicted is hipFree'd, which waits for the kernels still reading it.
    rsk_db *qm = const_cast<rsk_db *>(q);
    std::lock_guard<std::mutex> work_lock(qm->claim_mutex);
    rsk_db::gl_work *we = nullptr;
    for (auto &e : qm->work_cache)
        if (e.work_for == t->uid && e.work_tri == self_triangle && e.work_tb == tb && e.win_lo == win_lo && e.win_hi == win_hi) { we = &e; break; }
    if (!we) {
        if (qm->work_cache.size() < RSK_GL_WORK_ENTRIES) { qm->work_cache.emplace_back(); we = &qm->work_cache.back(); }
        else {
            we = &qm->work_cache[0];
            for (auto &e : qm->work_cache) if (e.last_use < we->last_use) we = &e;
        }
        // the key is invalid from here until the rebuild has completed: a failure on the way cannot leave a stale hit
        we->work_for = 0;
        if (we->d_work) { (void) hipFree(we->d_work); we->d_work = nullptr; }
        if (we->d_long_iq) { (void) hipFree(we->d_long_iq); we->d_long_iq = nullptr; }
        if (we->d_long_it) { (void) hipFree(we->d_long_it); we->d_long_it = nullptr; }
        we->long_pairs = 0; we->work_count[0] = we->work_count[1] = 0;
        const uint32_t TB[2] = { tb, tb };             // targets per workgroup (the claim order is built for this block size)
        std::vector<uint2> w[2];
        std::vector<uint64_t> cost[2];
        for (uint32_t ri = 0; ri < q->rings.size(); ++ri) {
            const rsk_ring &r = q->rings[ri];
            const int c = r.D == 8 ? 0 : 1;
            const uint32_t first = std::max(win_lo, self_triangle ? r.min_q : 0u);
            const uint32_t ts = (first / TB[c]) * TB[c];
            for (uint32_t t0 = ts; t0 < win_hi; t0 += TB[c]) {
                const uint32_t lo = std::max(t0, first), hi = std::min(win_hi, t0 + TB[c]);
                if (hi <= lo) continue;
                w[c].push_back(make_uint2(ri, t0));
                cost[c].push_back(128ull * r.D * (pre_slots[hi] - pre_slots[lo]));
            }
        }
        std::vector<uint2> all;
        for (int c = 0; c < 2; ++c) {
            std::vector<uint32_t> order(w[c].size());
            for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[c][a] > cost[c][b]; });
            we->work_count[c] = (uint32_t) order.size();
            for (uint32_t k : order) all.push_back(w[c][k]);
        }
        if (!all.empty()) {
            { const int rc_ = rsk_db_malloc(qm, ctx, (void **) &we->d_work, all.size() * sizeof(uint2)); if (rc_ != RSK_OK) return rc_; }
            RSK_HIP(hipMemcpy(we->d_work, all.data(), all.size() * sizeof(uint2), hipMemcpyHostToDevice));
        }
        // queries too long for a ring: (long query, target) pairs of the per-pair kernel, part of the same entry
        if (!q->long_q.empty()) {
            std::vector<uint32_t> iq, it;
            uint32_t pos = (uint32_t) (q->n - q->long_q.size());      // long chains close the processing order
            for (uint32_t lqi : q->long_q) {
                if (self_triangle) {
                    for (uint32_t p = std::max(pos, win_lo); p < win_hi; ++p) {           // symmetric score: stored at [min][max]
                        const uint32_t tj = q->h_ring_perm[p];
                        iq.push_back(std::min(lqi, tj)); it.push_back(std::max(lqi, tj));
                    }
                } else
                    for (uint32_t j = 0; j < t->n; ++j) { iq.push_back(lqi); it.push_back(j); }
                ++pos;
            }
            if (!iq.empty()) {
                { const int rc_ = rsk_db_malloc(qm, ctx, (void **) &we->d_long_iq, iq.size() * 4); if (rc_ != RSK_OK) return rc_; }
                { const int rc_ = rsk_db_malloc(qm, ctx, (void **) &we->d_long_it, it.size() * 4); if (rc_ != RSK_OK) return rc_; }
                RSK_HIP(hipMemcpy(we->d_long_iq, iq.data(), iq.size() * 4, hipMemcpyHostToDevice));
                RSK_HIP(hipMemcpy(we->d_long_it, it.data(), it.size() * 4, hipMemcpyHostToDevice));
                we->long_pairs = (uint32_t) iq.size();
            }
        }
        we->work_tri = self_triangle;
        we->work_tb = tb;
        we->win_lo = win_lo;
        we->win_hi = win_hi;
        we->work_for = t->uid;                          // published last
    }
    we->last_use = ++qm->work_clock;
    (void) nD4; (void) nD8;

    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    // queries too long for a ring: per-pair kernel over (long q) x targets.  A few dozen one-workgroup pairs: behind the ring
    // kernels on one stream they were a 0.35 ms tail of a 42.8 ms launch -- and all of it on the LAST of N windows; on a side stream
    // (forked here, joined before ev1) they run beside them.
    bool side = we->long_pairs && !getenv("RSK_GAPLESS_NO_SIDE_STREAM");
    if (side && !ctx->aux) {
        // stream + both events into locals, published only when all three exist (a half-built set left ctx->aux
        // non-null with null events on the next call); without them the per-pair kernel simply stays on the launch stream
        hipStream_t st = nullptr;
        hipEvent_t ef = nullptr, ej = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ef, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ej, hipEventDisableTiming) == hipSuccess) {
            ctx->aux = st; ctx->ev_fork = ef; ctx->ev_join = ej;
        } else {
            (void) hipGetLastError();
            if (ej) (void) hipEventDestroy(ej);
            if (ef) (void) hipEventDestroy(ef);
            if (st) (void) hipStreamDestroy(st);
            side = false;
        }
    }
    bool forked = false;
    // whatever happens after the fork, the launch stream waits for the side stream before this function returns: the caller may
    // free or reuse d_rec / d_count / d_scores as soon as ITS stream is done
    auto join = [&]() -> hipError_t { return forked ? hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) : hipSuccess; };
    if (we->long_pairs) {
        hipStream_t ps = ctx->stream;
        if (side) {
            RSK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
            RSK_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
            ps = ctx->aux;
        }
        hipLaunchKernelGGL(k_gapless_pairs, dim3(we->long_pairs), dim3(1024), 0, ps, q->d_mu, q->d_off, q->d_len,
                           t->d_mu, t->d_off, t->d_len, we->d_long_iq, we->d_long_it, we->long_pairs, (int32_t *) nullptr, (uint32_t *) nullptr,
                           (uint32_t *) nullptr, d_scores, ldo, hits);
        hipError_t le = hipGetLastError();
        if (side) {
            // the join event is recorded even if the launch failed: the stream then simply has nothing before it
            if (hipEventRecord(ctx->ev_join, ctx->aux) == hipSuccess) forked = true;
            else { (void) hipGetLastError(); (void) hipStreamSynchronize(ctx->aux); }
        }
        if (le != hipSuccess) { (void) join(); RSK_HIP(le); }
    }
    rc = launch_ring_class<16, 16>(ctx, q, t, (const uint2 *) we->d_work + we->work_count[0], we->work_count[1], d_claim, self_triangle, win_lo, win_hi,
                                  d_scores, ldo, tb, hits);
    if (rc != RSK_OK) { (void) join(); return rc; }
    rc = launch_ring_class<8, 16>(ctx, q, t, (const uint2 *) we->d_work, we->work_count[0], d_claim, self_triangle, win_lo, win_hi, d_scores, ldo, tb, hits);
    if (rc != RSK_OK) { (void) join(); return rc; }
    RSK_HIP(join());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    return RSK_OK;
}

int rsk_launch_gapless_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *d_iq,
                             const uint32_t *d_it, size_t npairs, int32_t *d_scores, uint32_t *d_besti,
                             uint32_t *d_bestj)
{
    int rc = rsk_upload_mu_tables(ctx);
    if (rc != RSK_OK) return rc;
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_gapless_pairs, dim3((unsigned) npairs), dim3(npairs > 4096 ? 64 : 1024), 0, ctx->stream, q->d_mu, q->d_off,
                       q->d_len, t->d_mu, t->d_off, t->d_len, d_iq, d_it, (uint32_t) npairs, d_scores, d_besti, d_bestj,
                       (uint16_t *) nullptr, (size_t) 0, gl_hits{});
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    return RSK_OK;
}

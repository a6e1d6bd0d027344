// k_mu_sw.hip -- Mu-letter affine local SW score and the Mu filter (the live -search filter;
// SURVEY.md 8a rows P3/P4).
//
// Reference semantics (bit-exact on the integer score): DSSAligner::AlignMuQP_Para
// parasail_mu.cpp:120 -> parasail_sw_striped_profile_avx2_256_8 parasail.cpp:515 == plain Gotoh
//   H(i,j) = max(0, H(i-1,j-1)+s, E(i,j), F(i,j));  E(i,j+1) = max(E(i,j)-ext, H(i,j)-open);
//   F(i+1,j) = max(F(i,j)-ext, H(i,j)-open);  s = IntScoreMx_Mu (mumx_data.cpp:42), open 2, ext 1.
//   "saturated" iff best > 250 (parasail.cpp:725-737): the raw result is then 255.
//   Filter (parasail_mu.cpp:120-161, dssaligner.cpp:619-631): fwd = sat ? 777 : raw;
//   fwd < OmegaFwd -> 0; rev = raw score of the REVERSED query (255 if saturated); pass iff
//   fwd - rev >= Omega.
//
// MI355X design (integer VALU bound; no GEMM, almost no HBM traffic):
//   * one workgroup = one query chain x a batch of target chains.  The query profile (packed int16, 37 letter
//     rows, two query rows per dword) lives in LDS (replaces the striped int8 AVX2 profile of SetMuQP_Para
//     parasail_mu.cpp:163).
//   * the query is cut into strips of R = 32 rows; a strip's H/E values stay in 16 VGPRs of packed int16 (row r low,
//     row r + 16 high, see the kernel).  The g = ceil(LQ/32) strips of ONE pair sit on g CONSECUTIVE LANES that run
//     behind each other (systolic array): lane k hands the bottom-row (H, F) of its strip to lane k+1 with
//     a single v_mov_b32_dpp wave_shr:1 per column -- no scratch memory, no LDS hand-off.
//     A wave therefore works on floor(64/g) targets at once.
//   * per column a lane fetches the 32 profile scores of its strip with four ds_read_b128 (row = its target letter;
//     the strips of a pair are 16 B apart, conflict-free).
//   * two cells per VALU op (v_pk_add_i16 / v_pk_max_i16 / v_pk_sub_u16 clamp): 10 ops per cell pair.
//   * reverse pass of the filter: exact early exit once every pair of the wave has failed (musw_args.thr).
//   * persistent workgroups pull (query, target batch) items from a device-side queue that is
//     also built on the device (the reverse pass runs on data-dependent survivor lists).
#include <algorithm>
#include <vector>

#include "rsk_dev_tables.h"

typedef int v4i __attribute__((ext_vector_type(4)));


#define MUSW_R 32
#define MUSW_PADSCORE (-1000)      // pad rows / pad letters: never part of an alignment
// query classes by padded length: LDS profile 37 * LQpad * 2 B; waves per workgroup chosen so a CU holds ~16-20 waves
#define MUSW_NCLASS 3
static const uint32_t musw_class_lqpad[MUSW_NCLASS] = { 416, 1024, 2048 };      // 30.8 KB, 75.8 KB, 151.6 KB
static const uint32_t musw_class_waves[MUSW_NCLASS] = { 4, 8, 16 };
#define MUSW_MAX_LQ 2048           // 64 strips of 32 rows (one pair per wave)
// A workgroup item = MUSW_CHUNK wave-batches per wave of consecutive list positions of one query; the waves claim the
// batches from an LDS counter, so a wave that is done early (reverse pass: early exit below) takes the next batch instead
// of waiting at the item barrier for the slowest wave.
#define MUSW_CHUNK 4

struct musw_args {
    const uint8_t *q_mu; const uint32_t *q_off; const uint32_t *q_len;
    const uint8_t *t_mu; const uint32_t *t_off; const uint32_t *t_len;
    const uint2 *items;         // (query, first list position) per workgroup item
    const uint32_t *nitems;     // device-side item count
    const uint32_t *cnt;        // listed targets per query
    const uint32_t *first;      // implicit lists: first position (in perm) of query q
    const uint32_t *perm;       // implicit lists: target of position k is perm[first[q] + k] (targets by increasing length)
    int tri;                    // self triangle: pair {q,t} is computed once and stored at out[min(q,t)][max(q,t)]
    const uint32_t *list;       // explicit lists (CSR): list[rowstart[q] + k]; NULL => implicit first[q] + k
    const uint32_t *rowstart;   // CSR row starts (explicit lists)
    int reverse;                // 1: use the reversed query (m_MuRevA, parasail_mu.cpp:174-179)
    int open, ext;
    uint8_t *out;               // raw score min(best,255) with 255 = saturated
    size_t ldo;                 // dense: out[q*ldo + t];  CSR: out[rowstart[q] + k]
    uint32_t *counter;          // work counter (persistent workgroups)
    // Reverse pass of the filter only (optional, aligned with `out`): the pair fails as soon as its running best exceeds
    // thr = floor(fwd' - Omega) (fwd' - rev >= Omega <=> rev <= thr, parasail_mu.cpp:147-160, dssaligner.cpp:626), and the
    // best only grows: a wave whose pairs have all failed stops.  The score it leaves (> thr, <= the true one) still fails
    // in k_musw_survivors; pairs that pass are never cut short, so the reverse scores reported for survivors are exact.
    const int16_t *thr;
};

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

// Packed HALF floats holding integers: every value the recurrence can form below the saturation test (best > 250) is an
// integer of magnitude <= 2048 and therefore exact; beyond 2048 a half float rounds, but such a pair is already saturated
// (the reported value is 255 whatever the exact maximum).  What the float unit buys: a three-operand maximum
// (v_pk_maximum3_f16, gfx950) -- the packed int16 unit has none -- at the same issue rate (tools/ubench_valu.hip).
__device__ __forceinline__ int pk_addh(int a, int b)
{
    int r;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int pk_max3h(int a, int b, int c)
{
    int r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int pk_max3h_0(int a, int b)               // max(a, b, 0)
{
    int r;
    asm("v_pk_maximum3_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned musw_half_bits(int v)             // integer |v| <= 2048 as half-float bits
{
    return (unsigned) __builtin_bit_cast(unsigned short, (_Float16) (float) v);
}
__device__ __forceinline__ int musw_int_of_half(int bits)             // non-negative half float -> integer (rounded values stay > 250)
{
    return (int) (float) __builtin_bit_cast(_Float16, (unsigned short) bits);
}
// (lo half of a, hi half of b)
__device__ __forceinline__ int pk_lo_hi(int a, int b)
{
    int r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0xFFFF), "v"(a), "v"(b));
    return r;
}

// The target letters of a lane whose column advances by one per step, four columns at a time: the lane keeps the two dwords
// that hold letters j0 .. j0+3 (j0 = the lane's column at a step count that is a multiple of 4; j0 & 3 never changes, so
// ONE v_alignbyte_b32 cuts the four letters out), a third dword is in flight.  Letters outside [0, LB) read as the pad
// letter: a dword is patched when it enters the window, and only the few dwords that straddle an end take that path.
// (r01-r04 fetched per column under a lane-dependent `j & 3 == 0` test: ~20 VALU ops and two divergent branches per column.)
#define MUSW_PAD4 0x24242424u
struct musw_letters {
    const uint8_t *B;
    uint32_t LB, nfull, lastq;
    unsigned w0, w1, w2, sh;
    int idx;                                         // dword index of w0 (negative while the lane waits for its first column)
    __device__ __forceinline__ unsigned patched(unsigned raw, int i) const
    {
        if ((unsigned) i < nfull) return raw;        // every letter of the dword is inside the chain
        const int nb = (int) LB - 4 * i;             // letters of the chain from this dword's first byte on
        if (i < 0 || nb <= 0) return MUSW_PAD4;
        const unsigned m = (1u << (8 * nb)) - 1u;    // 1 <= nb <= 3
        return (raw & m) | (MUSW_PAD4 & ~m);
    }
    __device__ __forceinline__ unsigned load(int i) const
    {
        const int ic = min(max(i, 0), (int) lastq);  // stays inside the chain's padded bytes (+ the set's tail slack)
        return *(const unsigned *) (B + 4 * ic);
    }
    __device__ __forceinline__ void start(const uint8_t *b, uint32_t lb, int j0)
    {
        B = b; LB = lb; nfull = lb >> 2; lastq = (lb + 3) >> 2;
        sh = (unsigned) j0 & 3u;
        idx = j0 >> 2;
        w0 = patched(load(idx), idx);
        w1 = patched(load(idx + 1), idx + 1);
        w2 = load(idx + 2);
    }
    // letters of the next four columns (byte k = column j0 + k), then the window moves on by one dword
    __device__ __forceinline__ unsigned next4()
    {
        const unsigned l4 = __builtin_amdgcn_alignbyte(w1, w0, sh);
        w0 = w1;
        w1 = patched(w2, idx + 2);
        ++idx;
        w2 = load(idx + 2);
        return l4;
    }
};
// lane l <- (lane l-1's x) & m, 0 from outside the wave: the hand-down and its mask in one VOP2 with a DPP source
__device__ __forceinline__ int dpp_wave_shr1_and(int x, int m)
{
    int r;
    asm("v_and_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(x), "v"(m));
    return r;
}

// Two cells per VALU op: the 32 rows of a strip are 16 registers of packed half floats, row r in the low
// half and row r + 16 in the high half.  The high half runs ONE COLUMN BEHIND the low half (the same
// systolic skew that separates neighbouring lanes, applied inside the lane), so the vertical F chain
// row 15 -> row 16 crosses from the low half of one step to the high half of the next; the next lane
// is two columns behind.  E and F are kept floored at 0 (a negative E or F is equivalent to 0 in H = max(0, ...)), which
// removes the explicit max(., 0) from H: the floor is the third operand of the maximum that updates E / F.
// Per 2 cells: bfi (merge the two profile rows), add, max3 (H), add, add, max3 (E), add, max3 (F), and half a max3 for the
// running best (two rows at a time) = 8.5 packed ops (r01-r02, packed int16: 10).
// GMAX = strips of the class's longest query: the profile is laid out for GMAX strips whatever the query's own count, so that
// the four b128 blocks of a letter row sit at immediate offsets from one address (see k_mu_sw2).
template <int GMAX>
__global__ __launch_bounds__(1024) void k_mu_sw(musw_args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // query profile (packed half floats), dword (c, k, st, w) = rows 32*st + 4*k + w (low half) and + 16 (high half) against
    // letter c: P[((c*4 + k)*GMAX + st)*4 + w]; a lane's four b128 reads per letter row are conflict-free across the strips of a pair
    int *prof = (int *) smem;
    signed char *mat = (signed char *) (prof + (size_t) 37 * GMAX * 16);
    uint32_t *wg_item = (uint32_t *) (mat + 1312);           // [0] item, [1] batch counter of the item
    int *pbest = (int *) (wg_item + 4);                      // per wave 64 running pair maxima (early exit of the reverse pass)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nwaves = blockDim.x >> 6;

    for (int i = tid; i < 1296; i += blockDim.x) mat[i] = (signed char) c_mu_int[i];
    uint32_t cur_q = 0xFFFFFFFFu;
    uint32_t LQ = 0, g = 1;
    const uint32_t nitems = *a.nitems;

    for (;;) {
        __syncthreads();
        if (tid == 0) { wg_item[0] = atomicAdd(a.counter, 1u); wg_item[1] = 0; }
        __syncthreads();
        const uint32_t item = wg_item[0];
        if (item >= nitems) break;
        const uint2 it = a.items[item];
        const uint32_t q = it.x;
        if (q != cur_q) {
            cur_q = q;
            LQ = a.q_len[q];
            g = (LQ + MUSW_R - 1) / MUSW_R;
            const uint8_t *Q = a.q_mu + a.q_off[q];
            const uint32_t per_c = g * 16;
            for (uint32_t idx = tid; idx < 37 * per_c; idx += blockDim.x) {
                const uint32_t c = idx / per_c, rem = idx - c * per_c;
                const uint32_t k = rem / (g * 4), rem2 = rem - k * (g * 4);
                const uint32_t sst = rem2 >> 2, w = rem2 & 3;
                const uint32_t ilo = sst * MUSW_R + 4 * k + w, ihi = ilo + 16;
                int vlo = MUSW_PADSCORE, vhi = MUSW_PADSCORE;
                if (c < 36 && ilo < LQ) vlo = mat[c * 36 + Q[a.reverse ? (LQ - 1 - ilo) : ilo]];
                if (c < 36 && ihi < LQ) vhi = mat[c * 36 + Q[a.reverse ? (LQ - 1 - ihi) : ihi]];
                prof[((c * 4 + k) * GMAX + sst) * 4 + w] = (int) (musw_half_bits(vlo) | (musw_half_bits(vhi) << 16));
            }
            __syncthreads();
        }
        const uint32_t ppw = 64 / g;                 // pairs per wave (g <= 64 guaranteed by the host)
        const uint32_t cnt = a.cnt[q];
        const uint32_t chunk_end = min(cnt, it.y + ppw * nwaves * MUSW_CHUNK);
        const uint32_t pr = lane / g, st = lane - pr * g;     // pair slot and strip of this lane
        for (;;) {                                   // batches of ppw consecutive list positions, claimed by the waves
        uint32_t bq = 0;
        if (lane == 0) bq = atomicAdd(&wg_item[1], 1u);
        bq = (uint32_t) __builtin_amdgcn_readfirstlane((int) bq);
        const uint32_t k0 = it.y + bq * ppw;         // first list position of this batch
        if (k0 >= chunk_end) break;
        const bool active = (pr < ppw) && (k0 + pr < chunk_end);
        uint32_t t = 0, LB = 0;
        const uint8_t *B = a.t_mu;
        if (active) {
            const uint32_t k = k0 + pr;
            t = a.list ? a.list[a.rowstart[q] + k] : a.perm[a.first[q] + k];
            LB = a.t_len[t];
            B = a.t_mu + a.t_off[t];
        }
        // steps this wave must run: low half of strip st is at column step - 2*st, the high half one behind
        uint32_t ncol = active ? (LB + 2 * st + 1) : 0;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) ncol = max(ncol, (uint32_t) __shfl_xor((int) ncol, s, 64));
        if (ncol == 0) continue;

        // H lives in two register sets that swap roles every step (no copies for the diagonal hand-down)
        int HA[16], HB[16], E[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { HA[r] = 0; HB[r] = 0; E[r] = 0; }
        int best = 0;
        int bot_h = 0, bot_f = 0;      // (row 15 | row 31) H and outgoing F of the previous step
        int diag_in = 0;               // H above the top rows at the previous column
        const int nopen2 = (int) (musw_half_bits(-a.open) * 0x10001u), next2 = (int) (musw_half_bits(-a.ext) * 0x10001u);
        // this lane's block of a letter row's first quarter, as an LDS byte address
        const unsigned lane_prof = (unsigned) (uintptr_t) (const __attribute__((address_space(3))) char *) prof + st * 16;
        constexpr uint32_t kstride = GMAX * 16, RS = GMAX * 64;       // bytes between k blocks / letter rows
        const int top_mask = st == 0 ? 0 : -1;                // the very first rows have no strip above: H = F = 0
        musw_letters tl;
        tl.start(B, LB, -2 * (int) st);                       // column of the low half; the high half is one column behind
        unsigned rowh = 36u * RS + lane_prof;                 // letter row of the previous column

        auto step = [&](const int (&Hin)[16], int (&Hout)[16], unsigned c) {
            // rows above: low half <- high half of the previous lane (its previous step), high half <- own low half
            const int xh = dpp_wave_shr1_and(bot_h, top_mask), xf = dpp_wave_shr1_and(bot_f, top_mask);
            const int up_h = (int) __builtin_amdgcn_alignbit((unsigned) bot_h, (unsigned) xh, 16);
            const int up_f = (int) __builtin_amdgcn_alignbit((unsigned) bot_f, (unsigned) xf, 16);
            const unsigned rowl = __umul24(c, RS) + lane_prof;
            v4i Pl[4], Ph[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Pl[k] = *(const v4i __attribute__((address_space(3))) *) (uintptr_t) (rowl + k * kstride);
                Ph[k] = *(const v4i __attribute__((address_space(3))) *) (uintptr_t) (rowh + k * kstride);
            }
            rowh = rowl;
            int diag = diag_in;
            int F = up_f;
            diag_in = up_h;
            int hprev = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int S = pk_lo_hi(Pl[r >> 2][r & 3], Ph[r >> 2][r & 3]);
                const int h = pk_max3h(pk_addh(diag, S), E[r], F);
                diag = Hin[r];
                Hout[r] = h;
                if (r & 1) best = pk_max3h(best, hprev, h);
                hprev = h;
                const int ho = pk_addh(h, nopen2);
                E[r] = pk_max3h_0(pk_addh(E[r], next2), ho);
                F = pk_max3h_0(pk_addh(F, next2), ho);
            }
            bot_h = Hout[15];
            bot_f = F;
        };
        int thr_l = -1;                               // inactive lanes count as decided
        if (a.thr) {
            if (active) {                               // in the order of half-float bits (non-negative values order like integers)
                const int tv = a.thr[a.rowstart[q] + k0 + pr];
                thr_l = tv < 0 ? -1 : (int) musw_half_bits(min(tv, 2047));
            }
            pbest[wave * 64 + lane] = 0;
        }
        // a step count that is not a multiple of 4 is rounded up: the extra steps only see pad letters / finished columns
        for (uint32_t col = 0; col < ncol; col += 4) {
            const unsigned l4 = tl.next4();
            step(HA, HB, l4 & 0xFF);
            step(HB, HA, (l4 >> 8) & 0xFF);
            step(HA, HB, (l4 >> 16) & 0xFF);
            step(HB, HA, l4 >> 24);
            if (a.thr && (col & 4)) {                // every 8 columns: has every pair of this wave failed already?
                const int mine = max(best & 0xFFFF, (int) ((unsigned) best >> 16));
                __hip_atomic_fetch_max(&pbest[wave * 64 + pr], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const int pb = pbest[wave * 64 + pr];
                if (__ballot(pb <= thr_l) == 0ull) break;
            }
        }
        int red = max(best & 0xFFFF, (int) ((unsigned) best >> 16));
        best = red;
        for (uint32_t d = 1; d < g; ++d) {
            const int o = __shfl(best, (int) ((lane + d) & 63), 64);
            if (st + d < g) red = max(red, o);
        }
        if (active && st == 0) {
            red = musw_int_of_half(red);
            const uint8_t v = (uint8_t) (red > 250 ? 255 : red);
            if (a.list) a.out[a.rowstart[q] + k0 + pr] = v;
            else if (a.tri) a.out[(size_t) min(q, t) * a.ldo + max(q, t)] = v;
            else a.out[(size_t) q * a.ldo + t] = v;
        }
        }   // batches
    }
}

// ---------------------------------------------------------------------------------------------
// k_mu_sw2 -- the dense forward pass (every query against one implicit target range) with TWO QUERIES per packed register.
// k_mu_sw packs rows r and r + 16 of ONE query into a dword; the vertical F chain between the halves forces the high half
// to run one column behind, i.e. on the PREVIOUS target letter: two profile rows per step (8 ds_read_b128), a v_bfi per cell
// pair to merge them and two alignbit / and per step for the crossing.  When the target range is the same for many queries
// -- the forward pass of a search, 300 ms of a 460 ms all-vs-all -- two queries of neighbouring length can share a register
// instead: low half = row r of query A, high half = row r of query B, both against the SAME target column.  One profile
// row per step (4 ds_read_b128), no merge, no crossing: 7.5 packed ops per cell pair instead of 8.5, 16-row strips (half the
// padding of a chain's last strip), lanes one column apart.  The profile of a query pair has the bytes of a one-query
// profile of twice the length, so the launch geometry of k_mu_sw is reused with the pair's "virtual length" 2 * max(LA, LB)
// (work items, classes, LDS sizes: run_mu_sw_querypairs); queries of more than 1024 residues stay with k_mu_sw.
// In the self triangle the pair (A, B) of ranks (r, r + 1) takes the targets of rank >= r: B meets A once more as a target
// (the same score lands in the same cell), everything else is the triangle.
// ---------------------------------------------------------------------------------------------
#define MUSW2_RMAX 16
#define MUSW_NOQ 0xFFFFFFFFu
// Geometry of a query pair (r05): g strips of R rows, R = 4 .. 16 -- a template parameter of the batch loop, the workgroup
// branches on its pair's R.  r02-r04 fixed R = 16 and g = ceil(L / 16): a pair of 174 residues took 11 lanes, a wave 5 targets
// on 55 of its 64 lanes.  Now the host picks per query pair, between g0 = ceil(L / 16) and the next power of two, the (g, R =
// ceil(L / g)) of least modelled cost (musw2_geometry: idle lanes 64 mod g, pad rows g R - L, the g - 1 columns of systolic
// skew, ~10 instructions per step beside 7.5 per row): 16 lanes x 11 rows for that pair.  The kernel is issue-bound (PMC: a
// VALU instruction in 99 % of the cycles at 2.37 GHz), so every idle lane-row is time.
// GMAX = strips of the class's longest query pair: the profile is laid out for GMAX strips whatever the pair's own count, so
// that the b128 blocks of a letter row sit at IMMEDIATE offsets from one address (one v_mad per column instead of
// a 64-bit multiply and three adds).
template <int GMAX, int R>
__device__ __forceinline__ void musw2_batches(const musw_args &a, const uint2 it, const uint32_t p, const uint32_t qa, const uint32_t qb, const uint32_t g,
                                              const int *prof, uint32_t *wg_item)
{
    constexpr int NK = (R + 3) / 4;                  // b128 blocks of a strip's letter row
    const int lane = threadIdx.x & 63;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t ppw = 64 / g;                 // targets per wave, each against both queries
    const uint32_t cnt = a.cnt[p];
    const uint32_t chunk_end = min(cnt, it.y + ppw * nwaves * MUSW_CHUNK);
    const uint32_t pr = lane / g, st = lane - pr * g;
    for (;;) {
        uint32_t bq = 0;
        if (lane == 0) bq = atomicAdd(&wg_item[1], 1u);
        bq = (uint32_t) __builtin_amdgcn_readfirstlane((int) bq);
        const uint32_t k0 = it.y + bq * ppw;
        if (k0 >= chunk_end) break;
        const bool active = (pr < ppw) && (k0 + pr < chunk_end);
        uint32_t t = 0, LB = 0;
        const uint8_t *B = a.t_mu;
        if (active) {
            t = a.perm[a.first[p] + k0 + pr];
            LB = a.t_len[t];
            B = a.t_mu + a.t_off[t];
        }
        uint32_t ncol = active ? (LB + st) : 0;      // strip st is st columns behind strip 0
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) ncol = max(ncol, (uint32_t) __shfl_xor((int) ncol, s, 64));
        if (ncol == 0) continue;
        int HA[R], HB[R], E[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { HA[r] = 0; HB[r] = 0; E[r] = 0; }
        int best = 0;
        int bot_h = 0, bot_f = 0;      // bottom row of this strip at its previous step: H and the outgoing F
        int diag_in = 0;               // H above the top row at the previous column
        const int nopen2 = (int) (musw_half_bits(-a.open) * 0x10001u), next2 = (int) (musw_half_bits(-a.ext) * 0x10001u);
        // this lane's block of a letter row's first quarter, as an LDS byte address
        const unsigned lane_prof = (unsigned) (uintptr_t) (const __attribute__((address_space(3))) char *) prof + st * 16;
        constexpr uint32_t kstride = GMAX * 16, RS = GMAX * 64;
        const int top_mask = st == 0 ? 0 : -1;       // the first strip has nothing above it: H = F = 0
        musw_letters tl;
        tl.start(B, LB, -(int) st);                  // strip st is st columns behind strip 0
        auto step = [&](const int (&Hin)[R], int (&Hout)[R], unsigned c) {
            const int up_h = dpp_wave_shr1_and(bot_h, top_mask), up_f = dpp_wave_shr1_and(bot_f, top_mask);
            const unsigned row = __umul24(c, RS) + lane_prof;
            v4i P[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) P[k] = *(const v4i __attribute__((address_space(3))) *) (uintptr_t) (row + k * kstride);
            int diag = diag_in;
            int F = up_f;
            diag_in = up_h;
            int hprev = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int h = pk_max3h(pk_addh(diag, P[r >> 2][r & 3]), E[r], F);
                diag = Hin[r];
                Hout[r] = h;
                if (r & 1) best = pk_max3h(best, hprev, h);
                else if (r == R - 1) best = pk_max3h(best, h, h);          // odd R: the last row on its own
                hprev = h;
                const int ho = pk_addh(h, nopen2);
                E[r] = pk_max3h_0(pk_addh(E[r], next2), ho);
                F = pk_max3h_0(pk_addh(F, next2), ho);
            }
            bot_h = Hout[R - 1];
            bot_f = F;
        };
        // a step count that is not a multiple of 4 is rounded up: the extra steps only see pad letters / finished columns
        for (uint32_t col = 0; col < ncol; col += 4) {
            const unsigned l4 = tl.next4();
            step(HA, HB, l4 & 0xFF);
            step(HB, HA, (l4 >> 8) & 0xFF);
            step(HA, HB, (l4 >> 16) & 0xFF);
            step(HB, HA, l4 >> 24);
        }
        int ra = best & 0xFFFF, rb = (int) ((unsigned) best >> 16);      // half-float bits >= 0: ordered like integers
        for (uint32_t d = 1; d < g; ++d) {
            const int o = __shfl(best, (int) ((lane + d) & 63), 64);
            if (st + d < g) { ra = max(ra, o & 0xFFFF); rb = max(rb, (int) ((unsigned) o >> 16)); }
        }
        if (active && st == 0) {
            const int sa = musw_int_of_half(ra), sb = musw_int_of_half(rb);
            const uint8_t va = (uint8_t) (sa > 250 ? 255 : sa), vb = (uint8_t) (sb > 250 ? 255 : sb);
            if (a.tri) {
                a.out[(size_t) min(qa, t) * a.ldo + max(qa, t)] = va;
                if (qb != MUSW_NOQ) a.out[(size_t) min(qb, t) * a.ldo + max(qb, t)] = vb;
            } else {
                a.out[(size_t) qa * a.ldo + t] = va;
                if (qb != MUSW_NOQ) a.out[(size_t) qb * a.ldo + t] = vb;
            }
        }
    }   // batches
}

// qgeom[p] = g | R << 8 of query pair p (musw2_geometry on the host)
template <int GMAX>
__global__ __launch_bounds__(1024) void k_mu_sw2(musw_args a, const uint2 *__restrict__ qpairs, const uint32_t *__restrict__ qgeom)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // dword (c, k, st, w) = row R*st + 4*k + w of query A (low half) and of query B (high half) against letter c:
    // P[((c*4 + k)*GMAX + st)*4 + w]; a lane's b128 reads per letter row are conflict-free across the strips of a group
    int *prof = (int *) smem;
    signed char *mat = (signed char *) (prof + (size_t) 37 * GMAX * 16);
    uint32_t *wg_item = (uint32_t *) (mat + 1312);
    const int tid = threadIdx.x;
    for (int i = tid; i < 1296; i += blockDim.x) mat[i] = (signed char) c_mu_int[i];
    uint32_t cur_p = 0xFFFFFFFFu, qa = 0, qb = MUSW_NOQ, g = 1, R = MUSW2_RMAX;
    const uint32_t nitems = *a.nitems;
    for (;;) {
        __syncthreads();
        if (tid == 0) { wg_item[0] = atomicAdd(a.counter, 1u); wg_item[1] = 0; }
        __syncthreads();
        const uint32_t item = wg_item[0];
        if (item >= nitems) break;
        const uint2 it = a.items[item];
        const uint32_t p = it.x;
        if (p != cur_p) {
            cur_p = p;
            qa = qpairs[p].x; qb = qpairs[p].y;
            g = qgeom[p] & 0xFFu; R = qgeom[p] >> 8;
            const uint32_t LA = a.q_len[qa], LBq = qb != MUSW_NOQ ? a.q_len[qb] : 0u;
            const uint8_t *QA = a.q_mu + a.q_off[qa], *QB = a.q_mu + (qb != MUSW_NOQ ? a.q_off[qb] : 0u);
            const uint32_t nk = (R + 3) / 4, per_c = g * nk * 4;
            for (uint32_t idx = tid; idx < 37 * per_c; idx += blockDim.x) {
                const uint32_t c = idx / per_c, rem = idx - c * per_c;
                const uint32_t k = rem / (g * 4), rem2 = rem - k * (g * 4);
                const uint32_t sst = rem2 >> 2, w = rem2 & 3;
                const uint32_t i = sst * R + 4 * k + w;
                const bool row = 4 * k + w < R;                      // rows beyond the strip's R: pad
                int va = MUSW_PADSCORE, vb = MUSW_PADSCORE;
                if (c < 36 && row && i < LA) va = mat[c * 36 + QA[a.reverse ? (LA - 1 - i) : i]];
                if (c < 36 && row && i < LBq) vb = mat[c * 36 + QB[a.reverse ? (LBq - 1 - i) : i]];
                prof[((c * 4 + k) * GMAX + sst) * 4 + w] = (int) (musw_half_bits(va) | (musw_half_bits(vb) << 16));
            }
            __syncthreads();
        }
        switch (R) {
        case 4: musw2_batches<GMAX, 4>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 5: musw2_batches<GMAX, 5>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 6: musw2_batches<GMAX, 6>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 7: musw2_batches<GMAX, 7>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 8: musw2_batches<GMAX, 8>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 9: musw2_batches<GMAX, 9>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 10: musw2_batches<GMAX, 10>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 11: musw2_batches<GMAX, 11>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 12: musw2_batches<GMAX, 12>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 13: musw2_batches<GMAX, 13>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 14: musw2_batches<GMAX, 14>(a, it, p, qa, qb, g, prof, wg_item); break;
        case 15: musw2_batches<GMAX, 15>(a, it, p, qa, qb, g, prof, wg_item); break;
        default: musw2_batches<GMAX, 16>(a, it, p, qa, qb, g, prof, wg_item); break;
        }
    }
}

// Any length: one thread per pair, DP rows in global scratch (rare: both chains > 2048).
__global__ void k_mu_sw_slow(musw_args a, const uint2 *pairs, const uint32_t *pair_k, uint32_t npairs, int *scratch,
                             size_t scratch_stride)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const uint32_t q = pairs[p].x, t = pairs[p].y;
    const uint8_t *A = a.q_mu + a.q_off[q];
    const uint8_t *B = a.t_mu + a.t_off[t];
    const int LA = (int) a.q_len[q], LB = (int) a.t_len[t];
    int *Hc = scratch + (size_t) p * scratch_stride;
    int *Ec = Hc + LA;
    for (int i = 0; i < LA; ++i) { Hc[i] = 0; Ec[i] = 0; }
    int best = 0;
    for (int j = 0; j < LB; ++j) {
        int diag = 0, F = 0;
        const int b = B[j];
        for (int i = 0; i < LA; ++i) {
            const int ai = a.reverse ? A[LA - 1 - i] : A[i];
            int h = diag + c_mu_int[b * 36 + ai];
            h = max(h, 0); h = max(h, Ec[i]); h = max(h, F);
            diag = Hc[i];
            Hc[i] = h;
            best = max(best, h);
            const int ho = h - a.open;
            Ec[i] = max(Ec[i] - a.ext, ho);
            F = max(F - a.ext, ho);
        }
    }
    const uint8_t v = (uint8_t) (best > 250 ? 255 : best);
    if (a.list) a.out[a.rowstart[q] + pair_k[p]] = v;
    else if (a.tri) a.out[(size_t) min(q, t) * a.ldo + max(q, t)] = v;
    else a.out[(size_t) q * a.ldo + t] = v;
}

// ---------------------------------------------------------------------------------------------
// device-side queue construction
// ---------------------------------------------------------------------------------------------
// class of a query by its padded length: 0: <= 416, 1: <= 1024, 2: <= 2048 (LDS int16 profile), 4: slow path
__device__ __forceinline__ int musw_class(uint32_t LQ)
{
    const uint32_t lp = (LQ + MUSW_R - 1) / MUSW_R * MUSW_R;
    if (lp <= 416) return 0;
    if (lp <= 1024) return 1;
    if (LQ <= MUSW_MAX_LQ) return 2;
    return 4;
}
__device__ __forceinline__ uint32_t musw_waves(int cls) { return cls == 0 ? 4u : cls == 1 ? 8u : 16u; }

// implicit lists: first[q], cnt[q]
// Self triangle: the Mu matrix and the gap costs are symmetric, so score(q,t) == score(t,q) (forward
// and reversed-query alike); each unordered pair is computed once with the SHORTER chain (lower rank)
// in the query role (fewer strips per pair, shorter systolic ramp), i.e. query q takes the targets of
// rank >= rank[q].
// [wlo, whi): a WINDOW of target positions (length ranks) of the triangle -- one shard of a self search (r06: every rank keeps the
// whole set and takes the pairs whose longer member stands in its window; the whole triangle is [0, nt)).
__global__ void k_musw_setup_implicit(uint32_t nq, uint32_t nt, int self_triangle, const uint32_t *rank, uint32_t *first, uint32_t *cnt,
                                      uint32_t wlo, uint32_t whi)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t f = self_triangle ? max(rank[q], wlo) : 0u;
    first[q] = f;
    cnt[q] = self_triangle ? (whi > f ? whi - f : 0u) : nt;
}

// Single workgroup: for class `cls` compute per-query item counts, exclusive scan, write item_start[q] and *nitems.
__global__ __launch_bounds__(1024) void k_musw_scan_items(const uint32_t *q_len, const uint32_t *cnt, uint32_t nq, int cls,
                                                          uint32_t *item_start, uint32_t *nitems)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nq; base += 1024) {
        const uint32_t q = base + tid;
        uint32_t v = 0;
        if (q < nq && musw_class(q_len[q]) == cls && cnt[q]) {
            const uint32_t LQ = q_len[q];
            if (cls == 4) v = cnt[q];                          // slow path: one item per pair
            else {
                const uint32_t g = (LQ + MUSW_R - 1) / MUSW_R;
                const uint32_t per_wg = (64 / g) * musw_waves(cls) * MUSW_CHUNK;
                v = (cnt[q] + per_wg - 1) / per_wg;
            }
        }
        // inclusive scan within wave
        uint32_t x = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = (uint32_t) __shfl_up((int) x, s, 64);
            if (lane >= s) x += o;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t excl = carry + woff + x - v;
        if (q < nq) item_start[q] = excl;
        __syncthreads();
        if (tid == 1023) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) { item_start[nq] = carry; *nitems = carry; }
}

__global__ void k_musw_fill_items(const uint32_t *q_len, const uint32_t *cnt, const uint32_t *first, const uint32_t *perm, const uint32_t *list,
                                  const uint32_t *rowstart, uint32_t nq, int cls, const uint32_t *item_start, uint2 *items,
                                  uint32_t *pair_k)
{
    const uint32_t q = blockIdx.x;
    if (q >= nq || musw_class(q_len[q]) != cls) return;
    const uint32_t n = item_start[q + 1] - item_start[q];
    const uint32_t LQ = q_len[q];
    if (cls == 4) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
            const uint32_t t = list ? list[rowstart[q] + k] : perm[first[q] + k];
            items[item_start[q] + k] = make_uint2(q, t);
            pair_k[item_start[q] + k] = k;
        }
        return;
    }
    const uint32_t g = (LQ + MUSW_R - 1) / MUSW_R;
    const uint32_t per_wg = (64 / g) * musw_waves(cls) * MUSW_CHUNK;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) items[item_start[q] + k] = make_uint2(q, k * per_wg);
}

// ---------------------------------------------------------------------------------------------
// filter plumbing: candidate lists (CSR) and the final survivor list
// ---------------------------------------------------------------------------------------------
// one wave per query row: count (pass 0) or write (pass 1) targets with fwd' >= omega_fwd
__global__ __launch_bounds__(256) void k_musw_candidates(const uint8_t *fwd, size_t ldo, const uint32_t *first, const uint32_t *cnt,
                                                         const uint32_t *perm, int tri,
                                                         uint32_t nq, float omega_fwd, int pass, uint32_t *ccnt,
                                                         const uint32_t *rowstart, uint32_t *list, int16_t *thr, float omega)
{
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // wave-uniform
    if (q >= nq) return;
    const int lane = threadIdx.x & 63;
    const uint32_t f0 = first[q], n = cnt[q];
    uint32_t run = 0;
    for (uint32_t k = 0; k < n; k += 64) {
        const uint32_t kk = k + lane;
        bool c = false;
        uint32_t t = 0;
        float f = 0.0f;
        if (kk < n) {
            t = perm[f0 + kk];
            const int raw = tri ? fwd[(size_t) min(q, t) * ldo + max(q, t)] : fwd[(size_t) q * ldo + t];
            f = raw == 255 ? 777.0f : (float) raw;                 // parasail_mu.cpp:135-139
            c = !(f < omega_fwd);                                    // :141-146
        }
        const unsigned long long m = __ballot(c);
        if (pass == 1 && c) {
            const uint32_t pos = rowstart[q] + run + __popcll(m & ((1ull << lane) - 1ull));
            list[pos] = t;
            if (thr) thr[pos] = (int16_t) fminf(fmaxf(floorf(f - omega), -1.0f), 32767.0f);   // rev <= thr <=> fwd' - rev >= Omega
        }
        run += (uint32_t) __popcll(m);
    }
    if (pass == 0 && lane == 0) ccnt[q] = run;
}

// exclusive scan of cnt[0..n) -> out[0..n], single workgroup
__global__ __launch_bounds__(1024) void k_exclusive_scan_u32(const uint32_t *in, uint32_t n, uint32_t *out)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? in[i] : 0;
        uint32_t x = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = (uint32_t) __shfl_up((int) x, s, 64);
            if (lane >= s) x += o;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t excl = carry + woff + x - v;
        if (i < n) out[i] = excl;
        __syncthreads();
        if (tid == 1023) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

// one wave per query row: survivors = candidates with fwd' - rev >= omega  (dssaligner.cpp:626-629)
__global__ __launch_bounds__(256) void k_musw_survivors(const uint8_t *fwd, size_t ldo, int tri, const uint32_t *ccnt, const uint32_t *rowstart,
                                                        const uint32_t *list, const uint8_t *rev, uint32_t nq, float omega,
                                                        uint32_t *pairs_q, uint32_t *pairs_t, int32_t *pairs_fwd, int32_t *pairs_rev,
                                                        uint32_t capacity, uint32_t *npairs)
{
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // wave-uniform
    if (q >= nq) return;
    const int lane = threadIdx.x & 63;
    const uint32_t n = ccnt[q], rs = rowstart[q];
    for (uint32_t k = 0; k < n; k += 64) {
        const uint32_t kk = k + lane;
        bool c = false;
        uint32_t t = 0;
        int f = 0, r = 0;
        if (kk < n) {
            t = list[rs + kk];
            const int raw = tri ? fwd[(size_t) min(q, t) * ldo + max(q, t)] : fwd[(size_t) q * ldo + t];
            f = raw == 255 ? 777 : raw;
            r = rev[rs + kk];                   // 255 when saturated (parasail_mu.cpp:152 read before the fix-up)
            const float sc = (float) f - (float) r;
            c = !(sc < omega);                  // MuFilter: MuScore < MCS -> reject (dssaligner.cpp:626-628)
        }
        const unsigned long long m = __ballot(c);
        const uint32_t tot = (uint32_t) __popcll(m);
        uint32_t base = 0;
        if (lane == 0 && tot) base = atomicAdd(npairs, tot);
        base = (uint32_t) __shfl((int) base, 0, 64);
        if (c) {
            const uint32_t pos = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
            if (pos < capacity) {
                pairs_q[pos] = tri ? min(q, t) : q; pairs_t[pos] = tri ? max(q, t) : t;
                if (pairs_fwd) pairs_fwd[pos] = f;
                if (pairs_rev) pairs_rev[pos] = r;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------

struct musw_ws {               // per-call device workspace
    uint32_t *first = nullptr, *cnt = nullptr, *item_start = nullptr, *nitems = nullptr, *counter = nullptr, *pair_k = nullptr;
    uint2 *items = nullptr;
    int *slow_scratch = nullptr;
    rsk_ctx *ctx = nullptr;
    std::vector<void *> all;
    explicit musw_ws(rsk_ctx *c) : ctx(c) {}
    ~musw_ws() { for (void *p : all) rsk_pool_free(ctx, p); }
    template <class T> int alloc(T **p, size_t n)
    {
        int r = rsk_pool_alloc(ctx, (void **) p, std::max<size_t>(n, 1) * sizeof(T));
        if (r != RSK_OK) return r;
        all.push_back(*p);
        return RSK_OK;
    }
};

// Runs the SW kernel for every class over lists described by (cnt, first | list+rowstart).
static int run_mu_sw_lists(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, musw_args base, uint32_t max_items_hint,
                           musw_ws &ws)
{
    const uint32_t nq = q->n;
    int rc;
    if ((rc = ws.alloc(&ws.item_start, (size_t) nq + 1)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&ws.nitems, 1)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&ws.counter, 1)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&ws.items, (size_t) max_items_hint)) != RSK_OK) return rc;
    uint32_t maxLQ = 0;
    bool has_class[5] = { false, false, false, false, false };
    for (uint32_t i = 0; i < nq; ++i) {
        const uint32_t L = q->len[i], lp = (L + MUSW_R - 1) / MUSW_R * MUSW_R;
        maxLQ = std::max(maxLQ, L);
        const int c = lp <= 416 ? 0 : lp <= 1024 ? 1 : L <= MUSW_MAX_LQ ? 2 : 4;
        has_class[c] = true;
    }
    for (int cls = 0; cls < 5; ++cls) {
        if (!has_class[cls]) continue;
        RSK_HIP(hipMemsetAsync(ws.counter, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_musw_scan_items, dim3(1), dim3(1024), 0, ctx->stream, q->d_len, base.cnt, nq, cls, ws.item_start,
                           ws.nitems);
        if (cls == 4 && !ws.pair_k) { if ((rc = ws.alloc(&ws.pair_k, (size_t) max_items_hint)) != RSK_OK) return rc; }
        hipLaunchKernelGGL(k_musw_fill_items, dim3(nq), dim3(64), 0, ctx->stream, q->d_len, base.cnt, base.first, base.perm, base.list,
                           base.rowstart, nq, cls, ws.item_start, ws.items, ws.pair_k);
        musw_args a = base;
        a.items = ws.items;
        a.nitems = ws.nitems;
        a.counter = ws.counter;
        if (cls < MUSW_NCLASS) {
            const uint32_t gmax = musw_class_lqpad[cls] / MUSW_R;
            const uint32_t waves = musw_class_waves[cls];
            const size_t lds = (size_t) 37 * gmax * 64 + 1312 + 16 + (size_t) waves * 256;
            static std::atomic<int> attr_set[64];
            const int arc = rsk_once_per_device(attr_set, ctx->device, [&]() -> int {
                RSK_HIP(hipFuncSetAttribute((const void *) k_mu_sw<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                RSK_HIP(hipFuncSetAttribute((const void *) k_mu_sw<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                RSK_HIP(hipFuncSetAttribute((const void *) k_mu_sw<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                return RSK_OK;
            });
            if (arc != RSK_OK) return arc;
            const int wg_per_cu = std::max(1, std::min<int>(20 / (int) waves, (int) (163840 / lds)));
            static_assert(MUSW_NCLASS == 3, "one k_mu_sw instance per class");
            const dim3 grid(ctx->num_cus * wg_per_cu), block(64 * waves);
            if (gmax == 13) hipLaunchKernelGGL(k_mu_sw<13>, grid, block, lds, ctx->stream, a);
            else if (gmax == 32) hipLaunchKernelGGL(k_mu_sw<32>, grid, block, lds, ctx->stream, a);
            else hipLaunchKernelGGL(k_mu_sw<64>, grid, block, lds, ctx->stream, a);
        } else {
            // slow path: needs the item count on the host to size the launch
            uint32_t n = 0;
            RSK_HIP(hipMemcpyAsync(&n, ws.nitems, 4, hipMemcpyDeviceToHost, ctx->stream));
            RSK_HIP(hipStreamSynchronize(ctx->stream));
            if (n) {
                const size_t stride = 2 * (size_t) maxLQ;
                if ((rc = ws.alloc(&ws.slow_scratch, stride * n)) != RSK_OK) return rc;
                hipLaunchKernelGGL(k_mu_sw_slow, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, a, (const uint2 *) ws.items,
                                   (const uint32_t *) ws.pair_k, n, ws.slow_scratch, stride);
            }
        }
        RSK_HIP(hipGetLastError());
    }
    return RSK_OK;
}

static uint32_t item_exact_implicit(const rsk_db *q, uint32_t nt, int self_triangle);

// The dense forward pass through k_mu_sw2: queries of <= 1024 residues are paired in order of length (virtual query p =
// (A, B), virtual length 2 * max(LA, LB): the work-item kernels and the class geometry of k_mu_sw apply unchanged), the rest
// -- and nothing else -- goes through k_mu_sw.  base.cnt / base.first describe the implicit lists per REAL query.
// (g, R) of a query pair of L residues for k_mu_sw2: among g = ceil(L / 16) .. 64 strips of R = ceil(L / g) rows each, the
// geometry of least modelled time per useful cell against a 174-residue target (the mean of SCOP40): step cost (8 + 7.5 R)
// instructions for R rows, times the idle shares -- lanes 64 mod g, pad rows g R - L, the g - 1 columns of skew.
// RSK_MUSW2_FIXED_R=1: the r04 geometry (R = 16, g = g0).
static void musw2_geometry(uint32_t L, uint32_t *g_out, uint32_t *R_out)
{
    static const bool fixed = getenv("RSK_MUSW2_FIXED_R") != nullptr;
    static const bool pow2_only = getenv("RSK_MUSW2_POW2") != nullptr;      // r05a: g0 or the next power of two only
    const uint32_t g0 = (L + MUSW2_RMAX - 1) / MUSW2_RMAX;
    uint32_t p2 = 1;
    while (p2 < g0) p2 *= 2;
    double best = 0;
    uint32_t bg = g0, bR = MUSW2_RMAX;
    for (uint32_t g = g0; g <= 64; ++g) {
        if (fixed && g != g0) break;
        if (pow2_only && g != g0 && g != p2) continue;
        const uint32_t R = fixed ? (uint32_t) MUSW2_RMAX : std::max<uint32_t>(4, (L + g - 1) / g);
        const double lanes = (double) ((64 / g) * g) / 64.0, rows = (double) L / ((double) g * R), cols = 174.0 / (174.0 + g - 1);
        const double cost = (8.0 + 7.5 * R) / R / (lanes * rows * cols);
        if (g == g0 || cost < best) { best = cost; bg = g; bR = R; }
        if (R == 4) break;
    }
    *g_out = bg; *R_out = bR;
}

static int run_mu_sw_querypairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, musw_args base, int self_triangle, musw_ws &ws,
                                uint32_t wlo = 0, uint32_t whi = 0xFFFFFFFFu)
{
    const uint32_t nq = q->n, nt = t->n;
    whi = std::min(whi, nt);
    // targets of (real or paired) query A in triangle mode: the positions max(rank[A], wlo) .. whi - 1 of the length order
    auto tri_first = [&](uint32_t A) { return std::max(t->h_len_rank[A], wlo); };
    auto tri_cnt = [&](uint32_t A) { const uint32_t f = tri_first(A); return whi > f ? whi - f : 0u; };
    std::vector<uint32_t> order(nq);
    for (uint32_t i = 0; i < nq; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return q->len[x] < q->len[y]; });   // == rsk_build_len_perm's order
    std::vector<uint2> qp;
    std::vector<uint32_t> vlen, vcnt, vfirst, vgeom, cnt_old(nq, 0);
    bool any_old = false;
    uint64_t nitems_ub = 16;
    bool has_class[3] = { false, false, false };
    for (uint32_t k = 0; k < nq;) {
        const uint32_t A = order[k];
        if (q->len[A] > 1024) { cnt_old[A] = self_triangle ? tri_cnt(A) : nt; any_old = any_old || cnt_old[A]; ++k; continue; }
        uint32_t Bq = MUSW_NOQ;
        if (k + 1 < nq && q->len[order[k + 1]] <= 1024) Bq = order[k + 1];
        const uint32_t L = std::max(q->len[A], Bq != MUSW_NOQ ? q->len[Bq] : 0u);
        uint32_t g, R;
        musw2_geometry(std::max(L, 1u), &g, &R);
        // the work-item kernels derive a query's strips from its length as ceil(length / 32): the pair's "virtual length" 32 g
        // gives them the g chosen here (classes, waves per workgroup and LDS sizes follow from it as for k_mu_sw)
        const uint32_t vl = MUSW_R * g;
        const uint32_t c = self_triangle ? tri_cnt(A) : nt;
        if (c == 0) { k += Bq != MUSW_NOQ ? 2 : 1; continue; }            // (a window that ends before this pair's first target)
        qp.push_back(make_uint2(A, Bq));
        vlen.push_back(vl); vcnt.push_back(c); vfirst.push_back(self_triangle ? tri_first(A) : 0u);
        vgeom.push_back(g | (R << 8));
        const uint32_t lp = g * MUSW_R;
        const int cls = lp <= 416 ? 0 : lp <= 1024 ? 1 : 2;
        has_class[cls] = true;
        const uint32_t per_wg = (64 / g) * musw_class_waves[cls] * MUSW_CHUNK;
        nitems_ub += (c + per_wg - 1) / per_wg;
        k += Bq != MUSW_NOQ ? 2 : 1;
    }
    int rc;
    const uint32_t npv = (uint32_t) qp.size();
    if (npv) {
        uint2 *d_qp = nullptr;
        uint32_t *d_vlen = nullptr, *d_vcnt = nullptr, *d_vfirst = nullptr, *d_vgeom = nullptr;
        if ((rc = ws.alloc(&d_qp, npv)) != RSK_OK || (rc = ws.alloc(&d_vlen, npv)) != RSK_OK || (rc = ws.alloc(&d_vcnt, npv)) != RSK_OK ||
            (rc = ws.alloc(&d_vfirst, npv)) != RSK_OK || (rc = ws.alloc(&d_vgeom, npv)) != RSK_OK)
            return rc;
        if ((rc = ws.alloc(&ws.item_start, (size_t) nq + 1)) != RSK_OK || (rc = ws.alloc(&ws.nitems, 1)) != RSK_OK ||
            (rc = ws.alloc(&ws.counter, 1)) != RSK_OK || (rc = ws.alloc(&ws.items, (size_t) std::min<uint64_t>(nitems_ub, 0xFFFFFFF0ull))) != RSK_OK)
            return rc;
        // (synchronous copies of a few hundred KB: the vectors die with this call)
        RSK_HIP(hipStreamSynchronize(ctx->stream));
        RSK_HIP(hipMemcpy(d_qp, qp.data(), (size_t) npv * sizeof(uint2), hipMemcpyHostToDevice));
        RSK_HIP(hipMemcpy(d_vlen, vlen.data(), (size_t) npv * 4, hipMemcpyHostToDevice));
        RSK_HIP(hipMemcpy(d_vcnt, vcnt.data(), (size_t) npv * 4, hipMemcpyHostToDevice));
        RSK_HIP(hipMemcpy(d_vfirst, vfirst.data(), (size_t) npv * 4, hipMemcpyHostToDevice));
        RSK_HIP(hipMemcpy(d_vgeom, vgeom.data(), (size_t) npv * 4, hipMemcpyHostToDevice));
        for (int cls = 0; cls < MUSW_NCLASS; ++cls) {
            if (!has_class[cls]) continue;
            RSK_HIP(hipMemsetAsync(ws.counter, 0, 4, ctx->stream));
            hipLaunchKernelGGL(k_musw_scan_items, dim3(1), dim3(1024), 0, ctx->stream, d_vlen, d_vcnt, npv, cls, ws.item_start, ws.nitems);
            hipLaunchKernelGGL(k_musw_fill_items, dim3(npv), dim3(64), 0, ctx->stream, d_vlen, d_vcnt, d_vfirst, base.perm, (const uint32_t *) nullptr,
                               (const uint32_t *) nullptr, npv, cls, ws.item_start, ws.items, (uint32_t *) nullptr);
            musw_args a = base;
            a.items = ws.items; a.nitems = ws.nitems; a.counter = ws.counter;
            a.cnt = d_vcnt; a.first = d_vfirst;
            const uint32_t gmax = musw_class_lqpad[cls] / MUSW_R;
            const uint32_t waves = musw_class_waves[cls];
            const size_t lds = (size_t) 37 * gmax * 64 + 1312 + 16;
            static std::atomic<int> attr_set[64];
            const int arc = rsk_once_per_device(attr_set, ctx->device, [&]() -> int {
                RSK_HIP(hipFuncSetAttribute((const void *) k_mu_sw2<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                RSK_HIP(hipFuncSetAttribute((const void *) k_mu_sw2<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                RSK_HIP(hipFuncSetAttribute((const void *) k_mu_sw2<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
                return RSK_OK;
            });
            if (arc != RSK_OK) return arc;
            const int wg_per_cu = std::max(1, std::min<int>(20 / (int) waves, (int) (163840 / lds)));
            static_assert(MUSW_NCLASS == 3, "one k_mu_sw2 instance per class");
            const dim3 grid(ctx->num_cus * wg_per_cu), block(64 * waves);
            if (gmax == 13) hipLaunchKernelGGL(k_mu_sw2<13>, grid, block, lds, ctx->stream, a, (const uint2 *) d_qp, (const uint32_t *) d_vgeom);
            else if (gmax == 32) hipLaunchKernelGGL(k_mu_sw2<32>, grid, block, lds, ctx->stream, a, (const uint2 *) d_qp, (const uint32_t *) d_vgeom);
            else hipLaunchKernelGGL(k_mu_sw2<64>, grid, block, lds, ctx->stream, a, (const uint2 *) d_qp, (const uint32_t *) d_vgeom);
            RSK_HIP(hipGetLastError());
        }
    }
    if (any_old) {
        uint32_t *d_cnt_old = nullptr;
        if ((rc = ws.alloc(&d_cnt_old, nq)) != RSK_OK) return rc;
        RSK_HIP(hipStreamSynchronize(ctx->stream));
        RSK_HIP(hipMemcpy(d_cnt_old, cnt_old.data(), (size_t) nq * 4, hipMemcpyHostToDevice));
        musw_args b = base;
        b.cnt = d_cnt_old;
        // (item bound of the whole query set: an upper bound for the few long queries left here)
        if ((rc = run_mu_sw_lists(ctx, q, t, b, item_exact_implicit(q, nt, self_triangle), ws)) != RSK_OK) return rc;
    }
    return RSK_OK;
}

static uint32_t item_exact_implicit(const rsk_db *q, uint32_t nt, int self_triangle)
{
    uint64_t n = 16;
    for (uint32_t i = 0; i < q->n; ++i) {
        const uint32_t L = q->len[i];
        const uint64_t cnt = self_triangle ? nt - q->h_len_rank[i] : nt;
        if (L > MUSW_MAX_LQ) { n += cnt; continue; }
        const uint32_t g = (L + MUSW_R - 1) / MUSW_R, lp = g * MUSW_R;
        const uint32_t per_wg = (64 / g) * (lp <= 416 ? 4u : lp <= 1024 ? 8u : 16u) * MUSW_CHUNK;
        n += (cnt + per_wg - 1) / per_wg;
    }
    return (uint32_t) std::min<uint64_t>(n, 0xFFFFFFF0ull);
}

static uint32_t item_upper_bound(const rsk_db *q, uint64_t total_listed)
{
    // every item covers >= 1 pair (slow path) ... typical >= 4; bound by pairs/1 is too large, so:
    // items <= sum_q ceil(cnt_q / per_wg_q) <= total/ per_wg_min + nq, per_wg_min = 4 (g = 64) .. slow path cnt.
    uint64_t ub = total_listed / 4 + q->n + 16;
    bool any_slow = false;
    for (uint32_t L : q->len) if (L > MUSW_MAX_LQ) { any_slow = true; break; }
    if (any_slow) ub = total_listed + q->n + 16;
    return (uint32_t) std::min<uint64_t>(ub, 0xFFFFFFF0ull);
}

int rsk_build_len_perm(rsk_db *db)
{
    if (db->d_len_perm) return RSK_OK;
    std::vector<uint32_t> perm(db->n), rank(db->n);
    for (uint32_t i = 0; i < db->n; ++i) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return db->len[x] < db->len[y]; });
    for (uint32_t k = 0; k < db->n; ++k) rank[perm[k]] = k;
    { const int rc_ = rsk_db_malloc(db, nullptr, (void **) &db->d_len_perm, std::max<size_t>(db->n, 1) * 4); if (rc_ != RSK_OK) return rc_; }
    { const int rc_ = rsk_db_malloc(db, nullptr, (void **) &db->d_len_rank, std::max<size_t>(db->n, 1) * 4); if (rc_ != RSK_OK) return rc_; }
    RSK_HIP(hipMemcpy(db->d_len_perm, perm.data(), (size_t) db->n * 4, hipMemcpyHostToDevice));
    RSK_HIP(hipMemcpy(db->d_len_rank, rank.data(), (size_t) db->n * 4, hipMemcpyHostToDevice));
    db->hbm_bytes += (uint64_t) db->n * 8;
    db->h_len_rank.swap(rank);
    return RSK_OK;
}

extern "C" int rsk_mu_sw_matrix_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, int reverse_query,
                                    int gap_open, int gap_ext, uint8_t *d_scores, size_t ldo)
{
    if (!ctx || !q || !t || !d_scores) { rsk_set_error("rsk_mu_sw_matrix_dev: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_sw_matrix_dev: chain set has no Mu letters"); return RSK_E_INVALID; }
    if (ldo < t->n) { rsk_set_error("rsk_mu_sw_matrix_dev: ldo < number of targets"); return RSK_E_INVALID; }
    if (self_triangle && q != t) { rsk_set_error("rsk_mu_sw_matrix_dev: self_triangle needs q == t"); return RSK_E_INVALID; }
    if (gap_open < 0 || gap_ext < 0) { rsk_set_error("rsk_mu_sw_matrix_dev: gap costs must be >= 0"); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = rsk_upload_mu_tables(ctx);
    if (rc != RSK_OK) return rc;
    musw_ws ws(ctx);
    if ((rc = ws.alloc(&ws.first, q->n)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&ws.cnt, q->n)) != RSK_OK) return rc;
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if ((rc = rsk_build_len_perm(const_cast<rsk_db *>(t))) != RSK_OK) return rc;
    hipLaunchKernelGGL(k_musw_setup_implicit, dim3((q->n + 255) / 256), dim3(256), 0, ctx->stream, q->n, t->n, self_triangle,
                       t->d_len_rank, ws.first, ws.cnt, 0u, t->n);
    musw_args a = {};
    a.q_mu = q->d_mu; a.q_off = q->d_off; a.q_len = q->d_len;
    a.t_mu = t->d_mu; a.t_off = t->d_off; a.t_len = t->d_len;
    a.cnt = ws.cnt; a.first = ws.first; a.list = nullptr; a.rowstart = nullptr;
    a.perm = t->d_len_perm; a.tri = self_triangle ? 1 : 0;
    a.reverse = reverse_query ? 1 : 0; a.open = gap_open; a.ext = gap_ext;
    a.out = d_scores; a.ldo = ldo;
    const bool pairs2 = !(getenv("RSK_MUSW_QUERY_PAIRS") && atoi(getenv("RSK_MUSW_QUERY_PAIRS")) == 0);
    rc = pairs2 ? run_mu_sw_querypairs(ctx, q, t, a, self_triangle, ws) : run_mu_sw_lists(ctx, q, t, a, item_exact_implicit(q, t->n, self_triangle), ws);
    if (rc != RSK_OK) return rc;
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));   // workspace is freed on return
    return RSK_OK;
}

static int mu_filter_impl(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, int gap_open, int gap_ext,
                          float omega, float omega_fwd, uint8_t *d_fwd, size_t ldo, uint32_t *d_pairs_q,
                          uint32_t *d_pairs_t, int32_t *d_pairs_fwd, int32_t *d_pairs_rev, size_t capacity,
                          uint32_t *d_npairs, uint32_t wlo, uint32_t whi);

extern "C" int rsk_mu_filter_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, int gap_open, int gap_ext,
                                 float omega, float omega_fwd, uint8_t *d_fwd, size_t ldo, uint32_t *d_pairs_q,
                                 uint32_t *d_pairs_t, int32_t *d_pairs_fwd, int32_t *d_pairs_rev, size_t capacity,
                                 uint32_t *d_npairs)
{
    return mu_filter_impl(ctx, q, t, self_triangle, gap_open, gap_ext, omega, omega_fwd, d_fwd, ldo, d_pairs_q, d_pairs_t, d_pairs_fwd, d_pairs_rev,
                          capacity, d_npairs, 0u, 0xFFFFFFFFu);
}

// One shard of the self-search triangle as a WINDOW [rank_lo, rank_hi) of positions of the set's length order (stable sort by
// length: rsk_len_rank): the pairs {a, b} whose LONGER member (the later one of equal lengths) stands in the window, each once,
// survivors as (min index, max index).  The windows 0 .. N - 1 of rsk_self_window tile the triangle; one launch per window of
// the same shape as the whole triangle's (the same query pairs against fewer targets).
extern "C" int rsk_mu_filter_window_dev(rsk_ctx *ctx, const rsk_db *db, uint32_t rank_lo, uint32_t rank_hi, int gap_open, int gap_ext,
                                        float omega, float omega_fwd, uint8_t *d_fwd, size_t ldo, uint32_t *d_pairs_q,
                                        uint32_t *d_pairs_t, int32_t *d_pairs_fwd, int32_t *d_pairs_rev, size_t capacity,
                                        uint32_t *d_npairs)
{
    if (db && (rank_lo > rank_hi || rank_hi > db->n)) { rsk_set_error("rsk_mu_filter_window_dev: window [%u, %u) outside the set's %u positions", rank_lo, rank_hi, db->n); return RSK_E_INVALID; }
    return mu_filter_impl(ctx, db, db, 1, gap_open, gap_ext, omega, omega_fwd, d_fwd, ldo, d_pairs_q, d_pairs_t, d_pairs_fwd, d_pairs_rev,
                          capacity, d_npairs, rank_lo, rank_hi);
}

// rank[i] = position of chain i in the set's length order (the order the triangle mode of the Mu filter walks its targets in)
extern "C" int rsk_len_rank(const rsk_db *db, uint32_t *rank)
{
    if (!db || !rank) { rsk_set_error("rsk_len_rank: NULL argument"); return RSK_E_INVALID; }
    const int rc = rsk_build_len_perm(const_cast<rsk_db *>(db));
    if (rc != RSK_OK) return rc;
    std::copy(db->h_len_rank.begin(), db->h_len_rank.end(), rank);
    return RSK_OK;
}

static int mu_filter_impl(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, int gap_open, int gap_ext,
                          float omega, float omega_fwd, uint8_t *d_fwd, size_t ldo, uint32_t *d_pairs_q,
                          uint32_t *d_pairs_t, int32_t *d_pairs_fwd, int32_t *d_pairs_rev, size_t capacity,
                          uint32_t *d_npairs, uint32_t wlo, uint32_t whi)
{
    if (!ctx || !q || !t || !d_fwd || !d_pairs_q || !d_pairs_t || !d_npairs) { rsk_set_error("rsk_mu_filter_dev: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_filter_dev: chain set has no Mu letters"); return RSK_E_INVALID; }
    if (ldo < t->n) { rsk_set_error("rsk_mu_filter_dev: ldo < number of targets"); return RSK_E_INVALID; }
    if (self_triangle && q != t) { rsk_set_error("rsk_mu_filter_dev: self_triangle needs q == t"); return RSK_E_INVALID; }
    if (capacity > 0xFFFFFFFFull) capacity = 0xFFFFFFFFull;
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = rsk_upload_mu_tables(ctx);
    if (rc != RSK_OK) return rc;
    const uint32_t nq = q->n;
    musw_ws ws(ctx), ws2(ctx);
    if ((rc = ws.alloc(&ws.first, nq)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&ws.cnt, nq)) != RSK_OK) return rc;
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_npairs, 0, 4, ctx->stream));
    if ((rc = rsk_build_len_perm(const_cast<rsk_db *>(t))) != RSK_OK) return rc;
    whi = std::min(whi, t->n);
    wlo = std::min(wlo, whi);
    hipLaunchKernelGGL(k_musw_setup_implicit, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, nq, t->n, self_triangle, t->d_len_rank,
                       ws.first, ws.cnt, wlo, whi);
    musw_args a = {};
    a.q_mu = q->d_mu; a.q_off = q->d_off; a.q_len = q->d_len;
    a.t_mu = t->d_mu; a.t_off = t->d_off; a.t_len = t->d_len;
    a.cnt = ws.cnt; a.first = ws.first;
    a.perm = t->d_len_perm; a.tri = self_triangle ? 1 : 0;
    a.reverse = 0; a.open = gap_open; a.ext = gap_ext;
    a.out = d_fwd; a.ldo = ldo;
    // (position p of the length order closes p + 1 pairs of the triangle)
    const uint64_t total = self_triangle ? ((uint64_t) whi * (whi + 1) - (uint64_t) wlo * (wlo + 1)) / 2 : (uint64_t) nq * t->n;
    const bool pairs2 = !(getenv("RSK_MUSW_QUERY_PAIRS") && atoi(getenv("RSK_MUSW_QUERY_PAIRS")) == 0) || (self_triangle && (wlo != 0 || whi != t->n));
    if ((rc = pairs2 ? run_mu_sw_querypairs(ctx, q, t, a, self_triangle, ws, wlo, whi)
                     : run_mu_sw_lists(ctx, q, t, a, item_exact_implicit(q, t->n, self_triangle), ws)) != RSK_OK)
        return rc;
    // candidates with fwd' >= OmegaFwd -> CSR lists
    uint32_t *ccnt = nullptr, *rowstart = nullptr, *list = nullptr;
    uint8_t *rev = nullptr;
    if ((rc = ws.alloc(&ccnt, nq)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&rowstart, (size_t) nq + 1)) != RSK_OK) return rc;
    hipLaunchKernelGGL(k_musw_candidates, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, d_fwd, ldo, ws.first, ws.cnt, a.perm, a.tri, nq, omega_fwd,
                       0, ccnt, (const uint32_t *) nullptr, (uint32_t *) nullptr, (int16_t *) nullptr, omega);
    hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(1024), 0, ctx->stream, ccnt, nq, rowstart);
    uint32_t ncand = 0;
    RSK_HIP(hipMemcpyAsync(&ncand, rowstart + nq, 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->mf_pairs = total;
    ctx->mf_candidates = ncand;
    if (getenv("RSK_TRACE")) fprintf(stderr, "[rsk_mu_filter_dev] %llu pairs, %u candidates for the reverse pass (%.1f %%)\n", (unsigned long long) total, ncand, 100.0 * ncand / (double) std::max<uint64_t>(total, 1));
    if (ncand) {
        int16_t *thr = nullptr;
        const bool early = !(getenv("RSK_MUSW_EARLY_EXIT") && atoi(getenv("RSK_MUSW_EARLY_EXIT")) == 0);
        if ((rc = ws.alloc(&list, ncand)) != RSK_OK) return rc;
        if ((rc = ws.alloc(&rev, ncand)) != RSK_OK) return rc;
        if (early && (rc = ws.alloc(&thr, ncand)) != RSK_OK) return rc;
        hipLaunchKernelGGL(k_musw_candidates, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, d_fwd, ldo, ws.first, ws.cnt, a.perm, a.tri, nq,
                           omega_fwd, 1, ccnt, rowstart, list, thr, omega);
        musw_args b = a;
        b.cnt = ccnt; b.first = nullptr; b.list = list; b.rowstart = rowstart;
        b.reverse = 1; b.out = rev; b.thr = thr;
        if ((rc = run_mu_sw_lists(ctx, q, t, b, item_upper_bound(q, ncand), ws2)) != RSK_OK) return rc;
        hipLaunchKernelGGL(k_musw_survivors, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, d_fwd, ldo, a.tri, ccnt, rowstart, list, rev, nq,
                           omega, d_pairs_q, d_pairs_t, d_pairs_fwd, d_pairs_rev, (uint32_t) capacity, d_npairs);
    }
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));   // workspace is freed on return
    return RSK_OK;
}

// Pair-list form of the filter (DSSAligner::AlignMuParaBags parasail_mu.cpp:183 as PostMuFilter calls it per
// (query, target) candidate, chainbag.cpp:68-74): host arrays in, per-pair verdict out.
// ---- pair lists (prefilter candidates, self-rev) ------------------------------------------------------------------
// The SW kernel wants a CSR by query with each query's partners by increasing length (the pairs of a wave then end
// together).  r04: the lists never exist on the host -- the caller's two index columns go up once, keys (query << 32 |
// partner length) are radix-sorted on the device with the list position as payload (a stable sort: equal keys keep the
// caller's order), row starts are binary searches over the sorted keys, scores are scattered back to the caller's order
// by the same payload, and the reverse pass's candidate list, thresholds and the final verdicts are kernels too.  (Three
// rounds of host counting sorts / per-query std::sorts / gathers over 16.5 M candidates were 0.65 s of a 0.93 s call.)
__global__ void k_mfp_keys(const uint32_t *iq, const uint32_t *it, const uint32_t *sel, size_t n, const uint32_t *t_len,
                           unsigned long long *keys, uint32_t *vals)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = sel ? sel[k] : (uint32_t) k;
    keys[k] = ((unsigned long long) iq[p] << 32) | t_len[it[p]];
    vals[k] = (uint32_t) k;
}

__global__ void k_mfp_rows(const unsigned long long *keys, size_t n, uint32_t nq, uint32_t *rowstart, uint32_t *cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nq) return;
    auto lb = [&](uint32_t q) -> uint32_t {
        size_t lo = 0, hi = n;
        const unsigned long long key = (unsigned long long) q << 32;
        while (lo < hi) { const size_t mid = lo + ((hi - lo) >> 1); if (keys[mid] < key) lo = mid + 1; else hi = mid; }
        return (uint32_t) lo;
    };
    const uint32_t a = lb(i);
    rowstart[i] = a;
    if (i < nq) cnt[i] = (i + 1 == nq ? (uint32_t) n : lb(i + 1)) - a;
}

__global__ void k_mfp_gather(const uint32_t *it, const uint32_t *sel, const uint32_t *ord, size_t n, uint32_t *list, const int16_t *thr,
                             int16_t *thr_sorted)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t o = ord[k];
    list[k] = it[sel ? sel[o] : o];
    if (thr) thr_sorted[k] = thr[o];
}

__global__ void k_mfp_scatter(const uint8_t *sorted, const uint32_t *ord, size_t n, uint8_t *raw)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) raw[ord[k]] = sorted[k];
}

// forward scores -> the list of pairs that need the reverse pass (fwd >= OmegaFwd, parasail_mu.cpp:141-146) and their
// early-exit thresholds floor(fwd' - Omega)
__global__ void k_mfp_candidates(const uint8_t *rawf, size_t n, float omega, float omega_fwd, uint32_t *cand, int16_t *thr, uint32_t *ncand)
{
    const size_t p = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    bool is = false;
    float f = 0.0f;
    if (p < n) {
        f = rawf[p] == 255 ? 777.0f : (float) rawf[p];                     // parasail_mu.cpp:135-139
        is = !(f < omega_fwd);
    }
    const unsigned long long m = __ballot(is);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(ncand, (uint32_t) __popcll(m));
    base = (uint32_t) __shfl((int) base, __builtin_ctzll(m), 64);
    if (is) {
        const uint32_t k = base + (uint32_t) __popcll(m & ((1ull << lane) - 1));
        cand[k] = (uint32_t) p;
        thr[k] = (int16_t) fminf(fmaxf(floorf(f - omega), -1.0f), 32767.0f);
    }
}

// reverse scores of the candidates (aligned with cand) -> by pair position
__global__ void k_mfp_rev_by_pos(const uint8_t *rawr, const uint32_t *cand, size_t ncand, uint8_t *rev_pos, uint8_t *is_cand)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ncand) { rev_pos[cand[k]] = rawr[k]; is_cand[cand[k]] = 1; }
}

__global__ void k_mfp_verdict(const uint8_t *rawf, const uint8_t *rev_pos, const uint8_t *is_cand, size_t n, float omega, uint8_t *pass,
                              int32_t *fwd, int32_t *rev)
{
    const size_t p = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int f = rawf[p] == 255 ? 777 : rawf[p];
    const int r = is_cand[p] ? rev_pos[p] : 0;
    const float score = is_cand[p] ? (float) f - (float) r : 0.0f;          // fwd < OmegaFwd: AlignMuQP_Para returns 0
    pass[p] = !(score < omega);                                             // chainbag.cpp:71-73
    if (fwd) fwd[p] = f;
    if (rev) rev[p] = r;
}

// One pass of the SW kernel over the pairs sel[0..n) (NULL = all n pairs) of the device columns d_iq / d_it; d_raw[k] =
// raw score of the k-th selected pair.
static int musw_run_pairlist_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *d_iq, const uint32_t *d_it, const uint32_t *d_sel,
                                 size_t n, int reverse, int gap_open, int gap_ext, uint8_t *d_raw, const int16_t *d_thr)
{
    if (n == 0) return RSK_OK;
    musw_ws ws(ctx);
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_vals = nullptr, *d_ord = nullptr, *d_cnt = nullptr, *d_rowstart = nullptr, *d_list = nullptr;
    uint8_t *d_out = nullptr;
    int16_t *d_thr_sorted = nullptr;
    int rc;
    if ((rc = ws.alloc(&d_keys, n)) || (rc = ws.alloc(&d_keys2, n)) || (rc = ws.alloc(&d_vals, n)) || (rc = ws.alloc(&d_ord, n)) ||
        (rc = ws.alloc(&d_cnt, (size_t) q->n)) || (rc = ws.alloc(&d_rowstart, (size_t) q->n + 1)) || (rc = ws.alloc(&d_list, n)) ||
        (rc = ws.alloc(&d_out, n)))
        return rc;
    if (d_thr && (rc = ws.alloc(&d_thr_sorted, n)) != RSK_OK) return rc;
    const unsigned nb = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(k_mfp_keys, dim3(nb), dim3(256), 0, ctx->stream, d_iq, d_it, d_sel, n, t->d_len, d_keys, d_vals);
    int qbits = 1;
    while (qbits < 32 && ((uint64_t) q->n >> qbits) != 0) ++qbits;
    if ((rc = rsk_sort_pairs_u64_u32(ctx, d_keys, d_keys2, d_vals, d_ord, n, 32 + qbits)) != RSK_OK) return rc;      // stable (k_pairs_sort.hip)
    hipLaunchKernelGGL(k_mfp_rows, dim3(q->n / 256 + 1), dim3(256), 0, ctx->stream, d_keys2, n, q->n, d_rowstart, d_cnt);
    hipLaunchKernelGGL(k_mfp_gather, dim3(nb), dim3(256), 0, ctx->stream, d_it, d_sel, d_ord, n, d_list, d_thr, d_thr_sorted);
    RSK_HIP(hipGetLastError());
    musw_args a = {};
    a.q_mu = q->d_mu; a.q_off = q->d_off; a.q_len = q->d_len;
    a.t_mu = t->d_mu; a.t_off = t->d_off; a.t_len = t->d_len;
    a.cnt = d_cnt; a.first = nullptr; a.perm = nullptr; a.tri = 0; a.list = d_list; a.rowstart = d_rowstart;
    a.reverse = reverse; a.open = gap_open; a.ext = gap_ext;
    a.out = d_out; a.ldo = 0;
    a.thr = d_thr ? d_thr_sorted : nullptr;
    if ((rc = run_mu_sw_lists(ctx, q, t, a, item_upper_bound(q, n), ws)) != RSK_OK) return rc;
    hipLaunchKernelGGL(k_mfp_scatter, dim3(nb), dim3(256), 0, ctx->stream, d_out, d_ord, n, d_raw);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipStreamSynchronize(ctx->stream));                              // the workspace goes back to the pool on return
    return RSK_OK;
}

extern "C" int rsk_mu_filter_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it, size_t npairs,
                                   int gap_open, int gap_ext, float omega, float omega_fwd, uint8_t *pass, int32_t *fwd, int32_t *rev)
{
    if (!ctx || !q || !t || (npairs && (!iq || !it || !pass))) { rsk_set_error("rsk_mu_filter_pairs: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_filter_pairs: chain set has no Mu letters"); return RSK_E_INVALID; }
    if (gap_open < 0 || gap_ext < 0) { rsk_set_error("rsk_mu_filter_pairs: gap costs must be >= 0"); return RSK_E_INVALID; }
    if (npairs > 0xFFFFFFF0ull) { rsk_set_error("rsk_mu_filter_pairs: too many pairs in one call"); return RSK_E_RANGE; }
    {
        std::atomic<size_t> bad{(size_t) -1};
        rsk_parallel_for(npairs, 1 << 18, [&](size_t lo, size_t hi) {
            for (size_t p = lo; p < hi; ++p)
                if (iq[p] >= q->n || it[p] >= t->n) { size_t cur = bad.load(); while (p < cur && !bad.compare_exchange_weak(cur, p)) {} break; }
        });
        if (bad.load() != (size_t) -1) { rsk_set_error("rsk_mu_filter_pairs: pair %zu out of range", bad.load()); return RSK_E_INVALID; }
    }
    ctx->mf_pairs = npairs;
    ctx->mf_candidates = 0;
    if (npairs == 0) return RSK_OK;
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = rsk_upload_mu_tables(ctx);
    if (rc != RSK_OK) return rc;
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    rsk_scratch ws(ctx);
    uint32_t *d_iq, *d_it, *d_cand, *d_ncand;
    uint8_t *d_rawf, *d_rawr, *d_rev_pos, *d_is_cand, *d_pass;
    int16_t *d_thr;
    int32_t *d_fwd = nullptr, *d_rev = nullptr;
    if ((rc = ws.alloc(&d_iq, npairs)) || (rc = ws.alloc(&d_it, npairs)) || (rc = ws.alloc(&d_cand, npairs)) || (rc = ws.alloc(&d_ncand, 1)) ||
        (rc = ws.alloc(&d_rawf, npairs)) || (rc = ws.alloc(&d_rawr, npairs)) || (rc = ws.alloc(&d_rev_pos, npairs)) ||
        (rc = ws.alloc(&d_is_cand, npairs)) || (rc = ws.alloc(&d_pass, npairs)) || (rc = ws.alloc(&d_thr, npairs)))
        return rc;
    if (fwd && (rc = ws.alloc(&d_fwd, npairs)) != RSK_OK) return rc;
    if (rev && (rc = ws.alloc(&d_rev, npairs)) != RSK_OK) return rc;
    RSK_HIP(hipMemcpyAsync(d_iq, iq, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_it, it, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_ncand, 0, 4, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_is_cand, 0, npairs, ctx->stream));
    if ((rc = musw_run_pairlist_dev(ctx, q, t, d_iq, d_it, nullptr, npairs, 0, gap_open, gap_ext, d_rawf, nullptr)) != RSK_OK) return rc;
    const unsigned nb = (unsigned) ((npairs + 255) / 256);
    hipLaunchKernelGGL(k_mfp_candidates, dim3(nb), dim3(256), 0, ctx->stream, d_rawf, npairs, omega, omega_fwd, d_cand, d_thr, d_ncand);
    uint32_t ncand = 0;
    RSK_HIP(hipMemcpyAsync(&ncand, d_ncand, 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    // the caller does not ask for the reverse scores: pairs that have failed may stop early (see musw_args::thr)
    const bool early = !rev && !(getenv("RSK_MUSW_EARLY_EXIT") && atoi(getenv("RSK_MUSW_EARLY_EXIT")) == 0);
    if ((rc = musw_run_pairlist_dev(ctx, q, t, d_iq, d_it, d_cand, ncand, 1, gap_open, gap_ext, d_rawr, early ? d_thr : nullptr)) != RSK_OK) return rc;
    if (ncand) hipLaunchKernelGGL(k_mfp_rev_by_pos, dim3((ncand + 255) / 256), dim3(256), 0, ctx->stream, d_rawr, d_cand, (size_t) ncand, d_rev_pos, d_is_cand);
    hipLaunchKernelGGL(k_mfp_verdict, dim3(nb), dim3(256), 0, ctx->stream, d_rawf, d_rev_pos, d_is_cand, npairs, omega, d_pass, d_fwd, d_rev);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RSK_HIP(hipMemcpyAsync(pass, d_pass, npairs, hipMemcpyDeviceToHost, ctx->stream));
    if (fwd) RSK_HIP(hipMemcpyAsync(fwd, d_fwd, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (rev) RSK_HIP(hipMemcpyAsync(rev, d_rev, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->mf_candidates = ncand;
    return RSK_OK;
}

// ---------------------------------------------------------------------------------------------
// D1: the reference's dead-but-named Mu kernels, pair-list form (not on the -search path; kept simple)
// ---------------------------------------------------------------------------------------------
static __device__ __constant__ float c_mu_f32[36 * 36];        // ScoreMx_Mu (mumx_data.cpp:3)

// SWFastPinop swfastpinop.cpp:6-78: int32 3-state local DP whose D/I states open from the diagonal
// predecessor (SavedM0), Open/Ext negative.  One thread per pair, row arrays in global scratch.
__global__ void k_mu_pinop(const uint8_t *q_mu, const uint32_t *q_off, const uint32_t *q_len, const uint8_t *t_mu, const uint32_t *t_off,
                           const uint32_t *t_len, const uint32_t *iq, const uint32_t *it, uint32_t npairs, int open, int ext,
                           int *scratch, size_t stride, int32_t *out)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const uint8_t *A = q_mu + q_off[iq[p]], *B = t_mu + t_off[it[p]];
    const int LA = (int) q_len[iq[p]], LB = (int) t_len[it[p]];
    int *Mrow = scratch + (size_t) p * stride, *Drow = Mrow + LB;
    for (int j = 0; j < LB; ++j) { Mrow[j] = 0; Drow[j] = 0; }
    int best = 0, M0 = 0;
    for (int i = 0; i < LA; ++i) {
        const signed char *srow = c_mu_int + 36 * A[i];
        int I0 = 0;
        for (int j = 0; j < LB; ++j) {
            const int saved = M0;
            int x = M0;
            if (Drow[j] > x) x = Drow[j];
            if (I0 > x) x = I0;
            if (0 >= x) x = 0;
            M0 = Mrow[j];
            x += srow[B[j]];
            if (x > best) best = x;
            Mrow[j] = x;
            const int md = saved + open;
            int d = Drow[j] + ext;
            if (md >= d) d = md;
            Drow[j] = d;
            I0 += ext;
            if (md >= I0) I0 = md;
        }
        M0 = 0;
    }
    out[p] = best;
}

// SWFastGaplessProfb swgaplessprofb.cpp:6-66: fused forward / reversed-A float gapless scores on the
// rows of ScoreMx_Mu; every diagonal is an independent sequential fp32 chain, so one thread walks one
// diagonal (same operand order as the reference) and the workgroup reduces the two maxima.
__global__ __launch_bounds__(256) void k_mu_gapless_profb(const uint8_t *q_mu, const uint32_t *q_off, const uint32_t *q_len,
                                                          const uint8_t *t_mu, const uint32_t *t_off, const uint32_t *t_len,
                                                          const uint32_t *iq, const uint32_t *it, float *out)
{
    __shared__ int sF, sR;
    const uint32_t p = blockIdx.x;
    const uint8_t *A = q_mu + q_off[iq[p]], *B = t_mu + t_off[it[p]];
    const int LA = (int) q_len[iq[p]], LB = (int) t_len[it[p]];
    if (threadIdx.x == 0) { sF = 0; sR = 0; }
    __syncthreads();
    float bestF = 0.0f, bestR = 0.0f;
    for (int d = threadIdx.x; d < LA + LB - 1; d += blockDim.x) {
        int i = LA - 1 - d; if (i < 0) i = 0;
        int j = d - (LA - 1); if (j < 0) j = 0;
        float xf = 0.0f, xr = 0.0f;
        for (; i < LA && j < LB; ++i, ++j) {
            if (xf < 0.0f) xf = 0.0f;
            if (xr < 0.0f) xr = 0.0f;
            const uint32_t b = B[j];
            xf += c_mu_f32[36 * A[i] + b];
            xr += c_mu_f32[36 * A[LA - i - 1] + b];
            if (xf > bestF) bestF = xf;
            if (xr > bestR) bestR = xr;
        }
    }
    atomicMax(&sF, __builtin_bit_cast(int, bestF));          // non-negative floats order like their bit patterns
    atomicMax(&sR, __builtin_bit_cast(int, bestR));
    __syncthreads();
    if (threadIdx.x == 0) out[p] = __builtin_bit_cast(float, sF) - __builtin_bit_cast(float, sR);
}

static int d1_stage_pairs(rsk_ctx *ctx, musw_ws &ws, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it, size_t n,
                          uint32_t **d_iq, uint32_t **d_it, const char *who)
{
    for (size_t p = 0; p < n; ++p)
        if (iq[p] >= q->n || it[p] >= t->n) { rsk_set_error("%s: pair %zu out of range", who, p); return RSK_E_INVALID; }
    int rc;
    if ((rc = ws.alloc(d_iq, n)) != RSK_OK) return rc;
    if ((rc = ws.alloc(d_it, n)) != RSK_OK) return rc;
    RSK_HIP(hipMemcpyAsync(*d_iq, iq, n * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(*d_it, it, n * 4, hipMemcpyHostToDevice, ctx->stream));
    return RSK_OK;
}

extern "C" int rsk_mu_pinop_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it, size_t npairs,
                                  int open, int ext, int32_t *scores)
{
    if (!ctx || !q || !t || (npairs && (!iq || !it || !scores))) { rsk_set_error("rsk_mu_pinop_pairs: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_pinop_pairs: chain set has no Mu letters"); return RSK_E_INVALID; }
    if (open > 0 || ext > 0 || open < -128 || ext < -128) { rsk_set_error("rsk_mu_pinop_pairs: Open/Ext are int8 <= 0 (swfastpinop.cpp:9)"); return RSK_E_INVALID; }
    if (npairs == 0) return RSK_OK;
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = rsk_upload_mu_tables(ctx);
    if (rc != RSK_OK) return rc;
    musw_ws ws(ctx);
    uint32_t *d_iq, *d_it;
    if ((rc = d1_stage_pairs(ctx, ws, q, t, iq, it, npairs, &d_iq, &d_it, "rsk_mu_pinop_pairs")) != RSK_OK) return rc;
    uint32_t maxLB = 1;
    for (size_t p = 0; p < npairs; ++p) maxLB = std::max(maxLB, t->len[it[p]]);
    const size_t stride = 2 * (size_t) maxLB;
    int *d_scratch;
    int32_t *d_out;
    if ((rc = ws.alloc(&d_scratch, stride * npairs)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&d_out, npairs)) != RSK_OK) return rc;
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_mu_pinop, dim3((unsigned) ((npairs + 63) / 64)), dim3(64), 0, ctx->stream, q->d_mu, q->d_off, q->d_len, t->d_mu,
                       t->d_off, t->d_len, d_iq, d_it, (uint32_t) npairs, open, ext, d_scratch, stride, d_out);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RSK_HIP(hipMemcpyAsync(scores, d_out, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}

extern "C" int rsk_mu_gapless_profb_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it,
                                          size_t npairs, float *scores)
{
    if (!ctx || !q || !t || (npairs && (!iq || !it || !scores))) { rsk_set_error("rsk_mu_gapless_profb_pairs: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_gapless_profb_pairs: chain set has no Mu letters"); return RSK_E_INVALID; }
    if (npairs == 0) return RSK_OK;
    if (npairs > 0x7FFFFFFFull) { rsk_set_error("rsk_mu_gapless_profb_pairs: too many pairs"); return RSK_E_RANGE; }
    RSK_HIP(hipSetDevice(ctx->device));
    {
        static std::atomic<int> up[64];
        const int urc = rsk_once_per_device(up, ctx->device, [&]() -> int {
            RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_mu_f32), rsk_mu_f32, sizeof(rsk_mu_f32)));
            return RSK_OK;
        });
        if (urc != RSK_OK) return urc;
    }
    musw_ws ws(ctx);
    uint32_t *d_iq, *d_it;
    int rc;
    if ((rc = d1_stage_pairs(ctx, ws, q, t, iq, it, npairs, &d_iq, &d_it, "rsk_mu_gapless_profb_pairs")) != RSK_OK) return rc;
    float *d_out;
    if ((rc = ws.alloc(&d_out, npairs)) != RSK_OK) return rc;
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_mu_gapless_profb, dim3((unsigned) npairs), dim3(256), 0, ctx->stream, q->d_mu, q->d_off, q->d_len, t->d_mu, t->d_off,
                       t->d_len, d_iq, d_it, d_out);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RSK_HIP(hipMemcpyAsync(scores, d_out, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}

extern "C" int rsk_mu_filter_last_work(rsk_ctx *ctx, uint64_t *pairs, uint64_t *candidates)
{
    if (!ctx) { rsk_set_error("rsk_mu_filter_last_work: ctx is NULL"); return RSK_E_INVALID; }
    if (pairs) *pairs = ctx->mf_pairs;
    if (candidates) *candidates = ctx->mf_candidates;
    return RSK_OK;
}

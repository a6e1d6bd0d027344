// k_sw_float.hip -- the main alignment of reseek -search on gfx950:
//   P5 DSSAligner::SetSMx_NoRev dssaligner.cpp:529  S(i,j) = ((((((M0+M1)+M2)+M3)+M4)+M5)+M6)+M7, fp32
//   P6 SWFast sw.cpp:79 + TraceBackBitSW sw.cpp:8   3-state affine local SW, 1 trace byte per cell
//   P7 GetLDDT_mu_fast lddt.cpp:63 (+ the test statistic of CalcEvalue dssaligner.cpp:852-889)
// Bit-exact: every fp32 operation of the reference is executed with the same operands in the same
// order (this file is compiled with -ffp-contract=off; no FMA, no reassociation).
//
// The reference evaluates one DP "grid point" (i,j) at a time (sw.cpp:119-197):
//   in : m = DPM[i][j], d = DPD[i][j], n = DPI[i][j], S(i,j)
//   out: DPM[i+1][j+1] = max*(m, d, n, 0) + S      (max* = the reference's tie order M, D(>), I(>), 0(>=))
//        DPD[i+1][j]   = max(m + Open (>=), d + Ext),   DPI[i][j+1] = max(m + Open (>=), n + Ext)
//        TB[i][j]      = how each of the three was chosen.
// A point depends only on its up-left / up / left neighbours, so any evaluation order gives the
// same bits.  Layout on the GPU (same systolic scheme as the Mu filter, k_mu_sw.hip):
//   * chain A is cut into strips of R = 16 rows that live in the VGPRs of one lane (DPM-diagonal and
//     DPI per row); the g = ceil(LA/16) strips of a pair sit on g consecutive lanes, one column behind
//     each other; the bottom row (DPM, DPD) moves to the next lane with two v_mov_b32_dpp wave_shr:1.
//     A wave works on floor(64/g) pairs at once.
//   * S(i,j) is fused: the 8 weighted feature tables (8.8 KB) sit in LDS; each lane keeps the table
//     row offsets of its 16 A-rows packed in registers, the B column contributes 8 byte offsets.
//   * trace bytes of the 16 rows of a lane are one 16-byte store into the wave's step-major trace block
//     [step][lane][16] in HBM: 1 KB of consecutive bytes per step (the only HBM-heavy stream of the path: 1 B per cell).
//   * a second kernel walks the trace (one thread per pair), a third computes LDDT over the aligned
//     columns (one wave per pair).
// That is the per-pair kernel (k_sw_float).  Groups of pairs that share a chain -- a query against a database -- take the
// query-profile kernel k_sw_qp further down: 16 lanes per pair, R = 4 .. 12 rows per lane, the shared chain expanded into
// an LDS profile whose rows are whole 256-byte bank rows, trace as four SGPR masks per cell in diagonal order.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <numeric>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "rsk_internal.h"
#include "rsk_tables_data.h"

#ifndef SWF_R
#define SWF_R 16                     // rows per lane (multiple of 4)
#endif
#ifndef SWF_WPE
#define SWF_WPE 2                    // waves per SIMD the kernel is built for (see k_sw_float)
#endif
#define SWF_PADR(L) ((((L) + SWF_R - 1) / SWF_R) * SWF_R)
#define SWF_WAVES 4
#define SWF_MINUS_INF (-9e9f)        // xdpmem.h:6
#define TB_DM 0x01                   // tracebit.h:4-8
#define TB_IM 0x02
#define TB_MD 0x04
#define TB_MI 0x08
#define TB_SM 0x10

// feature tables in LDS: table f at float offset swf_toff[f], row stride = alphabet size; then a pad row
struct swf_tables {
    float t[400 + 7 * 256 + 400];
};
static __device__ __constant__ swf_tables c_swf_tables;
static const int h_swf_toff[8] = { 0, 400, 656, 912, 1168, 1424, 1680, 1936 };
#define SWF_PAD_OFF 2192            // 400 floats of -1e30 (strip rows beyond the chain end, any step offset)
#define SWF_TABLE_FLOATS (2192 + 400)

static int swf_upload_tables(rsk_ctx *ctx)
{
    static std::atomic<int> done[64];
    return rsk_once_per_device(done, ctx->device, [&]() -> int {
    swf_tables h;
    for (int i = 0; i < SWF_TABLE_FLOATS; ++i) h.t[i] = -1e30f;
    for (int f = 0; f < RSK_NFEATURES; ++f) {
        const int as = (int) rsk_feature_alpha[f];
        for (int a = 0; a < as; ++a)
            for (int b = 0; b < as; ++b) h.t[h_swf_toff[f] + a * as + b] = rsk_feature_mx[f][a * RSK_FEATURE_DIM + b];
    }
    RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_swf_tables), &h, sizeof(h)));
    return RSK_OK;
    });
}

struct swf_item {            // one wave's work: pairs [first, first+count), g lanes per pair, row groups of g strips
    uint32_t first, count, g, ngroups;
    // trace block of the item: ngroups x ncol steps of 64 lanes x 16 B (one trace byte per register row), step-major --
    // the 64 lanes of a step store 1 KB of consecutive bytes.  (Round 2 kept a [column][row] block per PAIR: the systolic
    // skew puts neighbouring lanes on different columns, so every 16-byte store hit a line of its own and the writes
    // reached HBM as partial lines, 22 GB for 11 GB of trace.)
    uint32_t ncol, pad;
    uint64_t tb_base;
};

struct swf_args {
    // "strip" chain set (rows kept in registers) and "step" chain set (one residue per step).
    // normal orientation: strip = A (query rows), step = B; transposed: strip = B, step = A.
    const uint8_t *a_prof;   // A: [8][npadA] feature-major letters
    const uint32_t *a_off, *a_len;
    const uint16_t *a_ra;    // A: [npadA][8] table ROW offsets in bytes (letter * alphabet * 4)
    const uint16_t *a_cb;    // A: [npadA][8] letter * 4 (k_sw_qp, strips along B)
    const uint8_t *b_prof;   // B: [8][npadB]
    const uint16_t *b_cb;    // B: [npadB][8] table COLUMN offsets in bytes (letter*4)
    const uint32_t *b_off, *b_len;
    const uint32_t *ia, *ib; // pair lists (sorted order used by the items)
    size_t a_npad, b_npad;
    const swf_item *items;
    uint32_t nitems;
    float open, ext;
    uint8_t *tb;             // trace blocks, pair p at tb + tb_off[p], layout [j][LApad]
    const uint64_t *tb_off;
    float *score;            // per pair
    uint32_t *besti, *bestj;
    int *bnd;                // boundary rows of multi-group pairs: 2 ints (float bits) per step, at bnd + bnd_off[p]
    const uint64_t *bnd_off;
    unsigned long long *clk; // k_sw_qp: {shader cycles, 100 MHz ticks} summed over the workgroups (rsk_path_counters), or NULL
};

__device__ __forceinline__ float dpp_shr1_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x),
                                                                  0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}

// w = 2 * w + (x > y) / (x >= y): the comparison bit enters through the carry
#define SWQ_BIT_GT(w, x, y) asm("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x), "v"(y) : "vcc")
#define SWQ_BIT_GE(w, x, y) asm("v_cmp_ge_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x), "v"(y) : "vcc")
#define SWQ_BIT_0GE(w, y) asm("v_cmp_ge_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(y) : "vcc")


// v_max_f32 without the canonicalising v_max x, x, x that fmaxf() costs per fresh operand (no NaNs occur here)
__device__ __forceinline__ float swq_max(float x, float y)
{
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

__device__ __forceinline__ float swq_max3_0(float x, float y)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// T = false: strips along A (rows), the wave steps over the columns of B; trace block TB[j][LApad].
// T = true : strips along B (columns), the wave steps over the rows of A;  trace block TB[i][LBpad].
// Built for 2 waves per SIMD (<= 256 VGPRs, no spills).  The kernel is bound by the LDS: 8 random ds_read_b32 per cell,
// ~3.2 distinct addresses on the busiest bank of a 32-lane group = ~6 LDS cycles per wave-instruction (PMC, r02d: LDS
// busy 75 % of the kernel's cycles, two thirds of it bank conflicts; VALU issue 58 %).  Variants measured on the 199 k
// SCOP40-shaped -sensitive survivors (profiles/r02c_sw_float_notes.txt): R = 16 / 2 waves 20.2 ms, R = 12 / 3 waves 20.8,
// R = 16 / 3 waves (54 spilled VGPRs) 22.6, R = 20 and 24 / 2 waves 21.6, R = 12 / 4 waves (60 spills) 28.1.
template <bool T>
__global__ __launch_bounds__(64 * SWF_WAVES) __attribute__((amdgpu_waves_per_eu(SWF_WPE, SWF_WPE))) void k_sw_float(swf_args a, uint32_t item_base)
{
    __shared__ __attribute__((aligned(16))) float tab[SWF_TABLE_FLOATS];
    for (int i = threadIdx.x; i < SWF_TABLE_FLOATS; i += blockDim.x) tab[i] = c_swf_tables.t[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t item_id = blockIdx.x * SWF_WAVES + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // wave-uniform
    if (item_id >= a.nitems) return;
    const swf_item it = a.items[item_base + item_id];
    const uint32_t g = it.g;
    const uint32_t pr = lane / g, st = lane - pr * g;
    const bool active = pr < it.count;
    const uint32_t p = it.first + (active ? pr : 0);
    const uint32_t A = a.ia[p], B = a.ib[p];
    const uint32_t LA0 = a.a_len[A], LB0 = a.b_len[B];
    // from here on "LA"/rows/i refer to the strip chain and "LB"/columns/j to the step chain
    const uint32_t LA = T ? LB0 : LA0, LB = T ? LA0 : LB0;
    const uint32_t LApad = SWF_PADR(LA);
    const float Open = a.open, Ext = a.ext;
    const uint16_t *bcb = T ? (a.a_ra + (size_t) a.a_off[A] * 8) : (a.b_cb + (size_t) a.b_off[B] * 8);
    const char *tabb = (const char *) tab;
    const int toffb[8] = { 0 * 4, 400 * 4, 656 * 4, 912 * 4, 1168 * 4, 1424 * 4, 1680 * 4, 1936 * 4 };
    const int asz[8] = { 20, 16, 16, 16, 16, 16, 16, 16 };
    float best = 0.0f;
    uint32_t bi = 0xFFFFFFFFu, bj = 0xFFFFFFFFu;
    int *bnd = a.bnd ? a.bnd + a.bnd_off[p] : nullptr;     // only multi-group pairs (one pair per wave) use it
    // best cell of each register row: first step at which the row reached its maximum (strict >)
    float rb[SWF_R];
    uint32_t rj[SWF_R];

    // Chains with more than 64 strips are processed in row groups of 64 strips; the bottom row of a
    // group goes through `bnd` (HBM, agent-scope accesses: written by lane 63, read by lane 0 later).
    for (uint32_t rg = 0; rg < it.ngroups; ++rg) {
    const uint32_t i0 = (rg * g + st) * SWF_R;
    const bool lane_has_rows = active && i0 < LA;
    const bool writes_bnd = (rg + 1 < it.ngroups) && st == g - 1;
    const bool reads_bnd = rg > 0 && st == 0;

    // table row offsets (bytes) of this lane's 16 rows, two features per dword
    uint32_t ro[SWF_R][4];
#pragma unroll
    for (int r = 0; r < SWF_R; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) ro[r][k] = (uint32_t) (SWF_PAD_OFF * 4) * 0x10001u;
    if (lane_has_rows) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const uint8_t *src = T ? (a.b_prof + (size_t) f * a.b_npad + a.b_off[B] + i0)
                                   : (a.a_prof + (size_t) f * a.a_npad + a.a_off[A] + i0);   // 16-byte aligned (chains padded to 16)
            uint32_t ww[SWF_R / 4];                 // i0 is a multiple of 4 and the chain set has tail padding
#pragma unroll
            for (int k = 0; k < SWF_R / 4; ++k) ww[k] = ((const uint32_t *) src)[k];
#pragma unroll
            for (int r = 0; r < SWF_R; ++r) {
                const uint32_t letter = (ww[r >> 2] >> (8 * (r & 3))) & 0xFF;
                // strip part of the table address: row offset (normal) or column offset (transposed)
                uint32_t off = (uint32_t) toffb[f] + letter * (uint32_t) (T ? 4 : asz[f] * 4);
                if (i0 + r >= LA) off = SWF_PAD_OFF * 4;
                if (f & 1) ro[r][f >> 1] = (ro[r][f >> 1] & 0xFFFFu) | (off << 16);
                else ro[r][f >> 1] = (ro[r][f >> 1] & 0xFFFF0000u) | off;
            }
        }
    }

    float Md[SWF_R], In[SWF_R];      // DPM[i][j] (diagonal input of row r at the next column), DPI[i][j]
#pragma unroll
    for (int r = 0; r < SWF_R; ++r) { Md[r] = SWF_MINUS_INF; In[r] = SWF_MINUS_INF; rb[r] = 0.0f; rj[r] = 0; }
    if (st == 0 && rg == 0) Md[0] = 0.0f;   // DPM[0][0] = 0 (sw.cpp:117)
    float hand_m = SWF_MINUS_INF, hand_d = SWF_MINUS_INF;   // bottom row of this strip at its previous column
    float carry_in = SWF_MINUS_INF;                         // DPM[i0][j] from the lane above (arrives one step early)
    uint8_t *tbp = a.tb + it.tb_base + ((size_t) rg * it.ncol * 64 + lane) * 16;      // + step * 1024

    uint32_t ncol = lane_has_rows ? (LB + st) : 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) ncol = max(ncol, (uint32_t) __shfl_xor((int) ncol, s, 64));

    uint4 cbn = make_uint4(0, 0, 0, 0);
    if (lane_has_rows) cbn = *(const uint4 *) bcb;

    for (uint32_t col = 0; col < ncol; ++col) {
        const int j = (int) col - (int) st;
        // bottom row of the strip above: its DPD chain value for this column j and its new DPM (for column j+1)
        float in_m = dpp_shr1_f(hand_m);
        float in_d = dpp_shr1_f(hand_d);
        if (reads_bnd && j >= 0 && (uint32_t) j < LB) {
            // previous group: bnd[2j] = DPM[i0][j+1] (after its column j), bnd[2j+1] = DPD[i0][j]
            in_d = __builtin_bit_cast(float, __hip_atomic_load(bnd + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            carry_in = j > 0 ? __builtin_bit_cast(float, __hip_atomic_load(bnd + 2 * (j - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                             : SWF_MINUS_INF;
        }
        if (lane_has_rows && j >= 0 && (uint32_t) j < LB) {
            const uint4 cb = cbn;
            cbn = *(const uint4 *) (bcb + (size_t) (j + 1) * 8);      // prefetch (chain set has tail padding)
            const uint32_t cbw[4] = { cb.x, cb.y, cb.z, cb.w };
            uint32_t cbo[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) cbo[f] = (f & 1) ? (cbw[f >> 1] >> 16) : (cbw[f >> 1] & 0xFFFFu);
            // row 0 inputs
            // chain state entering row 0 of the strip: DPD[i0][j] (normal) / DPI[i][j0] (transposed)
            float ch = (st == 0 && rg == 0) ? SWF_MINUS_INF : in_d;
            if (st != 0 || rg != 0) Md[0] = carry_in;                   // DPM[i0][j] = bottom DPM of the strip above at column j-1
            else if (j > 0) Md[0] = SWF_MINUS_INF;                      // DPM[0][j>0] = -inf (sw.cpp:102-111)
            float carry = SWF_MINUS_INF;                                // DPM[i0+r][j+1] produced by row r-1
            // branch-free cell: the five comparisons of the recurrence enter a trace byte through the carry
            // (same bit order as k_sw_qp: DM candidate, IM candidate, SM, MD, MI; decoded in swf_trace_flags);
            // the byte of row r sits at bits 8 * (3 - r % 4) of its dword
            uint32_t tbw[SWF_R / 4];
            uint32_t w = 0;
#pragma unroll
            for (int r = 0; r < SWF_R; ++r) {
                // S(i,j): features summed 0 -> 7 (dssaligner.cpp:553-597)
                float S = *(const float *) (tabb + ((ro[r][0] & 0xFFFFu) + cbo[0]));
                S += *(const float *) (tabb + ((ro[r][0] >> 16) + cbo[1]));
                S += *(const float *) (tabb + ((ro[r][1] & 0xFFFFu) + cbo[2]));
                S += *(const float *) (tabb + ((ro[r][1] >> 16) + cbo[3]));
                S += *(const float *) (tabb + ((ro[r][2] & 0xFFFFu) + cbo[4]));
                S += *(const float *) (tabb + ((ro[r][2] >> 16) + cbo[5]));
                S += *(const float *) (tabb + ((ro[r][3] & 0xFFFFu) + cbo[6]));
                S += *(const float *) (tabb + ((ro[r][3] >> 16) + cbo[7]));
                const float m = Md[r];
                const float d = T ? In[r] : ch;     // DPD of this grid point
                const float n = T ? ch : In[r];     // DPI of this grid point
                Md[r] = carry;                      // becomes this register's diagonal input at the next step
                if (r & 3) w <<= 3;
                // MATCH (sw.cpp:123-155)
                SWQ_BIT_GT(w, d, m);                 // TB_DM candidate (sw.cpp:127)
                const float x1 = swq_max(m, d);
                SWQ_BIT_GT(w, n, x1);                // TB_IM (sw.cpp:135)
                const float x2 = swq_max(x1, n);
                SWQ_BIT_0GE(w, x2);                  // TB_SM (sw.cpp:143)
                const float xM = swq_max(x2, 0.0f) + S;
                if (xM > rb[r]) { rb[r] = xM; rj[r] = (uint32_t) j; }
                carry = xM;
                // DELETE (sw.cpp:163-176): DPD[i+1][j];  INSERT (sw.cpp:178-191): DPI[i][j+1]
                const float md = m + Open;
                const float de = d + Ext;
                SWQ_BIT_GE(w, md, de);               // TB_MD (sw.cpp:166)
                const float dd = swq_max(md, de);
                const float ne = n + Ext;
                SWQ_BIT_GE(w, md, ne);               // TB_MI (sw.cpp:181)
                const float ni = swq_max(md, ne);
                if (T) { ch = ni; In[r] = dd; }
                else { ch = dd; In[r] = ni; }
                if ((r & 3) == 3) { tbw[r >> 2] = w; w = 0; }
            }
            hand_m = carry;      // DPM[i0+16][j+1]
            hand_d = ch;         // DPD[i0+16][j] (normal) / DPI[i][j0+16] (transposed)
            {
                static_assert(SWF_R == 16, "one 16-byte trace record per lane and step");
                *(uint4 *) (tbp + (size_t) col * 1024) = make_uint4(tbw[0], tbw[1], tbw[2], tbw[3]);
            }
            if (writes_bnd) {
                __hip_atomic_store(bnd + 2 * j, __builtin_bit_cast(int, hand_m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(bnd + 2 * j + 1, __builtin_bit_cast(int, hand_d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!reads_bnd) carry_in = in_m;   // value sent by the lane above at this step is for its column j+1 == our next column
    }
    // fold the rows of this group into the lane's best: highest score, then smallest i, then smallest j
#pragma unroll
    for (int r = 0; r < SWF_R; ++r) {
        const uint32_t ii = T ? rj[r] : i0 + r, jj = T ? i0 + r : rj[r];
        if (rb[r] > best || (rb[r] == best && best > 0.0f && (ii < bi || (ii == bi && jj < bj)))) { best = rb[r]; bi = ii; bj = jj; }
    }
    if (it.ngroups > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // boundary stores of this group have left the wave
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    }
    }   // row groups
    // reduce (best, bi, bj) over the strips of each pair with the row-major-first rule (sw.cpp:153-158)
    for (uint32_t dlt = 1; dlt < g; ++dlt) {
        const int src = (lane + dlt) & 63;
        const float ob = __shfl(best, src, 64);
        const uint32_t oi = (uint32_t) __shfl((int) bi, src, 64), oj = (uint32_t) __shfl((int) bj, src, 64);
        if (st + dlt < g) {
            if (ob > best || (ob == best && ob > 0.0f && (oi < bi || (oi == bi && oj < bj)))) { best = ob; bi = oi; bj = oj; }
        }
    }
    if (active && st == 0) {
        a.score[p] = best;
        a.besti[p] = bi;
        a.bestj[p] = bj;
    }
}


// ---------------------------------------------------------------------------------------------
// k_sw_qp: the same recurrence for GROUPS of pairs that share their strip chain (a query against
// its filter survivors).  One workgroup = one group: it first expands the strip chain into a
// "query profile" in LDS,
//     QP[f][c][i] = feature table f, strip-chain letter of residue i, against step-chain letter c
//     (132 (f, c) rows x 4 B per residue),
// stored as records of R consecutive residues, so that a lane fetches the S-contributions of its
// whole strip for one feature with ceil(R/4) ds_read_b128 and no per-cell address arithmetic
// (the legacy kernel above spends 8 ds_read_b32 + 16 address ops per cell).  Waves claim batches
// of 4 pairs from an LDS counter.
//
// Geometry (r04c): a pair ALWAYS takes 16 lanes, and what adapts to the chain is R, the rows a lane keeps
// (R = 4 .. 12, a template parameter): a chain of L residues runs in P = ceil(L / 192) passes of 16 strips of
// R = ceil(L / 16P) rows.  The lanes of a pair then sit on a 16-lane boundary, a (f, c) row of the profile is a whole
// number of 256-byte bank rows (16 strips x 16 B per quad of residues), and the 16 lanes of every ds_read_b128 lane
// group {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- which always belong to two different pairs, i.e. read two
// unrelated rows -- have 16 different strip numbers = 16 different 16-byte bank slots: NO bank conflicts whatever the
// letters.  (r01-r04b gave a pair ceil(L / 12) lanes of 12 rows and packed floor(64 / g) pairs into a wave: rows of
// 1200 B at arbitrary bank offsets, 44 % of the LDS cycles were conflicts and 19 % of the wave cycles were spent waiting
// to issue LDS instructions; measured on 181..192-residue queries, where both layouts fill all 64 lanes: 22.3 -> 19.8 ms.)
// The LDS holds 4 quads of a row = 4 / ceil(R/4) passes (one "segment"); longer chains are processed in segments, one profile
// after the other, rows meeting through `bnd`.
// Trace: the 5 comparisons of a cell are v_cmp's that write their 64-lane masks to SGPR pairs, and the masks leave
// the wave through SCALAR stores (s_store_dwordx4, two masks each): no VALU op folds bits into a word and no
// vector store carries them (round 1 shifted them into a dword with v_addc_co, 5 extra VALU ops per cell, and
// wrote 8 B per lane and column at a lane stride).  A cell's record is FOUR masks, 32 bytes: the three "into M" flags
// {DM, IM, SM} are read by TraceBackBitSW in the order stop, I, D (sw.cpp:33-50), i.e. they encode one of four moves, so
// two masks carry them -- {SM | IM, SM | (DM & ~IM)}, three scalar ops on the SGPR pairs -- next to MD and MI.
// The trace of a wave batch is a block of records indexed [step - row + R - 1][row]: the cells of a lane's DIAGONAL are
// consecutive records.  k_traceback, which walks diagonals, finds four steps in one 128-byte line and needs one 16-byte
// load per step; with the [step][row] order of r01-r04b (40-byte records, two loads) every step was a line of its own and
// the kernel ran at the HBM's random-line rate.  For the writer the row offsets are immediates either way.
// ---------------------------------------------------------------------------------------------
#define SWQ_RMIN 4
#define SWQ_RMAX 12
#define SWQ_GS 16                                  // lanes (strips per pass) of a pair
#define SWQ_NPW (64 / SWQ_GS)                      // pairs per wave batch
#define SWQ_NQ(R) (((R) + 3) / 4)                  // ds_read_b128 per feature and step
#define SWQ_NPF(R) (4 / SWQ_NQ(R))                 // passes per LDS profile
#define SWQ_COLB(R) ((R) * 32)                     // trace of one (step - row) column of a wave batch: R rows x 4 masks of 64 lanes
#ifndef SWQ_NW
#define SWQ_NW 16
#endif
#define SWQ_NFC 132                               // 20 + 7 * 16 (feature, step letter) combinations
#define SWQ_LDS_BYTES ((size_t) SWQ_NFC * 4 * SWQ_GS * 16 + 16)

// one workgroup item: `count` consecutive pairs (sorted order) of one group, R rows per lane.
// A wave batch is 4 pairs, 16 lanes each; pass k covers the strips [16k, 16k + 16), the rows of consecutive passes meet
// through `bnd` -- run back to back by the wave that claimed the batch.  The trace blocks, one per (pass, wave batch), are
// ncol columns each and start at tb + tb_base: block (pass, b) is number pass * nbatch + b with nbatch = ceil(count / 4).
struct swq_item { uint32_t first, count, ncol, R; uint64_t tb_base; };

// comparison -> 64-lane mask in an SGPR pair (lanes outside EXEC read 0)
#define SWQ_MASK_GT(m, x, y) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x), "v"(y))
#define SWQ_MASK_GE(m, x, y) asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x), "v"(y))
#define SWQ_MASK_0GE(m, y) asm volatile("v_cmp_ge_f32_e64 %0, 0, %1" : "=s"(m) : "v"(y))
// two masks / one mask to trace memory (wave-uniform address)
#ifndef SWQ_EXPERIMENT_NO_TRACE_STORE
#define SWQ_STORE2(m0, m1, ptr, off) \
    asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(__uint128_t(m0) | (__uint128_t(m1) << 64)), "s"(ptr), "n"(off) : "memory")
#else   // timing experiment only (tools/exp): the masks are computed and dropped -- what the trace stores cost
#define SWQ_STORE2(m0, m1, ptr, off) asm volatile("" :: "s"(m0), "s"(m1), "s"(ptr))
#endif

// Best cell of a row (sw.cpp:153-158: highest score, first column): the running pair {~column, score} is kept as ONE 64-bit
// value and updated by v_max_f64 -- positive floats order like their bit patterns, and so do positive doubles: the score
// in the high word decides, among equal scores the larger ~column = the earlier column wins, and a score <= 0 (a negative
// or zero double) never displaces the initial {0, +0.0}.  One instruction per cell instead of v_cmp_gt_f32 + 2
// v_cndmask_b32 (tools/exp/ubench_f64max.hip: v_max_f64 issues at the rate of the 32-bit ops).  The operand {~column, xM}
// must be an aligned register pair: xM = xc + S is formed directly in v127 next to ~column in v126 (two registers the
// asm statements of a step name themselves; ~column is a per-step constant, so the allocator keeps it there).
// The update of row r is issued at the end of row r + 1's code (volatile, like the mask comparisons it follows): by then
// row r + 1 has read its old diagonal value, so v127 is copied straight into that register -- the one move per cell the
// diagonal rotation needs anyway.
#ifndef SWQ_EXPERIMENT_NO_BEST
#define SWQ_BEST64(acc, xm, xc, s, nj) \
    asm volatile("v_add_f32_e32 v127, %2, %3\n\tv_max_f64 %0, %0, v[126:127]" : "+v"(acc), "={v127}"(xm) : "v"(xc), "v"(s), "{v126}"(nj))
#else   // timing experiment only (tools/exp/swq_pmc.sh): what the 64-bit maximum costs (results are wrong: no best cell)
#define SWQ_BEST64(acc, xm, xc, s, nj) \
    asm volatile("v_add_f32_e32 v127, %2, %3" : "+v"(acc), "={v127}"(xm) : "v"(xc), "v"(s), "{v126}"(nj))
#endif

typedef float swq_v2f __attribute__((ext_vector_type(2)));
typedef float swq_v4f __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) swq_v4f *swq_ldsp;   // volatile: see the fetch in swq_group

template <bool T, int R>
__device__ __forceinline__ void swq_group(const swf_args &a, const swq_item &it, float4 *qp4)
{
    constexpr int NQ = SWQ_NQ(R), NPF = SWQ_NPF(R), G = SWQ_GS * NPF, COLB = SWQ_COLB(R);
    constexpr uint32_t ROWB = NQ * G * 16;           // bytes of a (f, c) row: a multiple of 256
    const uint32_t strip_chain = T ? a.ib[it.first] : a.ia[it.first];
    const uint32_t LA = T ? a.b_len[strip_chain] : a.a_len[strip_chain];     // strip chain length
    const uint32_t gtot = (LA + R - 1) / R;        // strips of the whole chain
    const uint32_t nseg = (gtot + G - 1) / G;
    const uint32_t nbatch = (it.count + SWQ_NPW - 1) / SWQ_NPW;
    uint32_t *next_batch = (uint32_t *) (qp4 + (size_t) SWQ_NFC * 4 * SWQ_GS);
    const int lane = threadIdx.x & 63;
    const uint32_t pr = lane / SWQ_GS, st = lane % SWQ_GS;
    const float Open = a.open, Ext = a.ext;
    for (uint32_t seg = 0; seg < nseg; ++seg) {
    const uint32_t sbase = seg * G;                // first strip of this segment
    const uint32_t g = min((uint32_t) G, gtot - sbase);
    // every wave is done with the previous profile.  The boundary rows (and the running best cells) are handed on within
    // this workgroup only -- an item runs on one CU, whose L1 is write-through and sees its own stores -- so they are plain
    // loads and stores ordered by s_waitcnt / the barrier.  (Agent-scope accesses go past the XCD's L2 on this part: measured, 31 ms instead of 23 for the 64-query
    // benchmark once most groups ran in two passes.)
    if (seg) { __threadfence_block(); __syncthreads(); }
    {
        const uint8_t *sp = T ? (a.b_prof + a.b_off[strip_chain]) : (a.a_prof + a.a_off[strip_chain]);
        const size_t snpad = T ? a.b_npad : a.a_npad;
        // float4 ((fc * NQ + quad) * G + strip) holds residues (sbase + strip) * R + quad * 4 + 0..3 of row fc: the
        // ds_read_b128 of the lanes of a pair are 16 B apart, and the NQ quads of a lane sit at immediate offsets
        // (quad * G * 16 B) from one address.  One record per thread and iteration.
        const uint32_t per_fc = NQ * g, tot = SWQ_NFC * per_fc;
        for (uint32_t idx = threadIdx.x; idx < tot; idx += blockDim.x) {
            const uint32_t fc = idx / per_fc, rem = idx - fc * per_fc;
            const uint32_t quad = rem / g, strip = rem - quad * g;
            const uint32_t i = (sbase + strip) * R + quad * 4;
            const uint32_t f = fc < 20 ? 0 : ((fc - 20) >> 4) + 1;
            const uint32_t c = fc < 20 ? fc : ((fc - 20) & 15);
            const uint32_t as = f == 0 ? 20 : 16, tof = f == 0 ? 0 : 400 + (f - 1) * 256;
            const float padv = f == 0 ? -1e30f : 0.0f;      // rows below the chain end: S = -1e30, never a maximum
            float v[4] = { padv, padv, padv, padv };
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (quad * 4 + w < (uint32_t) R && i + w < LA) {
                    const uint32_t letter = sp[(size_t) f * snpad + i + w];
                    v[w] = c_swf_tables.t[tof + (T ? c * as + letter : letter * as + c)];
                }
            }
            qp4[(fc * NQ + quad) * G + strip] = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (threadIdx.x == 0) *next_batch = 0;
    }
    __syncthreads();
    const uint32_t npass = (g + SWQ_GS - 1) / SWQ_GS;      // passes of this segment
    for (;;) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(next_batch, 1u);
        b = (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
        if (b >= nbatch) break;
        const uint32_t pidx = b * SWQ_NPW + pr;
        const bool active = pidx < it.count;
        const uint32_t p = it.first + (active ? pidx : 0);
        const uint32_t step_chain = T ? a.ia[p] : a.ib[p];
        const uint32_t LB = T ? a.a_len[step_chain] : a.b_len[step_chain];
        const uint16_t *bcb = T ? (a.a_cb + (size_t) a.a_off[step_chain] * 8) : (a.b_cb + (size_t) a.b_off[step_chain] * 8);
        // two rows of LB words per pair: a pass reads the row its predecessor wrote and writes the other one (reading and
        // writing one row in place puts loads and stores of the same cache lines in flight together: measured, +0.8 ms of 24)
        long long *bnd2 = gtot > SWQ_GS ? (long long *) (a.bnd + a.bnd_off[p]) : nullptr;
        // best cell of the pair over the passes of this segment (kept by the pair's lane st == 0)
        float pbest = 0.0f;
        uint32_t pbi = 0xFFFFFFFFu, pbj = 0xFFFFFFFFu;
        for (uint32_t pass = 0; pass < npass; ++pass) {
        const uint32_t gl = min((uint32_t) SWQ_GS, g - pass * SWQ_GS);   // strips of this pass
        const uint32_t sp = pass * SWQ_GS + st;                          // this lane's strip within the segment
        const bool on = active && st < gl;
        const uint32_t i0 = (sbase + sp) * R;
        // this lane's float4 slot within a quad block, as an LDS byte address (and the same from feature 4's first row on)
        const uint32_t qpl_lo = (uint32_t) (uintptr_t) ((swq_ldsp) qp4 + sp);
        const uint32_t qpl_hi = qpl_lo + (20 + 3 * 16) * ROWB;
        const bool first = seg == 0 && pass == 0;                        // the pass that holds row 0
        const bool last = sbase + pass * SWQ_GS + gl >= gtot;            // ... the last row
        const bool reads_bnd = !first && st == 0;
        const bool writes_bnd = !last && st == gl - 1;
        // trace block of this (pass, batch): wave-uniform address in SGPRs
        const unsigned long long *tblk;
        {
            const unsigned long long t0 = (unsigned long long) (a.tb + it.tb_base) +
                                          (unsigned long long) ((seg * NPF + pass) * nbatch + b) * it.ncol * COLB;
            const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) t0);
            const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (t0 >> 32));
            tblk = (const unsigned long long *) (((unsigned long long) hi << 32) | lo);
        }

        float Md[R], In[R];
        double rbj[R];                             // {~column, score}: see SWQ_BEST64
#pragma unroll
        for (int r = 0; r < R; ++r) rbj[r] = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) { Md[r] = SWF_MINUS_INF; In[r] = SWF_MINUS_INF; }
        if (st == 0 && first) Md[0] = 0.0f;
        float hand_m = SWF_MINUS_INF, hand_d = SWF_MINUS_INF, carry_in = SWF_MINUS_INF;
        uint32_t ncol = on ? (LB + st) : 0;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) ncol = max(ncol, (uint32_t) __shfl_xor((int) ncol, s, 64));
        uint4 cbn = make_uint4(0, 0, 0, 0);
        if (on) cbn = *(const uint4 *) bcb;
        // Rows of consecutive passes meet through `bnd` (one 8-byte word {M, D/I} per step).  Both directions are kept off
        // the step's critical path: every lane requests the word of its NEXT column right after it has consumed this step's
        // (only strip 0 uses it; the loop-carried register is then written by the load itself -- a copy would make the
        // compiler wait for the load on the spot), and the bottom row of a step is stored one step later at the same point,
        // so that the wait at the top of a step only ever covers operations issued a whole step earlier.
        // (a pass that has nothing to read points the load at a word of the pair's own offset table)
        const uint32_t gpass = seg * NPF + pass;                        // pass index over the whole chain
        long long *bnd = bnd2 ? bnd2 + (size_t) (gpass & 1) * LB : nullptr;                 // written by this pass
        const long long *bsrc = first ? (const long long *) (a.tb_off + p) : bnd2 + (size_t) ((gpass & 1) ^ 1) * LB;
        long long bnn = *bsrc;                     // boundary word of strip 0's column `col`

        for (uint32_t col = 0; col < ncol; ++col) {
            const int j = (int) col - (int) st;
            float in_m = dpp_shr1_f(hand_m);
            float in_d = dpp_shr1_f(hand_d);
            if (on && j >= 0 && (uint32_t) j < LB) {
                const uint4 cb = cbn;
                if (reads_bnd) {
                    in_m = __builtin_bit_cast(float, (int) (uint32_t) (unsigned long long) bnn);
                    in_d = __builtin_bit_cast(float, (int) ((unsigned long long) bnn >> 32));
                }
                // everything this step fetched is consumed: request the next step's data, then store the previous step's
                // bottom row -- all of it has the rows below to complete
                cbn = *(const uint4 *) (bcb + (size_t) (j + 1) * 8);
                bnn = bsrc[first ? 0u : min((uint32_t) j + 1, LB - 1)];
                if (!last) {                       // wave-uniform
                    // {hand_m, hand_d} still hold the bottom row of this lane's previous step (column j - 1)
                    if (writes_bnd && j > 0)
                        bnd[j - 1] = (long long) ((unsigned long long) (uint32_t) __builtin_bit_cast(int, hand_m) |
                                                  ((unsigned long long) (uint32_t) __builtin_bit_cast(int, hand_d) << 32));
                }
                const uint32_t cbw[4] = { cb.x, cb.y, cb.z, cb.w };
                swq_ldsp rec[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    // byte address of the lane's record of row (f, letter): (letter * 4) * (row bytes / 4) + lane base, one
                    // v_mad_u32_u16 that picks its half of the packed word itself; the feature's first row is a constant the
                    // ds_read carries as its immediate (features 4..7 relative to a second base: 16-bit immediates)
                    const uint32_t fcb = f == 0 ? 0 : 20 + (f - 1) * 16, fcb4 = 20 + 3 * 16;
                    uint32_t ad;
                    if (f & 1) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ad) : "v"(cbw[f >> 1]), "s"(ROWB / 4), "v"(f < 4 ? qpl_lo : qpl_hi));
                    else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(ad) : "v"(cbw[f >> 1]), "s"(ROWB / 4), "v"(f < 4 ? qpl_lo : qpl_hi));
                    rec[f] = (swq_ldsp) (uintptr_t) (ad + (f < 4 ? fcb : fcb - fcb4) * ROWB);
                }
                float ch = (st == 0 && first) ? SWF_MINUS_INF : in_d;
                if (st != 0 || !first) Md[0] = carry_in;
                else if (j > 0) Md[0] = SWF_MINUS_INF;
                float xc_p = 0.0f, S_p = 0.0f;                           // the row above: its best-cell update is still to come
                const uint32_t nj = ~(uint32_t) j;
                const unsigned long long *tcol = (const unsigned long long *) ((const char *) tblk + (size_t) col * COLB);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    // volatile keeps each fetch one ds_read_b128 (the SLP vectoriser would split it into b64 halves)
                    swq_v4f v[8];
#pragma unroll
                    for (int f = 0; f < 8; ++f) v[f] = rec[f][q * G];
                    swq_v2f Slo = v[0].lo, Shi = v[0].hi;              // v_pk_add_f32: two residues per add
#pragma unroll
                    for (int f = 1; f < 8; ++f) {
                        Slo += v[f].lo;
                        if (q * 4 + 2 < R) Shi += v[f].hi;
                    }
                    const float S4[4] = { Slo.x, Slo.y, Shi.x, Shi.y };
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = q * 4 + rr;
                        if (r >= R) continue;
                        unsigned long long tc[5];
                        const float m = Md[r];
                        const float d = T ? In[r] : ch;
                        const float n = T ? ch : In[r];
                        SWQ_MASK_GT(tc[0], d, m);            // TB_DM candidate (sw.cpp:127)
                        const float x1 = swq_max(m, d);
                        SWQ_MASK_GT(tc[1], n, x1);           // TB_IM (sw.cpp:135)
                        // max(m, d, n, 0) in one v_max3: it is 0 exactly when max(m, d, n) <= 0, the TB_SM test (sw.cpp:143)
                        const float xc = swq_max3_0(x1, n);
                        SWQ_MASK_0GE(tc[2], xc);
                        const float md = m + Open;
                        const float de = d + Ext;
                        SWQ_MASK_GE(tc[3], md, de);          // TB_MD (sw.cpp:166)
                        const float dd = swq_max(md, de);
                        const float ne = n + Ext;
                        SWQ_MASK_GE(tc[4], md, ne);          // TB_MI (sw.cpp:181)
                        const float ni = swq_max(md, ne);
                        if (T) { ch = ni; In[r] = dd; }
                        else { ch = dd; In[r] = ni; }
                        // xM of the row above = the diagonal value of this row at the next step (and its best-cell update)
                        if (r > 0) {
                            float xM;
                            SWQ_BEST64(rbj[r - 1], xM, xc_p, S_p, nj);
                            Md[r] = xM;
                        } else Md[0] = SWF_MINUS_INF;
                        xc_p = xc;
                        S_p = S4[rr];
                        // the record of (step, row r) sits in column step - r + R - 1 of the block: R - 1 - r columns on
                        {
                            const unsigned long long b1 = tc[2] | tc[1], b0 = tc[2] | (tc[0] & ~tc[1]);
                            SWQ_STORE2(b1, b0, tcol, (R - 1 - r) * COLB + r * 32);
                            SWQ_STORE2(tc[3], tc[4], tcol, (R - 1 - r) * COLB + r * 32 + 16);
                        }
                    }
                }
                {
                    float xM;
                    SWQ_BEST64(rbj[R - 1], xM, xc_p, S_p, nj);
                    hand_m = xM;
                }
                hand_d = ch;
            }
            carry_in = in_m;
        }
        if (!last && writes_bnd && on && LB > 0)   // the bottom row of the last column
            bnd[LB - 1] = (long long) ((unsigned long long) (uint32_t) __builtin_bit_cast(int, hand_m) |
                                       ((unsigned long long) (uint32_t) __builtin_bit_cast(int, hand_d) << 32));
        // best cell of the pair: highest score, then smallest i, then smallest j (sw.cpp:153-158 scans row-major with >)
        float best = 0.0f;
        uint32_t bi = 0xFFFFFFFFu, bj = 0xFFFFFFFFu;
        if (on) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned long long w = __builtin_bit_cast(unsigned long long, rbj[r]);
                const float rbr = __builtin_bit_cast(float, (uint32_t) (w >> 32));
                const uint32_t rjr = ~(uint32_t) w;
                const uint32_t ii = T ? rjr : i0 + r, jj = T ? i0 + r : rjr;
                if (rbr > best || (rbr == best && best > 0.0f && (ii < bi || (ii == bi && jj < bj)))) { best = rbr; bi = ii; bj = jj; }
            }
        }
        for (uint32_t dlt = 1; dlt < gl; ++dlt) {
            const int src = (lane + dlt) & 63;
            const float ob = __shfl(best, src, 64);
            const uint32_t oi = (uint32_t) __shfl((int) bi, src, 64), oj = (uint32_t) __shfl((int) bj, src, 64);
            if (st + dlt < gl) {
                if (ob > best || (ob == best && ob > 0.0f && (oi < bi || (oi == bi && oj < bj)))) { best = ob; bi = oi; bj = oj; }
            }
        }
        if (best > pbest || (best == pbest && best > 0.0f && (bi < pbi || (bi == pbi && bj < pbj)))) { pbest = best; pbi = bi; pbj = bj; }
        if (!last) {
            // the boundary words of this pass have left the wave before its next pass reads them (same CU, same L1); a later
            // segment is behind the workgroup barrier above
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        }   // passes
        if (active && st == 0) {
            float best = pbest;
            uint32_t bi = pbi, bj = pbj;
            if (seg) {                             // merge with the best of the earlier segments (same rule)
                const float ob = __builtin_bit_cast(float, __hip_atomic_load((int *) a.score + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const uint32_t oi = (uint32_t) __hip_atomic_load((int *) a.besti + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t oj = (uint32_t) __hip_atomic_load((int *) a.bestj + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (ob > best || (ob == best && ob > 0.0f && (oi < bi || (oi == bi && oj < bj)))) { best = ob; bi = oi; bj = oj; }
            }
            a.score[p] = best;
            a.besti[p] = bi;
            a.bestj[p] = bj;
        }
    }
    }
}

// One code object for every R: a group's item names its R and the workgroup branches (wave-uniform) to that instance, so
// that one launch per orientation covers all the groups of a call -- a launch per R would end in nine tails of
// one-workgroup-per-CU items.
template <bool T>
__global__ __launch_bounds__(64 * SWQ_NW) void k_sw_qp(swf_args a, const swq_item *items)
{
    extern __shared__ float4 qp4[];
    const swq_item it = items[blockIdx.x];
    // the clock this kernel HOLDS (it runs at the board's power limit): shader cycles against the 100 MHz reference
    // counter over the life of the workgroup, two atomics per workgroup (tens of thousands of cells each)
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    switch (it.R) {
    case 4: swq_group<T, 4>(a, it, qp4); break;
    case 5: swq_group<T, 5>(a, it, qp4); break;
    case 6: swq_group<T, 6>(a, it, qp4); break;
    case 7: swq_group<T, 7>(a, it, qp4); break;
    case 8: swq_group<T, 8>(a, it, qp4); break;
    case 9: swq_group<T, 9>(a, it, qp4); break;
    case 10: swq_group<T, 10>(a, it, qp4); break;
    case 11: swq_group<T, 11>(a, it, qp4); break;
    default: swq_group<T, 12>(a, it, qp4); break;
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");     // the trace masks sit in the scalar data cache
    if (a.clk && threadIdx.x == 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(a.clk, c1 - c0);
        atomicAdd(a.clk + 1, r1 - r0);
    }
}

// TraceBackBitSW sw.cpp:8-77.  One thread per pair; path chars are written backwards into
// paths[path_end[p]-1 ...]; path_start/path_len describe the result.  Pair classes (sorted order):
// 0 = k_sw_qp strips along A, 1 = k_sw_qp strips along B, 2 = k_sw_float<false>, 3 = k_sw_float<true>.
struct swf_classes { uint32_t first[5]; };

// One step of the walk examines ONE cell whatever the state (M: the cell it stands on, D / I: the reference's TB[i-1][j] /
// TB[i][j-1]) and needs the three "into M" bits (DM, IM, SM), the MD bit or the MI bit of it.  The body is branch-free over
// the state -- the lanes of a wave are in different states, and a branch per state put three dependent memory waits and
// three address computations into every iteration -- and the per-pair constants of the address (block, lane, strips per
// pass) are computed once.
__global__ void k_traceback(const uint8_t *tb, const uint64_t *tb_off, const uint32_t *ia, const uint32_t *a_len,
                            const uint32_t *ib, const uint32_t *b_len, swf_classes cl,
                            const float *score, const uint32_t *besti, const uint32_t *bestj, uint32_t npairs,
                            char *paths, const uint64_t *path_end, uint64_t *path_start, uint32_t *path_len,
                            uint32_t *lo_a, uint32_t *lo_b, const swq_item *qitems0, const swq_item *qitems1, const uint32_t *qp_item)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    path_len[p] = 0;
    path_start[p] = path_end[p];
    lo_a[p] = RSK_NO_POS;
    lo_b[p] = RSK_NO_POS;
    if (score[p] == 0.0f) return;                     // sw.cpp:200-201
#ifdef SWQ_EXPERIMENT_NO_TRACE_STORE
    return;                                           // (timing experiment: there is no trace to walk)
#endif
    const uint32_t cls = (p >= cl.first[1]) + (p >= cl.first[2]) + (p >= cl.first[3]);
    const bool rows_are_a = cls == 0 || cls == 2;     // strips along A: the strip row is i, the wave step is j
    const uint32_t LA = a_len[ia[p]], LB = b_len[ib[p]];
    const uint8_t *T;
    // k_sw_qp pairs (cls < 2): the trace lives in the blocks of the pair's workgroup item ([step - row + R - 1][row][4 masks], SWQ_COLB(R) per column)
    uint32_t q_R = SWQ_RMAX, q_colb = SWQ_COLB(SWQ_RMAX), q_lane0 = 0;
    size_t q_pass_stride = 0;
    // k_sw_float pairs: step-major block of the pair's wave item, T = its start + the pair's first lane; record (step, lane)
    // holds the 16 rows of the lane's strip, one byte per cell, the four rows of a dword in big-endian order
    uint32_t f_ld = 0, f_g = 1;
    float inv = 1.0f;                                 // 1 / strips per pass (per row group): quotients of small integers, exact below
    if (cls < 2) {
        const swq_item it = (cls == 0 ? qitems0 : qitems1)[qp_item[p]];
        const uint32_t pidx = p - it.first, nbatch = (it.count + SWQ_NPW - 1) / SWQ_NPW, b = pidx / SWQ_NPW;
        q_R = it.R;
        q_colb = SWQ_COLB(q_R);
        q_lane0 = (pidx - b * SWQ_NPW) * SWQ_GS;
        const size_t q_block = (size_t) it.ncol * q_colb;
        q_pass_stride = (size_t) nbatch * q_block;
        T = tb + it.tb_base + (size_t) b * q_block;
        inv = 1.0f / (float) q_R;
    } else {
        f_ld = qp_item[p];                            // steps per row group of the item
        f_g = min(max(1u, ((rows_are_a ? LA : LB) + SWF_R - 1) / SWF_R), 64u);
        T = tb + tb_off[p];
        inv = 1.0f / (float) f_g;
    }
    uint32_t i = besti[p] + 1, j = bestj[p] + 1;      // 1-based
    const uint32_t Besti = i, Bestj = j;
    uint64_t w = path_end[p];                          // the path is written backwards from here
    uint32_t state = 0;                               // 0 M, 1 D, 2 I
    uint32_t n = 0;
    // characters are collected four at a time and stored as one aligned dword (a byte store per step made
    // this kernel store-request bound); `acc` holds the `nacc` characters below address w - nacc
    uint32_t acc = 0, nacc = 0;
#define PATH_STORE() do { \
        if (nacc == 0 && (w & 3) != 0) paths[--w] = (char) ch;            /* head: up to 3 bytes down to a dword boundary */ \
        else { \
            acc = (acc << 8) | ch; \
            if (++nacc == 4) { w -= 4; *(uint32_t *) (paths + w) = acc; nacc = 0; } \
        } } while (0)
    if (cls < 2) {
        // k_sw_qp pairs.  The record of cell (strip row srow, wave step `step`) is
        //     block(pass) + (step + st + R - 1 - r) * colb + r * 32,   strip = srow / R, r = srow % R, pass = strip / 16, st = strip % 16,
        // and a step of the walk moves one row up and / or one wave step back: the address, r, st and the walker's mask bit (lane)
        // are carried along and UPDATED (r06) -- one division per pair instead of a reciprocal, two corrections, a division by 16
        // and two 64-bit multiply-adds per step (~100 instructions around one dependent 16-byte load; the kernel lasts as long
        // as its longest walk: 805 ns per step, r05).  Which way the next cell lies depends only on the state the walk enters:
        // M: (-1, -1), D: (-1, 0), I: (0, -1) in (i, j) -- whatever state it leaves (sw.cpp:33-70).
        const uint32_t ci0 = i - 1, cj0 = j - 1;
        const uint32_t srow0 = rows_are_a ? ci0 : cj0, step0 = rows_are_a ? cj0 : ci0;
        const uint32_t strip0 = srow0 / q_R;
        int r = (int) (srow0 - strip0 * q_R), st = (int) (strip0 % SWQ_GS), lane = (int) q_lane0 + st;
        const int R1 = (int) q_R - 1, colb = (int) q_colb;
        const uint8_t *rec = T + (size_t) (strip0 / SWQ_GS) * q_pass_stride + (size_t) (step0 + (uint32_t) st + q_R - 1 - (uint32_t) r) * q_colb + (uint32_t) r * 32u;
        for (;;) {
            const uint32_t ch = state == 0 ? 'M' : (state == 1 ? 'D' : 'I');
            ++n;
            // the cell's masks {SM | IM, SM | (DM & ~IM), MD, MI}: state M needs the first two, D / I the second two: one
            // 16-byte load, issued before this step's path store (the wait that follows must not cover a store).
            // (Measured and not kept: a walker in state M fetching the records of rows r .. r - 3 of its diagonal at once
            // and taking the next three steps from registers -- 2.15 vs 2.02 ms per 350 k-pair batch; r05: a request nobody waits
            // for to the 128-byte line before the current one -- 0.73-0.77 against 0.72-0.75 ms for 72,000 walks.)
            typedef unsigned tb_v4u __attribute__((ext_vector_type(4)));
            tb_v4u q;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(q) : "v"(rec + (state == 0 ? 0 : 16)) : "memory");
            PATH_STORE();
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q) :: "memory");
            const uint32_t b0 = (uint32_t) ((((uint64_t) q.y << 32) | q.x) >> lane) & 1u, b1 = (uint32_t) ((((uint64_t) q.w << 32) | q.z) >> lane) & 1u;
            uint32_t ns;
            if (state == 0) {
                if (b0 & b1) break;                   // SM; precedence as TraceBackBitSW reads its flags: stop, then I, then D
                ns = b0 ? 2u : (b1 ? 1u : 0u);
                --i; --j;
            } else if (state == 1) {
                ns = b0 ? 0u : 1u;                    // MD
                --i;
            } else {
                ns = b1 ? 0u : 2u;                    // MI
                --j;
            }
            state = ns;
            const bool di = ns != 2u, dj = ns != 1u;
            const bool drow = rows_are_a ? di : dj, dstp = rows_are_a ? dj : di;
            const bool cross = drow && r == 0, wrap = cross && st == 0;      // into the strip above / into the pass before
            const int dst = cross ? (wrap ? SWQ_GS - 1 : -1) : 0;
            const int dr = drow ? (cross ? R1 : -1) : 0;
            rec += (dst - dr - (int) dstp) * colb + dr * 32;
            if (wrap) rec -= q_pass_stride;
            r += dr; st += dst; lane += dst;
        }
    } else
    for (;;) {
        const uint32_t ch = state == 0 ? 'M' : (state == 1 ? 'D' : 'I');
        ++n;
        const uint32_t ci = state == 2 ? i : i - 1, cj = state == 1 ? j : j - 1;       // 0-based cell (sw.cpp:33-70)
        const uint32_t srow = rows_are_a ? ci : cj, step = rows_are_a ? cj : ci;
        uint32_t dm, im, sm, md, mi;                  // only the bits of the current state are meaningful
        {
            const uint32_t sa = srow / SWF_R, r = srow - sa * SWF_R;
            const uint32_t rg = (uint32_t) (((float) sa + 0.5f) * inv), st = sa - rg * f_g;
            const uint32_t bits = T[(((size_t) rg * f_ld + step + st) * 64 + st) * 16 + (r ^ 3u)];
            PATH_STORE();
            dm = (bits >> 4) & 1u; im = (bits >> 3) & 1u; sm = (bits >> 2) & 1u; md = (bits >> 1) & 1u; mi = bits & 1u;
        }
        if (state == 0) {
            if (sm) break;                            // precedence as TraceBackBitSW reads its flags: stop, then I, then D
            state = im ? 2u : (dm ? 1u : 0u);
            --i; --j;
        } else if (state == 1) {
            state = md ? 0u : 1u;
            --i;
        } else {
            state = mi ? 0u : 2u;
            --j;
        }
    }
    for (uint32_t k = nacc; k > 0; --k) paths[--w] = (char) ((acc >> (8 * (k - 1))) & 0xFFu);      // tail: oldest character first (highest address)
    path_start[p] = w;
    path_len[p] = n;
    const uint32_t leni = Besti - i + 1, lenj = Bestj - j + 1;
    lo_a[p] = Besti - leni;
    lo_b[p] = Bestj - lenj;
}

// GetLDDT_mu_fast lddt.cpp:63-124 over the M columns of a path (GetPosABs dssaligner.cpp:1282).
// One wave per pair.  The reference accumulates, over the column pairs c < c' within R0, (4, thresholds met)
// into both columns: integer counts, so any order of the additions gives the same per-column fractions.
#define LDDT_LDS_COLS 256
#define LDDT_SHORT_COLS 128
#define LDDT_LONG_COLS 4096
// <256, false> is the instance the library launches.  <128, true> (RSK_LDDT_SPLIT=1, an experiment kept for A/B runs: measured
// slower, see the launch) takes the pairs whose PATH has at most 128 characters with half the staging per wave; a wave whose pair
// belongs to the other instance returns after two loads.
template <int COLS, bool SHORT>
__global__ __launch_bounds__(256) void k_lddt(const char *paths, const uint64_t *path_start, const uint32_t *path_len,
                                              const uint32_t *lo_a, const uint32_t *lo_b, const uint32_t *ia, const uint32_t *ib,
                                              const uint32_t *a_off, const uint32_t *b_off,
                                              const float *ax, const float *ay, const float *az,
                                              const float *bx, const float *by, const float *bz,
                                              uint32_t npairs, uint32_t *scratch_pos, const uint64_t *scratch_off,
                                              float *frac_scratch, float *lddt_out, uint32_t *counts_out,
                                              const float *score, float min_fwd_score, const uint8_t *a_seq, const uint8_t *b_seq,
                                              uint32_t *long_cnt, uint32_t *long_list, int split)
{
    // per wave: coordinates of A and B at the aligned columns ({ax, ay, az, bx} / {by, bz}) and per-column counters
    // (considered | preserved << 16; at most 4 * 255 each)
    __shared__ float4 sc4[4][COLS];
    __shared__ float2 sc2[4][COLS];
    __shared__ uint32_t scnt[4][COLS];
    __shared__ uint32_t sq_cols[4][128];            // queue of column pairs within R0: (ci | cj << 16), squared distances
    __shared__ float2 sq_d[4][128];
    // the wave's pair: the same for its 64 lanes, which the compiler cannot see in `threadIdx.x >> 6` -- without the
    // readfirstlane every count, loop bound and queue length below would live in VGPRs and every loop test be a v_cmp
    const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (uint32_t) wv;
    if (p >= npairs) return;
    const int lane = threadIdx.x & 63;
    // CalcEvalue leaves everything unset below m_MinFwdScore (dssaligner.cpp:861): no LDDT needed for those pairs
    if (score[p] < min_fwd_score || score[p] == 0.0f) {
        if (!SHORT && lane == 0) { lddt_out[p] = 0.0f; counts_out[4 * p] = 0; counts_out[4 * p + 1] = 0; counts_out[4 * p + 2] = 0; counts_out[4 * p + 3] = RSK_NO_POS; }
        return;
    }
    const uint32_t len = path_len[p];
    if (split && (SHORT ? len > (uint32_t) LDDT_SHORT_COLS : len <= (uint32_t) LDDT_SHORT_COLS)) return;      // the other instance's pair
    uint32_t *posA = scratch_pos + 2 * scratch_off[p];
    uint32_t *posB = posA + (scratch_off[p + 1] - scratch_off[p]);
    float *frac = frac_scratch + scratch_off[p];
    // expand the path with wave-wide prefix counts (GetPosABs dssaligner.cpp:1282): 64 path characters per step
    const char *P = paths + path_start[p];
    const uint32_t la0 = lo_a[p], lb0 = lo_b[p];
    uint32_t nM = 0, nD = 0, nI = 0, nIdent = 0;
    // GetPctId dssaligner.cpp:1325: M columns whose two residue characters are equal -- counted while the path is walked
    // anyway (the host then needs neither chain's sequence for the pctid column)
    const uint8_t *SA = a_seq ? a_seq + a_off[ia[p]] : nullptr, *SB = b_seq ? b_seq + b_off[ib[p]] : nullptr;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float *AX = ax + a_off[ia[p]], *AY = ay + a_off[ia[p]], *AZ = az + a_off[ia[p]];
    const float *BX = bx + b_off[ib[p]], *BY = by + b_off[ib[p]], *BZ = bz + b_off[ib[p]];
    // A path of at most LDDT_LDS_COLS characters has at most that many M columns: their coordinates go straight into the
    // wave's staging while the path is expanded (r01-r04 wrote the column positions to scratch, fenced, and read them back
    // for the staging loop: two dependent global round trips per pair).  Longer paths keep the scratch lists, which
    // k_lddt_long reads.
    const bool direct = len <= (uint32_t) COLS;
    if (direct) {
        // every character of the path at once, then every coordinate at once: two global round trips per PAIR (r01-r05 took the
        // path 64 characters at a time, each step's coordinate loads behind its character load: two round trips per 64 characters)
        constexpr int NCH = COLS / 64;
        char ch[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const uint32_t c = (uint32_t) k * 64u + (uint32_t) lane;
            ch[k] = c < len ? P[c] : (char) 0;
        }
        uint32_t kMs[NCH], a1s[NCH], b1s[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const unsigned long long mM = __ballot(ch[k] == 'M'), mD = __ballot(ch[k] == 'D'), mI = __ballot(ch[k] == 'I');
            const uint32_t kM = nM + (uint32_t) __popcll(mM & lt), kD = nD + (uint32_t) __popcll(mD & lt), kI = nI + (uint32_t) __popcll(mI & lt);
            kMs[k] = kM; a1s[k] = la0 + kM + kD; b1s[k] = lb0 + kM + kI;
            nM += (uint32_t) __popcll(mM); nD += (uint32_t) __popcll(mD); nI += (uint32_t) __popcll(mI);
        }
        float4 v4[NCH];
        float2 v2[NCH];
        bool same[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            same[k] = false;
            if (ch[k] == 'M') {
                v4[k] = make_float4(AX[a1s[k]], BX[b1s[k]], AY[a1s[k]], BY[b1s[k]]);        // (A, B) pairs per axis: the two distances run as packed fp32 ops
                v2[k] = make_float2(AZ[a1s[k]], BZ[b1s[k]]);
                if (SA && SB) same[k] = SA[a1s[k]] == SB[b1s[k]];
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (ch[k] == 'M') { sc4[wv][kMs[k]] = v4[k]; sc2[wv][kMs[k]] = v2[k]; scnt[wv][kMs[k]] = 0; }
            nIdent += (uint32_t) __popcll(__ballot(same[k]));
        }
    } else
    for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t c = base + lane;
        const char ch = c < len ? P[c] : 0;
        const unsigned long long mM = __ballot(ch == 'M'), mD = __ballot(ch == 'D'), mI = __ballot(ch == 'I');
        bool same = false;
        if (ch == 'M') {
            const uint32_t kM = nM + (uint32_t) __popcll(mM & lt), kD = nD + (uint32_t) __popcll(mD & lt), kI = nI + (uint32_t) __popcll(mI & lt);
            const uint32_t a1 = la0 + kM + kD, b1 = lb0 + kM + kI;
            posA[kM] = a1;
            posB[kM] = b1;
            if (SA && SB) same = SA[a1] == SB[b1];
        }
        nIdent += (uint32_t) __popcll(__ballot(same));
        nM += (uint32_t) __popcll(mM); nD += (uint32_t) __popcll(mD); nI += (uint32_t) __popcll(mI);
    }
    const uint32_t ncols = nM;
    if (lane == 0) { counts_out[4 * p] = nM; counts_out[4 * p + 1] = nD; counts_out[4 * p + 2] = nI; counts_out[4 * p + 3] = (SA && SB) ? nIdent : RSK_NO_POS; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (ncols == 0) { if (lane == 0) lddt_out[p] = 0.0f; return; }
    const float R0sq = 15.0f * 15.0f;
    if (ncols > (uint32_t) COLS && ncols <= LDDT_LONG_COLS) {
        // k_lddt_long: a whole workgroup per pair, over the list this kernel appends to (class 0: <= 1024 columns, 1: <= 4096).
        // (r02-r04 launched one workgroup per CANDIDATE -- every pair whose shorter chain allows > 256 columns, tens of
        // thousands per batch of unrelated chains, nearly all of which returned at once: 2 ms per batch, 7 % of config 4.)
        if (lane == 0) {
            const uint32_t cls = ncols <= 1024 ? 0u : 1u;
            long_list[(size_t) cls * npairs + atomicAdd(&long_cnt[cls], 1u)] = p;
        }
        return;
    }
    const bool in_lds = ncols <= (uint32_t) COLS;
    if (in_lds) {
        // Every unordered column pair once: the ncols (ncols - 1) / 2 pairs, row-major (ci < cj), are cut into 64 equal
        // runs, one per lane; a pair within R0 adds (4, thresholds met) to the counters of BOTH columns (integers, so
        // the order of the additions is immaterial).  All lanes busy whatever ncols is, half the distance work.
        if (!direct) {
            for (uint32_t c = lane; c < ncols; c += 64) {
                const uint32_t a1 = posA[c], b1 = posB[c];
                sc4[wv][c] = make_float4(AX[a1], BX[b1], AY[a1], BY[b1]);
                sc2[wv][c] = make_float2(AZ[a1], BZ[b1]);
                scnt[wv][c] = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        const uint32_t C = ncols;
        // Every unordered column pair once, in TILES of 64 x 64 columns (r04).  A lane keeps ONE column of the tile's first
        // block in registers for the whole tile; inside a block the partner is the column s places further on (cyclically:
        // s = 1 .. n / 2 covers every pair of the block's n columns once), between two blocks the partner is the same column
        // for all lanes (a broadcast read).  No per-lane bookkeeping, no divergent reload of the held column, two LDS reads per
        // test.  (r01-r03 cut the row-major list of pairs into 64 equal runs: every lane walked its own (ci, cj) and reloaded
        // ci's coordinates whenever its row changed -- on most steps some lane did -- which doubled the LDS reads of a step.)
        // Two phases per step so that the expensive part runs on full waves: every lane tests one column pair (two squared
        // distances); the ~10 % that lie within R0 are compacted into a per-wave LDS queue (ballot + prefix count), and
        // whenever 64 are waiting each lane finishes one of them (two correctly rounded square roots, the four
        // thresholds, the counters of both columns).
        uint32_t qn = 0;                                          // entries waiting in the queue (wave-uniform)
        auto drain = [&](uint32_t n) {                            // the first n entries, one per lane
            if ((uint32_t) lane < n) {
                const uint32_t cc = sq_cols[wv][lane];
                const float2 dd = sq_d[wv][lane];
                const float d1 = sqrtf(dd.x), d2 = sqrtf(dd.y);
                const float diff = fabsf(d1 - d2);
                const uint32_t inc = 4u | (((diff <= 0.5f) + (diff <= 1.0f) + (diff <= 2.0f) + (diff <= 4.0f)) << 16);
                atomicAdd(&scnt[wv][cc & 0xFFFFu], inc);
                atomicAdd(&scnt[wv][cc >> 16], inc);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            // the entries behind the first 64 move to the front
            if (n == 64 && qn > 64) {
                const uint32_t rest = qn - 64;
                uint32_t cc = 0;
                float2 dd = make_float2(0, 0);
                if ((uint32_t) lane < rest) { cc = sq_cols[wv][64 + lane]; dd = sq_d[wv][64 + lane]; }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if ((uint32_t) lane < rest) { sq_cols[wv][lane] = cc; sq_d[wv][lane] = dd; }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            qn -= n;
        };
        // one test: column `mine` (this lane's, coordinates in p4 / p2) against column `other`; `live` = the lanes that take part.
        // Every lane computes (a lane without a column holds coordinates far outside any structure, `other` is always a
        // staged column): the two comparisons write their lane masks straight to SGPRs, the rest of the decision is scalar
        // (and with `live`, branch on the result, queue length), the queue slot of a hit is two v_mbcnt on the mask.
        // (r04a: `hit` was a per-lane flag set inside `if (live)`, which the compiler turned back into a mask with two more VALU
        // ops, and the slot was and / and / bcnt / bcnt / add: 22 -> 15 VALU instructions per test.)
        auto test = [&](unsigned long long live, uint32_t mine, uint32_t other, const float4 &p4, const float2 &p2) {
            const float4 q4 = sc4[wv][other];
            const float2 q2 = sc2[wv][other];
            // (x1-x2)^2 == (x2-x1)^2 exactly, so the reference's (lower column) - (higher column) order is immaterial.
            // pdbchain.cpp:320-335: dx*dx + dy*dy + dz*dz, every product and sum rounded separately (-ffp-contract=off);
            // lane-wise packed ops (v_pk_add_f32 / v_pk_mul_f32) do the A-side and the B-side distance at once
            swq_v2f X = swq_v2f{ p4.x, p4.y } - swq_v2f{ q4.x, q4.y };
            swq_v2f Y = swq_v2f{ p4.z, p4.w } - swq_v2f{ q4.z, q4.w };
            swq_v2f Z = swq_v2f{ p2.x, p2.y } - swq_v2f{ q2.x, q2.y };
            swq_v2f D = X * X;
            D += Y * Y;
            D += Z * Z;
            // lddt.cpp:88: skipped when both squared distances exceed R0^2
            const unsigned long long m = (__builtin_amdgcn_ballot_w64(!(D.x > R0sq)) | __builtin_amdgcn_ballot_w64(!(D.y > R0sq))) & live;
            if (m) {
                if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, qn));
                    sq_cols[wv][slot] = mine | (other << 16);
                    sq_d[wv][slot] = make_float2(D.x, D.y);
                }
                qn += (uint32_t) __builtin_popcountll(m);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (qn >= 64) drain(64);
            }
        };
        const uint32_t nblk = (C + 63) / 64;
        for (uint32_t bi = 0; bi < nblk; ++bi) {
            const uint32_t base_i = bi * 64, n = min(64u, C - base_i);      // this block's columns: base_i .. base_i + n - 1
            const bool have = (uint32_t) lane < n;
            const unsigned long long have_m = n == 64 ? ~0ull : (1ull << n) - 1ull;
            const uint32_t mine = base_i + (uint32_t) lane;
            float4 p4 = make_float4(1e15f, 1e15f, 1e15f, 1e15f);             // no column: 1e30 from everything
            float2 p2 = make_float2(1e15f, 1e15f);
            if (have) { p4 = sc4[wv][mine]; p2 = sc2[wv][mine]; }
            // pairs inside the block: partner = (lane + s) mod n; for even n the step s = n / 2 meets every pair from both ends
            // (only its lanes < s take part)
            for (uint32_t sft = 1; 2 * sft <= n; ++sft) {
                uint32_t o = (uint32_t) lane + sft;
                if (o >= n) o -= n;
                if (!have) o = 0;
                test(2 * sft == n ? (1ull << sft) - 1ull : have_m, mine, base_i + o, p4, p2);
            }
            // pairs with the columns of the later blocks: one partner column per step, the same for every lane
            for (uint32_t cj = base_i + 64; cj < C; ++cj) test(have_m, mine, cj, p4, p2);
        }
        if (qn) drain(qn);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // per-column fractions in registers (column c in lane c % 64, slot c / 64), then the reference's sequential
        // sum in column order (lddt.cpp:111-121): one v_readlane + v_add per column instead of a dependent LDS read
        float fr[COLS / 64];
#pragma unroll
        for (int k = 0; k < COLS / 64; ++k) {
            fr[k] = 0.0f;
            if ((uint32_t) k * 64 >= ncols) continue;                    // (wave-uniform: no division for slots beyond the alignment)
            const uint32_t c = (uint32_t) k * 64 + lane;
            const uint32_t v = c < ncols ? scnt[wv][c] : 0u, cons = v & 0xFFFFu, pres = v >> 16;
            fr[k] = cons > 0 ? (float) pres / (float) cons : 0.0f;
        }
        float total = 0.0f;
#pragma unroll
        for (int k = 0; k < COLS / 64; ++k) {
            const uint32_t n = ncols > (uint32_t) k * 64 ? min(64u, ncols - (uint32_t) k * 64) : 0u;
            for (uint32_t c = 0; c < n; ++c)
                total += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fr[k]), (int) c));
        }
        if (lane == 0) lddt_out[p] = total / (float) ncols;
        return;
    } else
    for (uint32_t ci = lane; ci < ncols; ci += 64) {
        // alignments longer than the LDS staging: a lane owns whole columns, coordinates from HBM
        const uint32_t a1 = posA[ci], b1 = posB[ci];
        const float x1 = AX[a1], y1 = AY[a1], z1 = AZ[a1], u1 = BX[b1], v1 = BY[b1], w1 = BZ[b1];
        uint32_t cons = 0, pres = 0;
        for (uint32_t cj = 0; cj < ncols; ++cj) {
            if (cj == ci) continue;
            const uint32_t a2 = posA[cj], b2 = posB[cj];
            const float x2 = AX[a2], y2 = AY[a2], z2 = AZ[a2], u2 = BX[b2], v2 = BY[b2], w2 = BZ[b2];
            const float dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;
            const float ex = u1 - u2, ey = v1 - v2, ez = w1 - w2;
            float d1s = dx * dx; d1s += dy * dy; d1s += dz * dz;
            float d2s = ex * ex; d2s += ey * ey; d2s += ez * ez;
            if (d1s > R0sq && d2s > R0sq) continue;
            const float d1 = sqrtf(d1s), d2 = sqrtf(d2s);
            const float diff = fabsf(d1 - d2);
            pres += (diff <= 0.5f) + (diff <= 1.0f) + (diff <= 2.0f) + (diff <= 4.0f);
            cons += 4;
        }
        frac[ci] = cons > 0 ? (float) pres / (float) cons : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane == 0) {
        float total = 0.0f;
        for (uint32_t c = 0; c < ncols; ++c) total += frac[c];      // sequential, column order (lddt.cpp:111-121)
        lddt_out[p] = total / (float) ncols;
    }
}


// Alignments of LDDT_LDS_COLS < columns <= LDDT_LONG_COLS (long homologous chains): the same unordered-pair scheme with a
// whole workgroup per pair (k_lddt has expanded the path into scratch_pos and left the column count in counts_out).
// Launched twice over host-built candidate lists: up to 1024 columns (32 KB of LDS) and up to 4096 (128 KB).
__global__ __launch_bounds__(256) void k_lddt_long(const uint32_t *list, const uint32_t *nlist_dev, uint32_t cap, const uint32_t *ia, const uint32_t *ib,
                                                   const uint32_t *a_off, const uint32_t *b_off,
                                                   const float *ax, const float *ay, const float *az,
                                                   const float *bx, const float *by, const float *bz,
                                                   const uint32_t *scratch_pos, const uint64_t *scratch_off,
                                                   float *lddt_out, const uint32_t *counts_out, const float *score, float min_fwd_score)
{
    extern __shared__ float4 lddt_smem[];                               // cap columns: float4, float2, counter, fraction
    float4 *c4 = lddt_smem;
    float2 *c2 = (float2 *) (c4 + cap);
    uint32_t *cnt = (uint32_t *) (c2 + cap);
    float *frs = (float *) (cnt + cap);
    const uint32_t nlist = *nlist_dev;                                  // pairs k_lddt listed for this column class
    const int tid = threadIdx.x;
    for (uint32_t li = blockIdx.x; li < nlist; li += gridDim.x) {
    __syncthreads();                                                    // the previous pair's staging is done with
    const uint32_t p = list[li];
    const uint32_t C = counts_out[4 * p];
    const uint32_t *posA = scratch_pos + 2 * scratch_off[p];
    const uint32_t *posB = posA + (scratch_off[p + 1] - scratch_off[p]);
    const float *AX = ax + a_off[ia[p]], *AY = ay + a_off[ia[p]], *AZ = az + a_off[ia[p]];
    const float *BX = bx + b_off[ib[p]], *BY = by + b_off[ib[p]], *BZ = bz + b_off[ib[p]];
    for (uint32_t c = tid; c < C; c += 256) {
        const uint32_t a1 = posA[c], b1 = posB[c];
        c4[c] = make_float4(AX[a1], AY[a1], AZ[a1], BX[b1]);
        c2[c] = make_float2(BY[b1], BZ[b1]);
        cnt[c] = 0;
    }
    __syncthreads();
    const float R0sq = 15.0f * 15.0f;
    const uint32_t npair = C * (C - 1) / 2, per = (npair + 255) / 256;
    const uint32_t t0 = (uint32_t) tid * per, t1 = min(npair, t0 + per);
    if (t0 < t1) {
        const float twoc = (float) (2 * C - 1);
        uint32_t ci = (uint32_t) ((twoc - sqrtf(fmaxf(twoc * twoc - 8.0f * (float) t0, 0.0f))) * 0.5f);
        ci = min(ci, C - 2);
        while (ci > 0 && ci * (2 * C - ci - 1) / 2 > t0) --ci;
        while ((ci + 1) * (2 * C - ci - 2) / 2 <= t0) ++ci;
        uint32_t cj = ci + 1 + (t0 - ci * (2 * C - ci - 1) / 2);
        float4 p4 = c4[ci];
        float2 p2 = c2[ci];
        uint32_t own = 0;
        for (uint32_t t = t0; t < t1; ++t) {
            const float4 q4 = c4[cj];
            const float2 q2 = c2[cj];
            const float dx = p4.x - q4.x, dy = p4.y - q4.y, dz = p4.z - q4.z;
            const float ex = p4.w - q4.w, ey = p2.x - q2.x, ez = p2.y - q2.y;
            float d1s = dx * dx; d1s += dy * dy; d1s += dz * dz;
            float d2s = ex * ex; d2s += ey * ey; d2s += ez * ez;
            if (!(d1s > R0sq && d2s > R0sq)) {
                const float d1 = sqrtf(d1s), d2 = sqrtf(d2s);
                const float diff = fabsf(d1 - d2);
                const uint32_t inc = 4u | (((diff <= 0.5f) + (diff <= 1.0f) + (diff <= 2.0f) + (diff <= 4.0f)) << 16);
                own += inc;
                atomicAdd(&cnt[cj], inc);
            }
            if (++cj == C) {
                if (own) atomicAdd(&cnt[ci], own);
                own = 0;
                ++ci;
                cj = ci + 1;
                if (ci < C - 1) { p4 = c4[ci]; p2 = c2[ci]; }
            }
        }
        if (own) atomicAdd(&cnt[ci], own);
    }
    __syncthreads();
    for (uint32_t c = tid; c < C; c += 256) {
        const uint32_t v = cnt[c], cons = v & 0xFFFFu, pres = v >> 16;
        frs[c] = cons > 0 ? (float) pres / (float) cons : 0.0f;
    }
    __syncthreads();
    if (tid == 0) {
        float total = 0.0f;
        for (uint32_t c = 0; c < C; ++c) total += frs[c];      // sequential, column order (lddt.cpp:111-121)
        lddt_out[p] = total / (float) C;
    }
    }
}

// caller-order path packing: out_len[p] = path_len[slot[p]] + 1 is scanned on the device, then one
// wave per pair copies its path (NUL-terminated) to its final offset.
__global__ void k_path_sizes(const uint32_t *slot, const uint32_t *path_len, uint32_t npairs, uint64_t *sizes)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npairs) sizes[p] = (uint64_t) path_len[slot[p]] + 1;
}
__global__ void k_path_pack(const uint32_t *slot, const uint32_t *path_len, const uint64_t *path_start, const char *paths,
                            const uint64_t *out_off, uint32_t npairs, char *out)
{
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // wave-uniform
    if (p >= npairs) return;
    const uint32_t k = slot[p], len = path_len[k];
    const char *src = paths + path_start[k];
    char *dst = out + out_off[p];
    for (uint32_t c = threadIdx.x & 63; c < len; c += 64) dst[c] = src[c];
    if ((threadIdx.x & 63) == 0) dst[len] = 0;
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static double swf_pvalue(double ts)   // StatSig::GetPvalue statsig.cpp:27-44
{
    const double l = (ts < 0.11) ? (-80.0 * ts + -0.58) : (-52.0 * ts + -3.7);
    double p = pow(10, l);
    if (p > 1) p = 1;
    return p;
}
static double swf_qual(double ts)     // StatSig::GetQual statsig.h:8-25
{
    const double logE = 5.0 + -40.0 * ts;
    if (logE < -20) return 1;
    const double x = pow(10, logE / 10);
    return 1 / (1 + x / 2);
}

extern "C" size_t rsk_align_paths_bytes(const rsk_db *a, const rsk_db *b, const uint32_t *ia, const uint32_t *ib, size_t n)
{
    if (!a || !b || (n && (!ia || !ib))) return 0;
    size_t tot = 0;
    for (size_t p = 0; p < n; ++p) {
        if (ia[p] >= a->n || ib[p] >= b->n) return 0;
        tot += (size_t) a->len[ia[p]] + b->len[ib[p]] + 1;
    }
    return tot;
}

namespace {
struct swf_timer {
    const void *who = nullptr;
    bool on;
    std::chrono::steady_clock::time_point t0;
    swf_timer() : on(getenv("RSK_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[rsk_align_pairs] %-18s %8.3f ms   @%.1f ctx %p\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t1.time_since_epoch()).count() - 1e5 * floor(std::chrono::duration<double>(t1.time_since_epoch()).count() / 100), who);
        t0 = t1;
    }
};

// host -> device staging of the small per-call arrays: one pinned blob, one copy
struct swf_blob {
    std::vector<std::pair<size_t, size_t>> sect;   // (offset, bytes)
    size_t bytes = 0;
    size_t add(size_t n) { const size_t o = bytes; sect.push_back({ o, n }); bytes = (bytes + n + 255) & ~(size_t) 255; return o; }
};
}   // namespace

extern "C" int rsk_align_pairs(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, const uint32_t *ia, const uint32_t *ib,
                               size_t npairs, float gap_open, float gap_ext, float min_fwd_score, rsk_aln *out, char *paths,
                               size_t paths_bytes)
{
    if (!ctx || !dba || !dbb || (npairs && (!ia || !ib || !out))) { rsk_set_error("rsk_align_pairs: NULL argument"); return RSK_E_INVALID; }
    if (!dba->d_prof || !dbb->d_prof) { rsk_set_error("rsk_align_pairs: chain set has no profiles"); return RSK_E_INVALID; }
    if (gap_open > 0 || gap_ext > 0) { rsk_set_error("rsk_align_pairs: gap penalties must be <= 0 (sw.cpp:86-87)"); return RSK_E_INVALID; }
    if (npairs == 0) return RSK_OK;
    if (npairs > 0x7FFFFFFFull) { rsk_set_error("rsk_align_pairs: too many pairs in one call"); return RSK_E_RANGE; }
    const bool want_stats = dba->d_x && dbb->d_x;
    swf_timer tm;
    tm.who = ctx;
    size_t need = 0;
    for (size_t p = 0; p < npairs; ++p) {
        if (ia[p] >= dba->n || ib[p] >= dbb->n) { rsk_set_error("rsk_align_pairs: pair %zu out of range", p); return RSK_E_INVALID; }
        need += (size_t) dba->len[ia[p]] + dbb->len[ib[p]] + 1;
    }
    if (paths && paths_bytes < need) { rsk_set_error("rsk_align_pairs: paths buffer too small (%zu < %zu)", paths_bytes, need); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = swf_upload_tables(ctx);
    if (rc != RSK_OK) return rc;

    // ---- classes: 0/1 = query-profile kernel (strips along A / along B), 2/3 = per-pair kernel --------
    // A pair goes to the query-profile kernel when one of its chains fits the LDS profile and enough
    // pairs of this call share that chain to fill a wave; everything else takes the per-pair kernel.
    // A workgroup of k_sw_qp holds ONE profile (its LDS), so its parallelism is (pairs of the group) x (strips of the
    // chain) lanes; below ~10 waves' worth the per-pair kernel, which fills the CU with unrelated pairs, is faster
    // (measured on the SCOP40 -sensitive survivors: crossover at 30-50 pairs per 175-residue query).
    // A group takes the query-profile kernel when its pairs x strips amount to >= min_lanes lanes.  Measured (r02c): after the
    // per-pair kernel's (strips, steps) ordering the profile kernel only wins for groups of hundreds of pairs (one
    // workgroup per CU idles while its longest pair finishes), e.g. a query against a database; all-vs-all survivor
    // groups (tens of pairs) stay with the per-pair kernel.
    const uint32_t min_lanes = getenv("RSK_SWQ_MIN_LANES") ? (uint32_t) atoi(getenv("RSK_SWQ_MIN_LANES")) : 4096;
    auto qp_bucket = [&](uint32_t count, uint32_t L) -> int {
        if (L == 0) return -1;
        return (uint64_t) count * SWQ_GS >= min_lanes ? 0 : -1;
    };
    auto qp_ok = [&](uint32_t count, uint32_t L) { return qp_bucket(count, L) >= 0; };
    std::vector<uint32_t> cntA(dba->n, 0), cntB;
    for (size_t p = 0; p < npairs; ++p) ++cntA[ia[p]];
    bool anyB = false;
    for (size_t p = 0; p < npairs && !anyB; ++p) anyB = !qp_ok(cntA[ia[p]], dba->len[ia[p]]);
    if (anyB) {
        cntB.assign(dbb->n, 0);
        for (size_t p = 0; p < npairs; ++p)
            if (!qp_ok(cntA[ia[p]], dba->len[ia[p]])) ++cntB[ib[p]];
    }
    struct keyed { uint64_t key; uint32_t idx; };
    std::vector<keyed> ord(npairs);
    rsk_parallel_for(npairs, 65536, [&](size_t p_lo, size_t p_hi) {
    for (size_t p = p_lo; p < p_hi; ++p) {
        const uint32_t LA = dba->len[ia[p]], LB = dbb->len[ib[p]];
        uint64_t key;
        if (qp_ok(cntA[ia[p]], LA))
            key = ((uint64_t) 0 << 62) | ((uint64_t) ia[p] << 24) | (0xFFFFFFu - std::min(LB, 0xFFFFFFu));
        else if (!cntB.empty() && qp_ok(cntB[ib[p]], LB))
            key = ((uint64_t) 1 << 62) | ((uint64_t) ib[p] << 24) | (0xFFFFFFu - std::min(LA, 0xFFFFFFu));
        // per-pair kernel: the pairs of a wave share the strip count g and run until the longest step chain among them is
        // done, so the order is (strips desc, steps desc): lane-slot efficiency 0.53 -> 0.84 on SCOP40 lengths compared
        // with ordering by the strip chain's length alone
        else if (!(LA > 64 * SWF_R && LB <= 64 * SWF_R))
            key = ((uint64_t) 2 << 62) | ((uint64_t) (0xFFFFFu - std::min((LA + SWF_R - 1) / SWF_R, 0xFFFFFu)) << 40) |
                  ((uint64_t) (0xFFFFFFu - std::min(LB, 0xFFFFFFu)) << 16);
        else
            key = ((uint64_t) 3 << 62) | ((uint64_t) (0xFFFFFu - std::min((LB + SWF_R - 1) / SWF_R, 0xFFFFFu)) << 40) |
                  ((uint64_t) (0xFFFFFFu - std::min(LA, 0xFFFFFFu)) << 16);
        ord[p] = keyed{ key, (uint32_t) p };
    }
    });
    {
        // slices sorted on the host worker threads, then merged pairwise (keys are unique: (key, idx) is a total order)
        const auto less = [](const keyed &x, const keyed &y) { return x.key != y.key ? x.key < y.key : x.idx < y.idx; };
        const unsigned T = (unsigned) std::min<size_t>(reseek_amd::HostThreads(64), npairs / 65536 + 1);
        std::vector<size_t> cut(T + 1);
        for (unsigned t = 0; t <= T; ++t) cut[t] = npairs * t / T;
        rsk_parallel_for(T, 1, [&](size_t lo, size_t hi) { for (size_t t = lo; t < hi; ++t) std::sort(ord.begin() + cut[t], ord.begin() + cut[t + 1], less); });
        for (unsigned w = 1; w < T; w *= 2) {
            const unsigned nm = (T + 2 * w - 1) / (2 * w);
            rsk_parallel_for(nm, 1, [&](size_t lo, size_t hi) {
                for (size_t m = lo; m < hi; ++m) {
                    const unsigned a = (unsigned) m * 2 * w, b = std::min(T, a + w), c = std::min(T, a + 2 * w);
                    if (b < c) std::inplace_merge(ord.begin() + cut[a], ord.begin() + cut[b], ord.begin() + cut[c], less);
                }
            });
        }
    }
    swf_classes cl;
    {
        uint32_t cnt[4] = { 0, 0, 0, 0 };
        for (size_t k = 0; k < npairs; ++k) ++cnt[ord[k].key >> 62];
        cl.first[0] = 0;
        for (int c = 0; c < 4; ++c) cl.first[c + 1] = cl.first[c] + cnt[c];
    }
    tm.lap("classify+sort");

    // ---- per-pair offsets and work items ------------------------------------------------------------
    swf_blob hb;
    const size_t o_ia = hb.add(npairs * 4), o_ib = hb.add(npairs * 4), o_slot = hb.add(npairs * 4);
    const size_t o_tboff = hb.add((npairs + 1) * 8), o_pend = hb.add(npairs * 8), o_scoff = hb.add((npairs + 1) * 8);
    const size_t o_bndoff = hb.add((npairs + 1) * 8), o_qpitem = hb.add(npairs * 4);
    std::vector<swq_item> qitems[2];
    std::vector<swf_item> items;
    uint32_t nitems_normal = 0;
    // items of the query-profile kernel: a group is cut into chunks of a few wave passes per workgroup
    uint32_t swq_rmax = SWQ_RMAX;
    if (const char *e = getenv("RSK_SWQ_MAXR")) swq_rmax = std::min<uint32_t>(SWQ_RMAX, std::max<uint32_t>(SWQ_RMIN, (uint32_t) atoi(e)));      // tests: more passes / segments
    for (int c = 0; c < 2; ++c) {
        const rsk_db *sdb = c == 0 ? dba : dbb;
        size_t k = cl.first[c];
        const size_t end = cl.first[c + 1];
        while (k < end) {
            const uint32_t chain = (uint32_t) ((ord[k].key >> 24) & 0xFFFFFFFFu);
            size_t e = k;
            while (e < end && (uint32_t) ((ord[e].key >> 24) & 0xFFFFFFFFu) == chain) ++e;
            // rows per lane (swq_item::R): a pair takes 16 lanes whatever its chain, a chain of L residues runs in P passes of
            // 16 strips of R = ceil(L / 16P) rows, R <= 12.  A step of a pass costs ~38 + 20 R VALU instructions (ISA of the hot
            // loop), a step of a chain done in several passes ~5 % more (boundary word load + store): the P of least cost,
            // which is the smallest possible one except just above a multiple of 192 residues.
            const uint32_t Ls = sdb->len[chain];
            const uint32_t rmax = swq_rmax;
            uint32_t R = rmax;
            {
                const uint32_t Pmin = std::max<uint32_t>(1, (Ls + SWQ_GS * rmax - 1) / (SWQ_GS * rmax));
                double best_cost = 0;
                for (uint32_t P = Pmin; P <= Pmin + 2; ++P) {
                    const uint32_t cand = std::max<uint32_t>(SWQ_RMIN, (Ls + SWQ_GS * P - 1) / (SWQ_GS * P));
                    const uint32_t passes = ((Ls + cand - 1) / cand + SWQ_GS - 1) / SWQ_GS;
                    const double cost = (double) passes * (38.0 + 20.0 * cand) * (passes > 1 ? 1.05 : 1.0);
                    if (P == Pmin || cost < best_cost) { best_cost = cost; R = cand; }
                }
            }
            const size_t chunk = (size_t) SWQ_NPW * SWQ_NW * 2;
            for (size_t s = k; s < e; s += chunk) qitems[c].push_back(swq_item{ (uint32_t) s, (uint32_t) std::min(chunk, e - s), 0, R, 0 });
            k = e;
        }
        // longest-running workgroups first
        std::stable_sort(qitems[c].begin(), qitems[c].end(), [&](const swq_item &x, const swq_item &y) {
            const uint32_t cx = c == 0 ? ia[ord[x.first].idx] : ib[ord[x.first].idx], cy = c == 0 ? ia[ord[y.first].idx] : ib[ord[y.first].idx];
            return (uint64_t) x.count * sdb->len[cx] > (uint64_t) y.count * sdb->len[cy];
        });
    }
    for (int c = 2; c < 4; ++c) {
        const bool tr = c == 3;
        for (size_t k = cl.first[c]; k < cl.first[c + 1];) {
            const uint32_t p = ord[k].idx;
            uint32_t g = ((tr ? dbb->len[ib[p]] : dba->len[ia[p]]) + SWF_R - 1) / SWF_R;
            if (g == 0) g = 1;
            uint32_t ngroups = 1;
            if (g > 64) { ngroups = (g + 63) / 64; g = 64; }
            const uint32_t cnt = (uint32_t) std::min<size_t>(64 / g, cl.first[c + 1] - k);
            uint32_t lmax = 0;                                   // longest step chain of the item
            for (uint32_t q = 0; q < cnt; ++q) {
                const uint32_t pq = ord[k + q].idx;
                lmax = std::max(lmax, tr ? dba->len[ia[pq]] : dbb->len[ib[pq]]);
            }
            items.push_back(swf_item{ (uint32_t) k, cnt, g, ngroups, lmax + g - 1, 0, 0 });
            if (!tr) ++nitems_normal;
            k += cnt;
        }
    }
    const size_t o_items = hb.add(items.size() * sizeof(swf_item) + 16);
    const size_t o_q[2] = { hb.add(qitems[0].size() * sizeof(swq_item) + 16), hb.add(qitems[1].size() * sizeof(swq_item) + 16) };
    // pairs (sorted order) whose alignment can exceed the per-wave LDDT staging: min(LA, LB) bounds the aligned columns
    // (host bound only -- how many pairs CAN have that many columns; the pairs that do are listed by k_lddt on the device)
    std::vector<uint32_t> lddt_list[2];
    if (want_stats)
        for (size_t k = 0; k < npairs; ++k) {
            const uint32_t m = std::min(dba->len[ia[ord[k].idx]], dbb->len[ib[ord[k].idx]]);
            if (m > LDDT_LDS_COLS) lddt_list[m <= 1024 ? 0 : 1].push_back((uint32_t) k);
        }
    void *hpin = nullptr;
    if ((rc = rsk_pinned(ctx, 0, hb.bytes, &hpin)) != RSK_OK) return rc;
    char *H = (char *) hpin;
    uint32_t *sia = (uint32_t *) (H + o_ia), *sib = (uint32_t *) (H + o_ib), *slot = (uint32_t *) (H + o_slot);
    uint64_t *tb_off = (uint64_t *) (H + o_tboff), *path_end = (uint64_t *) (H + o_pend), *sc_off = (uint64_t *) (H + o_scoff);
    uint64_t *bnd_off = (uint64_t *) (H + o_bndoff);
    uint32_t *qp_item = (uint32_t *) (H + o_qpitem);
    uint64_t tbo = 0, pe = 0, so = 0, bno = 0, cells = 0;
    // trace blocks of the query-profile items: one block of ncol columns per (segment, wave batch), see swq_item
    for (int c = 0; c < 2; ++c)
        for (size_t q = 0; q < qitems[c].size(); ++q) {
            swq_item &it = qitems[c][q];
            const uint32_t p0 = ord[it.first].idx;
            const uint32_t Ls = c == 0 ? dba->len[ia[p0]] : dbb->len[ib[p0]];               // strip chain of the group
            uint32_t lmax = 0;
            for (uint32_t k = 0; k < it.count; ++k) {
                const uint32_t p = ord[it.first + k].idx;
                lmax = std::max(lmax, c == 0 ? dbb->len[ib[p]] : dba->len[ia[p]]);          // step chains
                qp_item[it.first + k] = (uint32_t) q;
            }
            const uint32_t gtot = (Ls + it.R - 1) / it.R;
            const uint64_t nblocks = (uint64_t) ((gtot + SWQ_GS - 1) / SWQ_GS) * ((it.count + SWQ_NPW - 1) / SWQ_NPW);      // passes x wave batches
            // column index = step + strip-in-pass.  A block is written by ONE wave through its CU's scalar cache: blocks
            // start on 128-byte lines and span whole lines (ncol a multiple of 8, column bytes a multiple of 16), so no cache
            // line is ever shared between the scalar caches of two CUs
            it.ncol = (lmax + SWQ_GS + it.R - 1 + 7) & ~7u;          // + the R - 1 columns of the diagonal order
            tbo = (tbo + 127) & ~(uint64_t) 127;
            it.tb_base = tbo;
            tbo += nblocks * it.ncol * SWQ_COLB(it.R);
        }
    // trace blocks of the per-pair items (swf_item): ngroups x ncol steps of 1 KB; a pair's trace offset = its item's block
    // + its first lane's record, its entry of qp_item = the item's steps per row group (k_traceback's decode)
    for (swf_item &it : items) {
        tbo = (tbo + 127) & ~(uint64_t) 127;
        it.tb_base = tbo;
        tbo += (uint64_t) it.ngroups * it.ncol * 1024;
        for (uint32_t q = 0; q < it.count; ++q) {
            tb_off[it.first + q] = it.tb_base + (uint64_t) q * it.g * 16;
            qp_item[it.first + q] = it.ncol;
        }
    }
    for (size_t k = 0; k < npairs; ++k) {
        const uint32_t p = ord[k].idx;
        const uint32_t c = (uint32_t) (ord[k].key >> 62);
        sia[k] = ia[p]; sib[k] = ib[p];
        slot[p] = (uint32_t) k;
        const uint32_t LA = dba->len[ia[p]], LB = dbb->len[ib[p]];
        bnd_off[k] = bno;
        if (c < 2) tb_off[k] = 0;                                // the item's blocks (swq_item::tb_base)
        if (c == 0) {                                           // trace: the item's blocks (above)
            if (LA > SWQ_GS * qitems[0][qp_item[k]].R) bno += 4 * (uint64_t) LB;       // several passes: two rows of 2 words per step
        } else if (c == 1) {
            if (LB > SWQ_GS * qitems[1][qp_item[k]].R) bno += 4 * (uint64_t) LA;
        }
        else if (c == 2) {
            if (LA > 64 * SWF_R) bno += 2 * (uint64_t) LB;      // multi-group pair: 2 words per step
        }
        pe += (uint64_t) LA + LB + 1;
        path_end[k] = pe;
        sc_off[k] = so;
        so += std::min(LA, LB);
        cells += (uint64_t) LA * LB;
    }
    tb_off[npairs] = tbo;
    sc_off[npairs] = so;
    bnd_off[npairs] = bno;
    memcpy(H + o_items, items.data(), items.size() * sizeof(swf_item));
    for (int c = 0; c < 2; ++c) memcpy(H + o_q[c], qitems[c].data(), qitems[c].size() * sizeof(swq_item));
    if (tm.on) {
        uint64_t cc[4] = { 0, 0, 0, 0 };
        for (size_t k = 0; k < npairs; ++k) cc[ord[k].key >> 62] += (uint64_t) dba->len[ia[ord[k].idx]] * dbb->len[ib[ord[k].idx]];
        fprintf(stderr, "[rsk_align_pairs] classes (pairs / Gcells): qp %u / %.2f, qp-transposed %u / %.2f, per-pair %u / %.2f, per-pair-transposed %u / %.2f\n",
                cl.first[1] - cl.first[0], cc[0] / 1e9, cl.first[2] - cl.first[1], cc[1] / 1e9, cl.first[3] - cl.first[2], cc[2] / 1e9,
                cl.first[4] - cl.first[3], cc[3] / 1e9);
    }
    tm.lap("offsets+items");

    struct ws_t {
        rsk_ctx *ctx;
        std::vector<void *> all;
        ~ws_t() { for (void *p : all) rsk_pool_free(ctx, p); }
    } ws{ ctx, {} };
    auto dalloc = [&](void **p, size_t bytes) -> int {
        int r = rsk_pool_alloc(ctx, p, std::max<size_t>(bytes, 16));
        if (r != RSK_OK) return r;
        ws.all.push_back(*p);
        return RSK_OK;
    };
    char *D = nullptr;
    if ((rc = dalloc((void **) &D, hb.bytes)) != RSK_OK) return rc;
    RSK_HIP(hipMemcpyAsync(D, H, hb.bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *d_ia = (uint32_t *) (D + o_ia), *d_ib = (uint32_t *) (D + o_ib), *d_slot = (uint32_t *) (D + o_slot);
    uint64_t *d_tboff = (uint64_t *) (D + o_tboff), *d_pend = (uint64_t *) (D + o_pend), *d_scoff = (uint64_t *) (D + o_scoff);
    uint64_t *d_bndoff = (uint64_t *) (D + o_bndoff);
    // device results: one blob, copied back in one piece
    swf_blob rb;
    const size_t r_score = rb.add(npairs * 4), r_loa = rb.add(npairs * 4), r_lob = rb.add(npairs * 4), r_plen = rb.add(npairs * 4);
    const size_t r_lddt = rb.add(npairs * 4), r_counts = rb.add(npairs * 16), r_outoff = rb.add((npairs + 1) * 8);
    char *RD = nullptr;
    if ((rc = dalloc((void **) &RD, rb.bytes)) != RSK_OK) return rc;
    float *d_score = (float *) (RD + r_score), *d_lddt = (float *) (RD + r_lddt);
    uint32_t *d_loa = (uint32_t *) (RD + r_loa), *d_lob = (uint32_t *) (RD + r_lob), *d_plen = (uint32_t *) (RD + r_plen);
    uint32_t *d_counts = (uint32_t *) (RD + r_counts);
    uint64_t *d_outoff = (uint64_t *) (RD + r_outoff);
    uint32_t *d_bi, *d_bj, *d_pos = nullptr;
    uint64_t *d_pstart, *d_sizes;
    int *d_bnd = nullptr;
    uint8_t *d_tb;
    float *d_frac = nullptr;
    char *d_paths, *d_packed = nullptr;
    if ((rc = dalloc((void **) &d_tb, tbo + 64)) != RSK_OK) return rc;
    if (bno && (rc = dalloc((void **) &d_bnd, bno * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_bi, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_bj, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_pstart, npairs * 8)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_sizes, (npairs + 1) * 8)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_paths, pe + 16)) != RSK_OK) return rc;
    tm.lap("alloc+h2d");

    swf_args a = {};
    a.a_prof = dba->d_prof; a.a_off = dba->d_off; a.a_len = dba->d_len; a.a_npad = dba->npad; a.a_ra = dba->d_prof_ra; a.a_cb = dba->d_prof_cb;
    a.b_prof = dbb->d_prof; a.b_npad = dbb->npad;
    a.b_cb = dbb->d_prof_cb; a.b_off = dbb->d_off; a.b_len = dbb->d_len;
    a.ia = d_ia; a.ib = d_ib;
    a.items = (const swf_item *) (D + o_items);
    a.open = gap_open; a.ext = gap_ext;
    a.tb = d_tb; a.tb_off = d_tboff;
    a.score = d_score; a.besti = d_bi; a.bestj = d_bj;
    a.bnd = d_bnd; a.bnd_off = d_bndoff;
    a.clk = rsk_swqp_clock_words(ctx->device);
    // the four events rsk_align_last_times reads belong to THIS function alone and are valid only as a set (ev0 / ev1
    // are shared with every other launch of the context)
    ctx->al_times_valid = false;
    if (!ctx->ev_tb) {
        hipEvent_t e[4] = { nullptr, nullptr, nullptr, nullptr };
        for (int k = 0; k < 4; ++k)
            if (hipError_t ee = hipEventCreate(&e[k])) { for (int j = 0; j < k; ++j) (void) hipEventDestroy(e[j]); return rsk_hip_fail(ee, "hipEventCreate", __FILE__, __LINE__); }
        ctx->ev_al0 = e[0]; ctx->ev_al1 = e[1]; ctx->ev_st = e[2]; ctx->ev_tb = e[3];
    }
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    RSK_HIP(hipEventRecord(ctx->ev_al0, ctx->stream));
    {
        static std::atomic<int> attr_done[64];      // per device: the attribute belongs to the device's code object
        const int arc = rsk_once_per_device(attr_done, ctx->device, [&]() -> int {
            RSK_HIP(hipFuncSetAttribute((const void *) k_sw_qp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) SWQ_LDS_BYTES));
            RSK_HIP(hipFuncSetAttribute((const void *) k_sw_qp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) SWQ_LDS_BYTES));
            RSK_HIP(hipFuncSetAttribute((const void *) k_lddt_long, hipFuncAttributeMaxDynamicSharedMemorySize, LDDT_LONG_COLS * 32));
            return RSK_OK;
        });
        if (arc != RSK_OK) return arc;
    }
    for (int c = 0; c < 2; ++c) {
        if (qitems[c].empty()) continue;
        const swq_item *d_q = (const swq_item *) (D + o_q[c]);
        const dim3 grid((unsigned) qitems[c].size()), block(64 * SWQ_NW);
        if (c == 0) hipLaunchKernelGGL(k_sw_qp<false>, grid, block, SWQ_LDS_BYTES, ctx->stream, a, d_q);
        else hipLaunchKernelGGL(k_sw_qp<true>, grid, block, SWQ_LDS_BYTES, ctx->stream, a, d_q);
    }
    if (nitems_normal) {
        a.nitems = nitems_normal;
        hipLaunchKernelGGL(k_sw_float<false>, dim3((a.nitems + SWF_WAVES - 1) / SWF_WAVES), dim3(64 * SWF_WAVES), 0, ctx->stream, a, 0u);
    }
    if (items.size() > nitems_normal) {
        a.nitems = (uint32_t) items.size() - nitems_normal;
        hipLaunchKernelGGL(k_sw_float<true>, dim3((a.nitems + SWF_WAVES - 1) / SWF_WAVES), dim3(64 * SWF_WAVES), 0, ctx->stream, a,
                           nitems_normal);
    }
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RSK_HIP(hipEventRecord(ctx->ev_al1, ctx->stream));
    hipLaunchKernelGGL(k_traceback, dim3((unsigned) ((npairs + 63) / 64)), dim3(64), 0, ctx->stream, d_tb, d_tboff, d_ia, dba->d_len,
                       d_ib, dbb->d_len, cl, d_score, d_bi, d_bj, (uint32_t) npairs, d_paths, d_pend, d_pstart, d_plen, d_loa, d_lob,
                       (const swq_item *) (D + o_q[0]), (const swq_item *) (D + o_q[1]), (const uint32_t *) (D + o_qpitem));
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev_tb, ctx->stream));
    if (want_stats) {
        if ((rc = dalloc((void **) &d_pos, 2 * so * 4)) != RSK_OK) return rc;
        if ((rc = dalloc((void **) &d_frac, so * 4)) != RSK_OK) return rc;
        // pairs whose alignment has more columns than a wave stages (> 256): listed by k_lddt itself, two classes
        uint32_t *d_long_cnt = nullptr, *d_long_list = nullptr;
        const bool any_long = !lddt_list[0].empty() || !lddt_list[1].empty();     // (host bound: a pair's shorter chain allows > 256 columns)
        if ((rc = dalloc((void **) &d_long_cnt, 16)) != RSK_OK) return rc;
        if ((rc = dalloc((void **) &d_long_list, any_long ? 2 * npairs * 4 : 16)) != RSK_OK) return rc;
        RSK_HIP(hipMemsetAsync(d_long_cnt, 0, 16, ctx->stream));
        // RSK_LDDT_SPLIT=1: the pairs with paths of <= 128 characters through an instance of their own (20 KB of LDS per workgroup: 7
        // waves per SIMD instead of 4).  Measured r06, same box, alternating: 0.674 / 0.674 ms against 0.644 / 0.651 ms for the one
        // instance on the 72,000-pair structure set, configs[4] share 5.37 / 5.26 s against 5.34 / 5.17 s -- more resident waves do
        // not shorten the kernel (its waits are the LDS queue round trips of a step, not memory), a second launch costs: off.
        static const int lddt_split = getenv("RSK_LDDT_SPLIT") && atoi(getenv("RSK_LDDT_SPLIT")) == 1;
        if (lddt_split)
            hipLaunchKernelGGL((k_lddt<LDDT_SHORT_COLS, true>), dim3((unsigned) ((npairs + 3) / 4)), dim3(256), 0, ctx->stream, d_paths, d_pstart, d_plen, d_loa, d_lob,
                               d_ia, d_ib, dba->d_off, dbb->d_off, dba->d_x, dba->d_y, dba->d_z, dbb->d_x, dbb->d_y, dbb->d_z,
                               (uint32_t) npairs, d_pos, d_scoff, d_frac, d_lddt, d_counts, d_score, min_fwd_score, dba->d_seq, dbb->d_seq,
                               d_long_cnt, d_long_list, 1);
        hipLaunchKernelGGL((k_lddt<LDDT_LDS_COLS, false>), dim3((unsigned) ((npairs + 3) / 4)), dim3(256), 0, ctx->stream, d_paths, d_pstart, d_plen, d_loa, d_lob,
                           d_ia, d_ib, dba->d_off, dbb->d_off, dba->d_x, dba->d_y, dba->d_z, dbb->d_x, dbb->d_y, dbb->d_z,
                           (uint32_t) npairs, d_pos, d_scoff, d_frac, d_lddt, d_counts, d_score, min_fwd_score, dba->d_seq, dbb->d_seq,
                           d_long_cnt, d_long_list, lddt_split);
        for (int c = 0; c < 2; ++c) {
            // candidates of class 1 (shorter chain > 1024) may end with <= 1024 columns: both launches run whenever the host
            // bound allows the class or a longer one; a launch is a few workgroups per CU that walk the device list
            const size_t bound = c == 0 ? lddt_list[0].size() + lddt_list[1].size() : lddt_list[1].size();
            const uint32_t cap = c == 0 ? 1024u : (uint32_t) LDDT_LONG_COLS;
            if (bound == 0) continue;
            const unsigned grid = (unsigned) std::min<size_t>(bound, (size_t) std::max(1, ctx->num_cus) * (c == 0 ? 4 : 1));
            hipLaunchKernelGGL(k_lddt_long, dim3(grid), dim3(256), (size_t) cap * 32, ctx->stream, (const uint32_t *) (d_long_list + (size_t) c * npairs),
                               (const uint32_t *) (d_long_cnt + c), cap, d_ia,
                               d_ib, dba->d_off, dbb->d_off, dba->d_x, dba->d_y, dba->d_z, dbb->d_x, dbb->d_y, dbb->d_z, d_pos, d_scoff, d_lddt,
                               d_counts, d_score, min_fwd_score);
        }
        RSK_HIP(hipGetLastError());
    }
    RSK_HIP(hipEventRecord(ctx->ev_st, ctx->stream));
    ctx->al_times_valid = true;
    if (paths) {
        // pack the paths in the caller's order on the device
        hipLaunchKernelGGL(k_path_sizes, dim3((unsigned) ((npairs + 255) / 256)), dim3(256), 0, ctx->stream, d_slot, d_plen, (uint32_t) npairs, d_sizes);
        size_t tmp_bytes = 0;
        RSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_sizes, d_outoff, (int) npairs + 1, ctx->stream));
        void *d_tmp = nullptr;
        if ((rc = dalloc(&d_tmp, tmp_bytes)) != RSK_OK) return rc;
        RSK_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_sizes, d_outoff, (int) npairs + 1, ctx->stream));
        if ((rc = dalloc((void **) &d_packed, need + 16)) != RSK_OK) return rc;
        hipLaunchKernelGGL(k_path_pack, dim3((unsigned) ((npairs + 3) / 4)), dim3(256), 0, ctx->stream, d_slot, d_plen, d_pstart, d_paths,
                           d_outoff, (uint32_t) npairs, d_packed);
        RSK_HIP(hipGetLastError());
    }
    void *rpin = nullptr;
    if ((rc = rsk_pinned(ctx, 1, rb.bytes, &rpin)) != RSK_OK) return rc;
    RSK_HIP(hipMemcpyAsync(rpin, RD, rb.bytes, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = rsk_stream_wait(ctx)) != RSK_OK) return rc;
    tm.lap("kernels+d2h");
    const char *RH = (const char *) rpin;
    const float *h_score = (const float *) (RH + r_score), *h_lddt = (const float *) (RH + r_lddt);
    const uint32_t *h_loa = (const uint32_t *) (RH + r_loa), *h_lob = (const uint32_t *) (RH + r_lob), *h_plen = (const uint32_t *) (RH + r_plen);
    const uint32_t *h_counts = (const uint32_t *) (RH + r_counts);
    const uint64_t *h_outoff = (const uint64_t *) (RH + r_outoff);
    if (paths) {
        const uint64_t total = h_outoff[npairs];
        RSK_HIP(hipMemcpyAsync(paths, d_packed, total, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipStreamSynchronize(ctx->stream));
        tm.lap("paths d2h");
    }

    rsk_parallel_for(npairs, 16384, [&](size_t p_lo, size_t p_hi) {
    for (size_t p = p_lo; p < p_hi; ++p) {
        const size_t k = slot[p];
        rsk_aln &o = out[p];
        const uint32_t LA = dba->len[ia[p]], LB = dbb->len[ib[p]];
        o.score = h_score[k];
        o.lo_a = h_loa[k]; o.lo_b = h_lob[k];
        o.path_len = h_plen[k];
        o.hi_a = o.hi_b = o.ids = o.gaps = o.nident = RSK_NO_POS;
        o.lddt = o.pvalue = o.evalue = o.qual = FLT_MAX;
        o.ts = -FLT_MAX;
        o.path_off = paths ? h_outoff[p] : 0;
        // CalcEvalue dssaligner.cpp:852-904 (the double-precision pow stays on the host: libm)
        if (want_stats && paths && !(o.score < min_fwd_score)) {
            const uint32_t nM = h_counts[4 * k], nD = h_counts[4 * k + 1], nI = h_counts[4 * k + 2];
            o.hi_a = o.lo_a + nM + nD - 1;
            o.hi_b = o.lo_b + nM + nI - 1;
            o.ids = nM;
            o.gaps = nD + nI;
            o.nident = h_counts[4 * k + 3];
            const float sra = dba->h_selfrev[ia[p]], srb = dbb->h_selfrev[ib[p]];
            float rev = 0;
            if (sra != FLT_MAX && srb != FLT_MAX) rev = (sra + srb) / 2;
            const float L = float(LA + LB) / 2;
            const float dpw = 1.7f, lddtw = 0.13f, ladd = 250.0f, revtsw = 2.0f;
            float ts = lddtw * h_lddt[k];
            ts += (dpw * o.score - revtsw * rev) / (L + ladd);
            o.lddt = h_lddt[k];
            o.ts = ts;
            const double pv = swf_pvalue(ts);
            o.pvalue = (float) pv;
            o.qual = (float) swf_qual(ts);
            o.evalue = (float) (pv * 8340.0);      // SCOP40c_DBSIZE statsig.h:3
        }
    }
    });
    tm.lap("host stats");
    ctx->al_pairs = npairs; ctx->al_cells = cells; ctx->al_tb_bytes = tbo;
    return RSK_OK;
}

// CalcEvalue (dssaligner.cpp:852-904) for alignments that already sit on the device as paths -- the long-chain (MKF) pairs
// of rsk_mkf_align_pairs (k_xdrop.hip): LDDT over the aligned columns (k_lddt / k_lddt_long), path packing in pair order,
// the statistics with libm pow on the host.  d_* arrays are in pair order; d_pstart = absolute offset of a pair's path in
// d_paths, d_plen its length (0 = no alignment), d_score the alignment score (0 = no alignment).
int rsk_paths_stats_pack(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, size_t npairs, const uint32_t *ia, const uint32_t *ib,
                         const uint32_t *d_ia, const uint32_t *d_ib, const char *d_paths, const uint64_t *d_pstart, const uint32_t *d_plen,
                         const uint32_t *d_loa, const uint32_t *d_lob, const float *d_score, float min_fwd_score, rsk_aln *out, char *paths,
                         size_t paths_bytes)
{
    if (npairs == 0) return RSK_OK;
    if (!dba->d_x || !dbb->d_x) { rsk_set_error("rsk_paths_stats_pack: chain sets have no coordinates"); return RSK_E_INVALID; }
    struct ws_t {
        rsk_ctx *ctx;
        std::vector<void *> all;
        ~ws_t() { for (void *p : all) rsk_pool_free(ctx, p); }
    } ws{ ctx, {} };
    auto dalloc = [&](void **p, size_t bytes) -> int {
        int r = rsk_pool_alloc(ctx, p, std::max<size_t>(bytes, 16));
        if (r != RSK_OK) return r;
        ws.all.push_back(*p);
        return RSK_OK;
    };
    int rc;
    // scratch offsets (aligned columns <= min(LA, LB)), candidate lists of the long-alignment LDDT kernel, identity slot
    std::vector<uint64_t> sc_off(npairs + 1);
    std::vector<uint32_t> lddt_list[2], ident(npairs);
    uint64_t so = 0;
    size_t need = 0;
    for (size_t p = 0; p < npairs; ++p) {
        const uint32_t LA = dba->len[ia[p]], LB = dbb->len[ib[p]], m = std::min(LA, LB);
        sc_off[p] = so;
        so += m;
        if (m > LDDT_LDS_COLS) lddt_list[m <= 1024 ? 0 : 1].push_back((uint32_t) p);
        ident[p] = (uint32_t) p;
        need += (size_t) LA + LB + 1;
    }
    sc_off[npairs] = so;
    if (paths_bytes < need) { rsk_set_error("rsk_paths_stats_pack: paths buffer too small (%zu < %zu)", paths_bytes, need); return RSK_E_INVALID; }
    uint64_t *d_scoff, *d_sizes, *d_outoff;
    uint32_t *d_pos, *d_counts, *d_ident, *d_list[2] = { nullptr, nullptr };
    float *d_frac, *d_lddt;
    char *d_packed;
    if ((rc = dalloc((void **) &d_scoff, (npairs + 1) * 8)) || (rc = dalloc((void **) &d_sizes, (npairs + 1) * 8)) ||
        (rc = dalloc((void **) &d_outoff, (npairs + 1) * 8)) || (rc = dalloc((void **) &d_pos, 2 * so * 4)) || (rc = dalloc((void **) &d_frac, so * 4)) ||
        (rc = dalloc((void **) &d_counts, npairs * 16)) || (rc = dalloc((void **) &d_lddt, npairs * 4)) || (rc = dalloc((void **) &d_ident, npairs * 4)) ||
        (rc = dalloc((void **) &d_packed, need + 16)))
        return rc;
    RSK_HIP(hipMemcpyAsync(d_scoff, sc_off.data(), (npairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_ident, ident.data(), npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    (void) d_list;
    uint32_t *d_long_cnt = nullptr, *d_long_list = nullptr;
    const bool any_long = !lddt_list[0].empty() || !lddt_list[1].empty();
    if ((rc = dalloc((void **) &d_long_cnt, 16)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_long_list, any_long ? 2 * npairs * 4 : 16)) != RSK_OK) return rc;
    RSK_HIP(hipMemsetAsync(d_long_cnt, 0, 16, ctx->stream));
    // (the long-chain path's alignments: the one-instance form)
    hipLaunchKernelGGL((k_lddt<LDDT_LDS_COLS, false>), dim3((unsigned) ((npairs + 3) / 4)), dim3(256), 0, ctx->stream, d_paths, d_pstart, d_plen, d_loa, d_lob, d_ia, d_ib,
                       dba->d_off, dbb->d_off, dba->d_x, dba->d_y, dba->d_z, dbb->d_x, dbb->d_y, dbb->d_z, (uint32_t) npairs, d_pos, d_scoff, d_frac,
                       d_lddt, d_counts, d_score, min_fwd_score, dba->d_seq, dbb->d_seq, d_long_cnt, d_long_list, 0);
    for (int c = 0; c < 2; ++c) {
        const size_t bound = c == 0 ? lddt_list[0].size() + lddt_list[1].size() : lddt_list[1].size();
        const uint32_t cap = c == 0 ? 1024u : (uint32_t) LDDT_LONG_COLS;
        if (bound == 0) continue;
        static std::atomic<int> lddt_attr[64];
        const int arc = rsk_once_per_device(lddt_attr, ctx->device, [&]() -> int {
            RSK_HIP(hipFuncSetAttribute((const void *) k_lddt_long, hipFuncAttributeMaxDynamicSharedMemorySize, LDDT_LONG_COLS * 32));
            return RSK_OK;
        });
        if (arc != RSK_OK) return arc;
        const unsigned grid = (unsigned) std::min<size_t>(bound, (size_t) std::max(1, ctx->num_cus) * (c == 0 ? 4 : 1));
        hipLaunchKernelGGL(k_lddt_long, dim3(grid), dim3(256), (size_t) cap * 32, ctx->stream, (const uint32_t *) (d_long_list + (size_t) c * npairs),
                           (const uint32_t *) (d_long_cnt + c), cap, d_ia, d_ib, dba->d_off, dbb->d_off,
                           dba->d_x, dba->d_y, dba->d_z, dbb->d_x, dbb->d_y, dbb->d_z, d_pos, d_scoff, d_lddt, d_counts, d_score, min_fwd_score);
    }
    RSK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_path_sizes, dim3((unsigned) ((npairs + 255) / 256)), dim3(256), 0, ctx->stream, d_ident, d_plen, (uint32_t) npairs, d_sizes);
    size_t tmp_bytes = 0;
    RSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_sizes, d_outoff, (int) npairs + 1, ctx->stream));
    void *d_tmp = nullptr;
    if ((rc = dalloc(&d_tmp, tmp_bytes)) != RSK_OK) return rc;
    RSK_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_sizes, d_outoff, (int) npairs + 1, ctx->stream));
    hipLaunchKernelGGL(k_path_pack, dim3((unsigned) ((npairs + 3) / 4)), dim3(256), 0, ctx->stream, d_ident, d_plen, d_pstart, d_paths, d_outoff,
                       (uint32_t) npairs, d_packed);
    RSK_HIP(hipGetLastError());
    std::vector<float> h_score(npairs), h_lddt(npairs);
    std::vector<uint32_t> h_loa(npairs), h_lob(npairs), h_plen(npairs), h_counts(4 * npairs);
    std::vector<uint64_t> h_outoff(npairs + 1);
    RSK_HIP(hipMemcpyAsync(h_score.data(), d_score, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(h_lddt.data(), d_lddt, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(h_loa.data(), d_loa, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(h_lob.data(), d_lob, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(h_plen.data(), d_plen, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(h_counts.data(), d_counts, npairs * 16, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(h_outoff.data(), d_outoff, (npairs + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    if (h_outoff[npairs]) {
        RSK_HIP(hipMemcpyAsync(paths, d_packed, h_outoff[npairs], hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipStreamSynchronize(ctx->stream));
    }
    rsk_parallel_for(npairs, 16384, [&](size_t p_lo, size_t p_hi) {
        for (size_t p = p_lo; p < p_hi; ++p) {
            rsk_aln &o = out[p];
            const uint32_t LA = dba->len[ia[p]], LB = dbb->len[ib[p]];
            o.score = h_score[p];
            o.lo_a = h_loa[p]; o.lo_b = h_lob[p];
            o.path_len = h_plen[p];
            o.hi_a = o.hi_b = o.ids = o.gaps = o.nident = RSK_NO_POS;
            o.lddt = o.pvalue = o.evalue = o.qual = FLT_MAX;
            o.ts = -FLT_MAX;
            o.path_off = h_outoff[p];
            if (o.path_len == 0) continue;
            // PostAlignMKF sets Hi from the path counts before CalcEvalue (dssaligner.cpp:1425-1428): also below MinFwdScore
            uint32_t nM = 0, nD = 0, nI = 0;
            if (!(o.score < min_fwd_score)) { nM = h_counts[4 * p]; nD = h_counts[4 * p + 1]; nI = h_counts[4 * p + 2]; }
            else
                for (const char *c = paths + o.path_off; *c; ++c) { nM += *c == 'M'; nD += *c == 'D'; nI += *c == 'I'; }
            o.hi_a = o.lo_a + nM + nD - 1;
            o.hi_b = o.lo_b + nM + nI - 1;
            if (o.score < min_fwd_score) continue;                         // CalcEvalue dssaligner.cpp:861
            o.ids = nM;
            o.gaps = nD + nI;
            o.nident = h_counts[4 * p + 3];
            const float sra = dba->h_selfrev[ia[p]], srb = dbb->h_selfrev[ib[p]];
            float rev = 0;
            if (sra != FLT_MAX && srb != FLT_MAX) rev = (sra + srb) / 2;
            const float L = float(LA + LB) / 2;
            const float dpw = 1.7f, lddtw = 0.13f, ladd = 250.0f, revtsw = 2.0f;
            float ts = lddtw * h_lddt[p];
            ts += (dpw * o.score - revtsw * rev) / (L + ladd);
            o.lddt = h_lddt[p];
            o.ts = ts;
            const double pv = swf_pvalue(ts);
            o.pvalue = (float) pv;
            o.qual = (float) swf_qual(ts);
            o.evalue = (float) (pv * 8340.0);
        }
    });
    return RSK_OK;
}

// ---------------------------------------------------------------------------------------------
// D1: SWFastGapless swgapless.cpp:46-97 on the SetSMx_NoRev matrix (dead code in the reference; pair-list
// form).  One workgroup per pair, one thread per diagonal: S(i,j) is summed in feature order from the
// tables, H = max(H, 0) + S runs along the diagonal in the reference's operand order; the best cell is
// the first maximum in row-major order (strict > per diagonal, then value desc / i asc / j asc).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gapless_float(swf_args a, float *score, uint32_t *besti, uint32_t *bestj)
{
    __shared__ float sb[256];
    __shared__ uint32_t si[256], sj[256];
    const uint32_t p = blockIdx.x;
    const uint32_t A = a.ia[p], B = a.ib[p];
    const int LA = (int) a.a_len[A], LB = (int) a.b_len[B];
    const uint8_t *pa = a.a_prof + a.a_off[A], *pb = a.b_prof + a.b_off[B];
    const int toff[8] = { 0, 400, 656, 912, 1168, 1424, 1680, 1936 };
    const int asz[8] = { 20, 16, 16, 16, 16, 16, 16, 16 };
    float best = 0.0f;
    uint32_t bi = 0xFFFFFFFFu, bj = 0xFFFFFFFFu;
    for (int d = threadIdx.x; d < LA + LB - 1; d += blockDim.x) {
        int i = LA - 1 - d; if (i < 0) i = 0;
        int j = d - (LA - 1); if (j < 0) j = 0;
        float x = 0.0f;
        for (; i < LA && j < LB; ++i, ++j) {
            float S = c_swf_tables.t[toff[0] + pa[i] * asz[0] + pb[j]];
#pragma unroll
            for (int f = 1; f < 8; ++f) S += c_swf_tables.t[toff[f] + pa[(size_t) f * a.a_npad + i] * asz[f] + pb[(size_t) f * a.b_npad + j]];
            if (x < 0.0f) x = 0.0f;
            x += S;
            if (x > best || (x == best && x > 0.0f && ((uint32_t) i < bi || ((uint32_t) i == bi && (uint32_t) j < bj)))) {
                best = x; bi = (uint32_t) i; bj = (uint32_t) j;
            }
        }
    }
    sb[threadIdx.x] = best; si[threadIdx.x] = bi; sj[threadIdx.x] = bj;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int) threadIdx.x < s) {
            const float ob = sb[threadIdx.x + s];
            const uint32_t oi = si[threadIdx.x + s], oj = sj[threadIdx.x + s];
            const float mb = sb[threadIdx.x];
            const uint32_t mi = si[threadIdx.x], mj = sj[threadIdx.x];
            if (ob > mb || (ob == mb && ob > 0.0f && (oi < mi || (oi == mi && oj < mj)))) { sb[threadIdx.x] = ob; si[threadIdx.x] = oi; sj[threadIdx.x] = oj; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { score[p] = sb[0]; besti[p] = si[0]; bestj[p] = sj[0]; }
}

extern "C" int rsk_gapless_float_pairs(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, const uint32_t *ia, const uint32_t *ib,
                                       size_t npairs, float *scores, uint32_t *besti, uint32_t *bestj)
{
    if (!ctx || !dba || !dbb || (npairs && (!ia || !ib || !scores))) { rsk_set_error("rsk_gapless_float_pairs: NULL argument"); return RSK_E_INVALID; }
    if (!dba->d_prof || !dbb->d_prof) { rsk_set_error("rsk_gapless_float_pairs: chain set has no profiles"); return RSK_E_INVALID; }
    if (npairs == 0) return RSK_OK;
    if (npairs > 0x7FFFFFFFull) { rsk_set_error("rsk_gapless_float_pairs: too many pairs"); return RSK_E_RANGE; }
    for (size_t p = 0; p < npairs; ++p)
        if (ia[p] >= dba->n || ib[p] >= dbb->n) { rsk_set_error("rsk_gapless_float_pairs: pair %zu out of range", p); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = swf_upload_tables(ctx);
    if (rc != RSK_OK) return rc;
    struct ws_t {
        rsk_ctx *ctx;
        std::vector<void *> all;
        ~ws_t() { for (void *p : all) rsk_pool_free(ctx, p); }
    } ws{ ctx, {} };
    auto dalloc = [&](void **p, size_t bytes) -> int {
        int r = rsk_pool_alloc(ctx, p, std::max<size_t>(bytes, 16));
        if (r != RSK_OK) return r;
        ws.all.push_back(*p);
        return RSK_OK;
    };
    uint32_t *d_ia, *d_ib, *d_bi, *d_bj;
    float *d_sc;
    if ((rc = dalloc((void **) &d_ia, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_ib, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_bi, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_bj, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_sc, npairs * 4)) != RSK_OK) return rc;
    RSK_HIP(hipMemcpyAsync(d_ia, ia, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_ib, ib, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    swf_args a = {};
    a.a_prof = dba->d_prof; a.a_off = dba->d_off; a.a_len = dba->d_len; a.a_npad = dba->npad;
    a.b_prof = dbb->d_prof; a.b_off = dbb->d_off; a.b_len = dbb->d_len; a.b_npad = dbb->npad;
    a.ia = d_ia; a.ib = d_ib;
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_gapless_float, dim3((unsigned) npairs), dim3(256), 0, ctx->stream, a, d_sc, d_bi, d_bj);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    RSK_HIP(hipMemcpyAsync(scores, d_sc, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (besti) RSK_HIP(hipMemcpyAsync(besti, d_bi, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (bestj) RSK_HIP(hipMemcpyAsync(bestj, d_bj, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}

// kernel times of the stages of the last rsk_align_pairs call on this context (HIP events on its stream): Smith-Waterman
// kernels, traceback kernel, statistics kernels (LDDT); -1 where a stage did not run
// {cycles, ticks} of k_sw_qp per device: allocated on first use, never freed (16 bytes), zeroed by rsk_path_counters_reset
unsigned long long *rsk_swqp_clock_words(int device)
{
    static std::mutex m;
    static unsigned long long *words[64];
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(m);
    if (!words[device]) {
        rsk_device_guard g(device);
        unsigned long long *p = nullptr;
        if (hipMalloc((void **) &p, 16) != hipSuccess || hipMemset(p, 0, 16) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
        words[device] = p;
    }
    return words[device];
}

extern "C" int rsk_align_last_times(rsk_ctx *ctx, float *sw_ms, float *traceback_ms, float *stats_ms)
{
    if (!ctx) { rsk_set_error("rsk_align_last_times: ctx is NULL"); return RSK_E_INVALID; }
    float a = -1.0f, b = -1.0f, c = -1.0f;
    // -1 unless the last rsk_align_pairs call of this context recorded all four of its events
    if (ctx->al_times_valid && ctx->ev_st && hipEventSynchronize(ctx->ev_st) == hipSuccess) {
        (void) hipEventElapsedTime(&a, ctx->ev_al0, ctx->ev_al1);
        (void) hipEventElapsedTime(&b, ctx->ev_al1, ctx->ev_tb);
        (void) hipEventElapsedTime(&c, ctx->ev_tb, ctx->ev_st);
    }
    if (sw_ms) *sw_ms = a;
    if (traceback_ms) *traceback_ms = b;
    if (stats_ms) *stats_ms = c;
    return RSK_OK;
}

extern "C" int rsk_align_last_work(rsk_ctx *ctx, uint64_t *pairs, uint64_t *cells, uint64_t *tb_bytes)
{
    if (!ctx) { rsk_set_error("rsk_align_last_work: ctx is NULL"); return RSK_E_INVALID; }
    if (pairs) *pairs = ctx->al_pairs;
    if (cells) *cells = ctx->al_cells;
    if (tb_bytes) *tb_bytes = ctx->al_tb_bytes;
    return RSK_OK;
}

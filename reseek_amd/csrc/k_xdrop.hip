// k_xdrop.hip -- P9, second half: the banded ("X-drop") float DP of the long-chain MKF path on the device.
//
//   XDropFwd  xdropfwd.cpp:71-390 (+ its traceback :10-67)
//   XDropBwd  xdropbwd.cpp:28-52  (= XDropFwd on the reversed prefixes, path reversed)
//   SubstScore xdrophsp.cpp:8     (sum of the 8 weighted feature tables, feature order 0 -> 7)
// as XDropHSP (xdrophsp.cpp:42) calls them for one seeded pair: forward from (LoA, LoB) to the chain ends,
// backward from (LoA - 1, LoB - 1) to the chain starts.
//
// ONE WAVE PER EXTENSION (k_xdrop_wave).  The reference's code reads as sequential along a row (the insert state I0 and
// the running row bounds change from cell to cell), but every value of row i depends on row i - 1 only, so the band
// columns of a row are computed 64 at a time with exact scans (below).  Row state lives in an LDS ring, trace bytes go
// to HBM scratch (the trace of a pair can reach LA x LB bytes; 288 GB of HBM take tens of thousands of them at once).
// Every float operation is the reference's, in its order (-ffp-contract=off), so scores and paths are bit-identical:
// tests/test_gpu_xdrop.py runs the kernel on the reference's own -test_xdrop vectors (explicit score matrices, through
// rsk_xdrop_fwd / rsk_xdrop_bwd), on the reference's per-stage fixtures of real long-chain pairs, and against the CPU
// oracle's restatement (oracle/rsk_oracle.c rsko_xdrop_*) on seeded pairs with the length tail.
// This is the only implementation of the gapped extensions in the product: there is no host or thread-per-extension form.
//
// Algorithmic bytes per extension: 16 B of table offsets per row + 16 B per cell (column offsets), 1 B of trace per cell.
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rsk_internal.h"
#include "rsk_tables_data.h"

#define XD_MINUS_INF (-9e9f)         // xdpmem.h:6
#define XD_DM 0x01                   // tracebit.h:4-8
#define XD_IM 0x02
#define XD_MD 0x04
#define XD_MI 0x08
#define XD_TABLE_FLOATS (400 + 7 * 256)

struct xd_tables { float t[XD_TABLE_FLOATS]; };
static __device__ __constant__ xd_tables c_xd_tables;
static const int h_xd_toff[8] = { 0, 400, 656, 912, 1168, 1424, 1680, 1936 };

static int xd_upload_tables(rsk_ctx *ctx)
{
    static std::atomic<int> done[64];
    return rsk_once_per_device(done, ctx->device, [&]() -> int {
    xd_tables h;
    for (int i = 0; i < XD_TABLE_FLOATS; ++i) h.t[i] = 0.0f;
    for (int f = 0; f < RSK_NFEATURES; ++f) {
        const int as = (int) rsk_feature_alpha[f];
        for (int a = 0; a < as; ++a)
            for (int b = 0; b < as; ++b) h.t[h_xd_toff[f] + a * as + b] = rsk_feature_mx[f][a * RSK_FEATURE_DIM + b];
    }
    RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_xd_tables), &h, sizeof(h)));
    return RSK_OK;
    });
}

struct xd_args {
    const uint16_t *a_ra;            // A: [npadA][8] table row offsets in bytes (letter * alphabet * 4)
    const uint16_t *b_cb;            // B: [npadB][8] column offsets in bytes (letter * 4)
    const uint32_t *a_off, *b_off, *a_len, *b_len;
    const uint32_t *ia, *ib, *lo_a, *lo_b;     // requests
    uint32_t nreq;
    float X, open, ext;
    float *rows;                     // per extension: LB' + 9 {Mrow, Drow} float2 records, at rows + row_off[e]
    const uint64_t *row_off;
    uint8_t *tb;                     // per extension: (LA' + 9) x (LB' + 9) trace bytes (zeroed), at tb + tb_off[e]
    const uint64_t *tb_off;
    float *score;                    // [2 * nreq]: fwd, bwd
    char *paths;                     // per extension a slot of LA' + LB' + 2 chars at paths + path_off[e]
    const uint64_t *path_off;
    uint32_t *path_start;            // offset of the first path character inside the slot
    uint32_t *path_len;
    // rsk_mkf_align_pairs: the starts come from k_mkf_start on the device, so the scratch is laid out per PAIR (sized for
    // the worst split) and the two extensions of a pair divide it by the start they find there
    int per_pair;                    // row_off / tb_off / path_off are indexed by request; the backward extension follows the forward one
    const uint8_t *valid;            // per request (optional): 0 = no extension wanted
    // EXPLICIT form (rsk_xdrop_fwd / rsk_xdrop_bwd): scores from smx[posA * smx_ld + posB]; a_len / b_len hold the matrix
    // shape, ia = ib = {0}; only the extension of direction only_dir (0 forward, 1 backward) is computed
    const float *smx;
    uint32_t smx_ld;
    int only_dir;                    // -1 = both
};

// ---------------------------------------------------------------------------------------------------------------------
// k_xdrop_wave: ONE WAVE per extension.  Every value of row i depends on row i - 1 only --
// Mrow[j] = Sub + max(M'[j-1], D'[j], I0), Drow[j] = max(M'[j-1] + Open, D'[j] + Ext), and even the insert state I0 is a
// running max over the PREVIOUS row's M (I0 <- max(I0 + Ext, M'[j-1] + Open)).  So the columns jlo .. jhi of a row are
// computed 64 at a time, one per lane:
//   * I0: a max-plus scan whose float additions must happen in the reference's order; f(x) = x + Ext is monotone, so
//     max(f(a), f(b)) = f(max(a, b)) holds exactly: a doubling scan whose step d applies f d times, add by add;
//   * BestScore as seen by column j = max(BestScore, prefix max of the row's earlier cells) -- max is associative, the
//     6-step DPP scan is exact; the best cell is the LAST one that reaches the row maximum (the reference updates on >=);
//   * the band of the next row: next_jlo is a min over per-cell candidates; next_jhi is the fold of "assign j+1" (h > 0)
//     and "max" (hd, hi) events in cell order, including the reference's UINT_MAX start value that max() cannot leave:
//     the last assigning cell and the events after it decide.
// The cells a row grows by (the last band column's tests may extend the row, cell by cell) run one at a time, executed
// uniformly by the wave.  An extension whose band outgrows the LDS ring is run again on HBM rows (same code, RING = false).
// EXPLICIT = true takes the substitution scores from a caller matrix instead of the profile tables (rsk_xdrop_fwd / _bwd:
// the form the reference's own self-test drives XDropFwd / XDropBwd in, test_xdrop.cpp:81-175).
#define XDW_WAVES 4
#ifdef XDW_PROF
// debug build only (tools/exp): wave-cycles per phase, rows, chunks, sequential cells, traceback steps
__device__ unsigned long long g_xdw_prof[12];
#define XDW_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define XDW_ADD(k, v) do { if (lane == 0) atomicAdd(&g_xdw_prof[k], (unsigned long long) (v)); } while (0)
extern "C" int rsk_debug_xdw_prof(unsigned long long *out)
{
    unsigned long long z[12] = { 0 };
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xdw_prof), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_xdw_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#else
#define XDW_T(var)
#define XDW_ADD(k, v)
#endif
__device__ __forceinline__ float xdw_shr1(float v, float lane0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane0), __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
__device__ __forceinline__ float xdw_bcast(float v, uint32_t l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int) l)); }

// Row state {Mrow[j], Drow[j]}: a ring of XDW_RING columns in LDS (the band of a row spans a few hundred columns), or the
// HBM array when the band of some row does not fit -- the extension is then simply run again in that mode.
#define XDW_RING 512
template <bool RING> struct xdw_rows {
    float2 *p;
    __device__ __forceinline__ float2 &operator()(uint32_t j) const { return RING ? p[j & (XDW_RING - 1)] : p[(int32_t) j]; }
};

// returns false iff RING and the band outgrew the ring (nothing final has been written then)
template <bool RING, bool EXPLICIT>
__device__ __forceinline__ bool xdw_extend(const xd_args &a, const float *tab, float2 *ring, uint32_t e, uint32_t lane)
{
    const uint32_t req = e >> 1, dir = e & 1;
    if ((a.valid && a.valid[req] != 1) || (a.only_dir >= 0 && (int) dir != a.only_dir)) { a.path_start[e] = 0; a.path_len[e] = 0; a.score[e] = 0.0f; return true; }
    const uint32_t A = a.ia[req], B = a.ib[req];
    const uint32_t LoA = a.lo_a[req], LoB = a.lo_b[req];
    const uint32_t LA = dir ? LoA : a.a_len[A] - LoA, LB = dir ? LoB : a.b_len[B] - LoB;      // extents of this extension
    uint64_t row_o, tb_o, path_o;
    if (a.per_pair) {
        row_o = a.row_off[req]; tb_o = a.tb_off[req]; path_o = a.path_off[req];
        if (dir) {                                                                              // behind the forward extension's share
            const uint64_t FA = a.a_len[A] - LoA, FB = a.b_len[B] - LoB;
            row_o += 2 * (FB + 9);
            tb_o += ((FA + 9) * (FB + 9) + 15) & ~15ull;
            path_o += FA + FB + 2;
        }
    } else { row_o = a.row_off[e]; tb_o = a.tb_off[e]; path_o = a.path_off[e]; }
    const uint16_t *RA = EXPLICIT ? nullptr : a.a_ra + (size_t) a.a_off[A] * 8, *CB = EXPLICIT ? nullptr : a.b_cb + (size_t) a.b_off[B] * 8;
    const char *tabb = (const char *) tab;
    const uint32_t toffb[8] = { 0 * 4, 400 * 4, 656 * 4, 912 * 4, 1168 * 4, 1424 * 4, 1680 * 4, 1936 * 4 };
    char *slot = a.paths + path_o;
    const uint32_t cap = LA + LB + 2;
    a.path_start[e] = 0;
    a.path_len[e] = 0;
    auto posA = [&](uint32_t i) { return dir ? LoA - i : LoA + i - 1; };
    auto posB = [&](uint32_t j) { return dir ? LoB - j : LoB + j - 1; };
    uint32_t rowo[8];
    const float *srow = nullptr;      // EXPLICIT: the matrix row of the current DP row
    // The 16-byte profile records of the row (A side) and of the columns (B side) are requested one step ahead of their
    // use: an extension is a chain of dependent steps, what it waits for is memory latency.  (EXPLICIT: the "record" of a
    // row is its matrix row index, the record of a column the score itself -- load_cb is only called for the current row.)
    auto load_ra = [&](uint32_t i) {
        if constexpr (EXPLICIT) return make_uint4(posA(i), 0, 0, 0);
        return *(const uint4 *) (RA + (size_t) posA(i) * 8);
    };
    auto load_cb = [&](uint32_t j) {
        if constexpr (EXPLICIT) return make_uint4(__builtin_bit_cast(uint32_t, srow[posB(j)]), 0, 0, 0);
        return *(const uint4 *) (CB + (size_t) posB(j) * 8);
    };
    auto set_row_w = [&](const uint4 w) {
        if constexpr (EXPLICIT) { srow = a.smx + (size_t) w.x * a.smx_ld; return; }
        const uint32_t ww[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
        for (int f = 0; f < 8; ++f) rowo[f] = toffb[f] + ((f & 1) ? (ww[f >> 1] >> 16) : (ww[f >> 1] & 0xFFFFu));
    };
    auto sub_w = [&](const uint4 w) { // SubstScore: Total = 0; Total += feature f, f = 0..7
        if constexpr (EXPLICIT) return __builtin_bit_cast(float, w.x);
        const uint32_t ww[4] = { w.x, w.y, w.z, w.w };
        float Total = 0.0f;
#pragma unroll
        for (int f = 0; f < 8; ++f) Total += *(const float *) (tabb + rowo[f] + ((f & 1) ? (ww[f >> 1] >> 16) : (ww[f >> 1] & 0xFFFFu)));
        return Total;
    };
    if (LA == 1 || LB == 1) {         // xdropfwd.cpp:84-92
        set_row_w(load_ra(1));
        const float Score = sub_w(load_cb(1));
        if (Score > 0) { slot[0] = 'M'; a.path_len[e] = 1; }
        a.score[e] = Score;
        return true;
    }
    const float Open = a.open, Ext = a.ext, X = a.X;
    const float AbsOpen = -Open, AbsExt = -Ext;
    const xdw_rows<RING> MDr{ RING ? ring : (float2 *) (a.rows + row_o) + 1 };     // index -1 is valid in either form
#define MD(j) MDr((uint32_t) (j))
    if (RING) {                                                   // as XDPMem::Alloc leaves its rows
        for (uint32_t k = lane; k < XDW_RING; k += 64) ring[k] = make_float2(0.0f, 0.0f);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    uint8_t *TB = a.tb + tb_o;
    const uint32_t Cols = LB + 1 + 8;                            // XDPMem::Alloc(LA + 1, LB + 1)
    const uint32_t U = 0xFFFFFFFFu;
    MD(-1).x = XD_MINUS_INF;
    MD(0).y = XD_MINUS_INF;
    MD(1).y = XD_MINUS_INF;
    float BestScore = 0;
    uint32_t Besti = 0, Bestj = 0;
    uint32_t prev_jlo = 0, prev_jhi = 0, jlo = 1, jhi = 1;
    float M0 = BestScore;
    uint4 ra_next = load_ra(1);
    for (uint32_t i = 1; i <= LA; ++i) {
        if (RING && jhi - jlo + 5 > XDW_RING) return false;       // columns jlo - 1 .. jhi + 2 are touched
        if (jlo == prev_jlo) { MD(jlo - 1).x = XD_MINUS_INF; MD(jlo).y = XD_MINUS_INF; }
        uint32_t endj = min(prev_jhi + 1, LB);
        for (uint32_t j = endj + 1 + lane; j <= min(jhi + 1, LB); j += 64) { MD(j - 1).x = XD_MINUS_INF; MD(j).y = XD_MINUS_INF; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t next_jlo = U, next_jhi = U;
        float I0 = XD_MINUS_INF;
        XDW_T(t_row);
        set_row_w(ra_next);
        if (i < LA) ra_next = load_ra(i + 1);
        const size_t tb_row = (size_t) i * Cols;
        // ---- columns jlo .. jhi (the band the previous row asked for), 64 per step ----
        const uint32_t jlast = jhi;                              // its cell may grow the row: handled after the step that holds it
        const uint32_t jb_end = jhi + 1;
        // the sequential part (cells the row grows by) starts behind the band: two columns in flight
        uint4 cb_t0 = load_cb(min(jb_end, LB)), cb_t1 = load_cb(min(jb_end + 1, LB));
        uint4 cb_next = make_uint4(0, 0, 0, 0);
        if (jlo + lane < jb_end) cb_next = load_cb(jlo + lane);
        float hL = 0.0f, hiL = 0.0f;
        for (uint32_t jc = jlo; jc < jb_end; jc += 64) {
            const uint32_t n = min(64u, jb_end - jc);
            const uint32_t j = jc + lane;
            const bool act = lane < n;
            const uint4 cbw = cb_next;
            if (j + 64 < jb_end) cb_next = load_cb(j + 64);
            float2 md0 = make_float2(XD_MINUS_INF, XD_MINUS_INF);
            float sj = 0.0f;
            if (act) { md0 = MD(j); sj = sub_w(cbw); }
            const float Mjm1 = xdw_shr1(md0.x, M0);              // M'[j-1] (SavedM0 of the cell)
            const float d_cur = md0.y;
            const float mi = Mjm1 + Open;                        // = md of the delete state, = mi of the insert state
            // insert state after each column: v[l] = max over k <= l of f^(l-k)(mi[k]) and of f^(l+1)(I0), f(x) = x + Ext applied
            // add by add (the reference's roundings); doubling steps, a lane without a source keeps its value (f^d(v) <= v)
            float v = mi;
            { const float x = I0 + Ext; if (lane == 0) v = x > mi ? x : mi; }
#pragma unroll
            for (int sft = 0; sft < 6; ++sft) {
                float t = __shfl_up(v, 1u << sft, 64);
#pragma unroll
                for (int q = 0; q < (1 << sft); ++q) t += Ext;
                v = t > v ? t : v;
            }
            const float I0in = xdw_shr1(v, I0);                  // insert state entering the cell
            uint32_t bits = 0;
            float xM = Mjm1;
            if (d_cur > xM) { xM = d_cur; bits = XD_DM; }
            if (I0in > xM) { xM = I0in; bits = XD_IM; }
            float s = sj;
            s += xM;
            // BestScore as the cell sees it: before / after its own match update
            const float sv = act ? s : -3.0e38f;
            float incl = sv;
            // inclusive max scan on DPP operands: within rows of 16 lanes, then across the rows (no LDS round trips)
#define XDW_SCAN_STEP(ctrl, rows) { const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -3.0e38f), \
                                        __builtin_bit_cast(int, incl), ctrl, rows, 0xF, false)); incl = t > incl ? t : incl; }
            XDW_SCAN_STEP(0x111, 0xF) XDW_SCAN_STEP(0x112, 0xF) XDW_SCAN_STEP(0x114, 0xF) XDW_SCAN_STEP(0x118, 0xF)      // row_shr:1,2,4,8
            XDW_SCAN_STEP(0x142, 0xA) XDW_SCAN_STEP(0x143, 0xC)                                                          // row_bcast:15, row_bcast:31
#undef XDW_SCAN_STEP
            const float excl = xdw_shr1(incl, -3.0e38f);
            const float pm = excl > BestScore ? excl : BestScore;
            const float pmi = incl > BestScore ? incl : BestScore;
            const float h = s - pm + X;
            // DELETE
            float d_new = d_cur;
            bool E2 = false;
            const bool notfirst = j != jlo;
            if (notfirst) {
                float d = d_cur;
                d += Ext;
                if (mi >= d) { d = mi; bits |= XD_MD; }
                d_new = d;
                const float hd = d - pmi + X;
                E2 = hd > 0;
            }
            // INSERT
            {
                const float fI = I0in + Ext;
                if (mi >= fI) bits |= XD_MI;
            }
            const float hi = v - pmi + X;
            const bool E1 = act && h > 0, E3 = act && hi > 0;
            E2 = E2 && act;
            // next row's band: the candidates are j - 1, j, j + 1 of the cells with an event, so the min / max over the lanes
            // are the first / last lanes of the event masks
            const bool E0 = act && h > AbsOpen;
            const unsigned long long m1 = __ballot(E1), m0 = __ballot(E0), m2 = __ballot(E2), m3 = __ballot(E3);
            uint32_t lo_c = U;
            if (m1 | m3) lo_c = jc + (uint32_t) __builtin_ctzll(m1 | m3) + 1;
            if (m0) lo_c = min(lo_c, jc + (uint32_t) __builtin_ctzll(m0));
            if (m2) lo_c = min(lo_c, jc + (uint32_t) __builtin_ctzll(m2) - 1);
            next_jlo = min(next_jlo, lo_c);
            {
                uint32_t l1 = 0;
                unsigned long long keep = ~0ull;                 // events that count: those of the last assigning cell and after it
                if (m1) { l1 = 63u - (uint32_t) __builtin_clzll(m1); keep = ~0ull << l1; }
                uint32_t e23 = 0;
                if (m2 & keep) e23 = jc + 63u - (uint32_t) __builtin_clzll(m2 & keep) - 1;
                if (m3 & keep) e23 = max(e23, jc + 63u - (uint32_t) __builtin_clzll(m3 & keep) + 1);
                if (m1) next_jhi = max(jc + l1 + 1, e23);
                else if (next_jhi != U) next_jhi = max(next_jhi, e23);
            }
            // best cell: the last one that reaches the maximum (s >= BestScore updates, xdropfwd.cpp)
            const float mx = xdw_bcast(incl, 63);
            if (mx >= BestScore) {
                BestScore = mx;
                Besti = i;
                Bestj = jc + 63u - (uint32_t) __builtin_clzll(__ballot(act && s == mx));
            }
            if (act) { MD(j) = make_float2(s, d_new); TB[tb_row + j] = (uint8_t) bits; }
            M0 = xdw_bcast(md0.x, n - 1);
            I0 = xdw_bcast(v, n - 1);
            hL = xdw_bcast(h, n - 1);                            // of column jlast after the last step
            hiL = xdw_bcast(hi, n - 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // the cell of column jlast may extend the row (xdropfwd.cpp: the two "j == jhi" tests of a cell): the columns that
        // enter the band are cleared as the reference clears them, including its overwrite of the cell's own fresh Mrow[j]
        if (hL > AbsExt && jhi + 1 < LB) {
            ++jhi;
            if (RING && jhi - jlo + 5 > XDW_RING) return false;
            const uint32_t new_endj = max(min(jhi + 1, LB), endj);
            for (uint32_t j2 = endj + 1; j2 <= new_endj; ++j2) {
                if (j2 - 1 > jlast) MD(j2 - 1).x = XD_MINUS_INF;
                MD(j2).y = XD_MINUS_INF;
            }
            endj = new_endj;
        } else if (hiL > AbsExt && jhi + 1 < LB) {
            ++jhi;
            if (RING && jhi - jlo + 5 > XDW_RING) return false;
            const uint32_t new_endj = max(min(jhi + 1, LB), endj);
            for (uint32_t j2 = endj + 1; j2 <= new_endj; ++j2) {
                MD(j2 - 1).x = XD_MINUS_INF;                     // j2 - 1 == jlast: Mrow[jlast] just stored is lost, as in the reference
                MD(j2).y = XD_MINUS_INF;
            }
            endj = new_endj;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        XDW_T(t_body);
        XDW_ADD(0, t_body - t_row); XDW_ADD(4, 1); XDW_ADD(5, (jb_end - jlo + 63) / 64);
        // ---- the cells the row grows by: the reference's loop, run uniformly ----
        for (uint32_t j = jb_end; j <= jhi; ++j) {
            uint8_t TraceBits = 0;
            const float SavedM0 = M0;
            const float2 md0 = MD(j);
            float m_new, d_cur = md0.y;
            float xM = M0;
            if (d_cur > xM) { xM = d_cur; TraceBits = XD_DM; }
            if (I0 > xM) { xM = I0; TraceBits = XD_IM; }
            M0 = md0.x;
            float s = sub_w(cb_t0);
            cb_t0 = cb_t1;
            cb_t1 = load_cb(min(j + 2, LB));
            s += xM;
            m_new = s;
            const float h = s - BestScore + X;
            if (h > 0) { next_jlo = min(next_jlo, j + 1); next_jhi = j + 1; }
            if (h > AbsOpen) next_jlo = min(next_jlo, j);
            if (h > AbsExt && j == jhi && jhi + 1 < LB) {        // match-insert may extend the current row
                ++jhi;
                if (RING && jhi - jlo + 5 > XDW_RING) return false;
                const uint32_t new_endj = max(min(jhi + 1, LB), endj);
                for (uint32_t j2 = endj + 1; j2 <= new_endj; ++j2) {
                    if (j2 - 1 > j) MD(j2 - 1).x = XD_MINUS_INF;
                    if (j2 == j) d_cur = XD_MINUS_INF;
                    else MD(j2).y = XD_MINUS_INF;
                }
                endj = new_endj;
            }
            if (s >= BestScore) { BestScore = s; Besti = i; Bestj = j; }
            float d_new = d_cur;
            if (j != jlo) {
                const float md = SavedM0 + Open;
                float d = d_cur;
                d += Ext;
                if (md >= d) { d = md; TraceBits |= XD_MD; }
                d_new = d;
                const float hd = d - BestScore + X;
                if (hd > 0) { next_jlo = min(next_jlo, j - 1); next_jhi = max(next_jhi, j - 1); }
            }
            {
                const float mi = SavedM0 + Open;
                I0 += Ext;
                if (mi >= I0) { I0 = mi; TraceBits |= XD_MI; }
                const float hi = I0 - BestScore + X;
                if (hi > 0) { next_jlo = min(next_jlo, j + 1); next_jhi = max(next_jhi, j + 1); }
                if (hi > AbsExt && j == jhi && jhi + 1 < LB) {
                    ++jhi;
                    if (RING && jhi - jlo + 5 > XDW_RING) return false;
                    const uint32_t new_endj = max(min(jhi + 1, LB), endj);
                    for (uint32_t j2 = endj + 1; j2 <= new_endj; ++j2) {
                        if (j2 - 1 == j) m_new = XD_MINUS_INF;
                        else MD(j2 - 1).x = XD_MINUS_INF;
                        if (j2 == j) d_new = XD_MINUS_INF;
                        else MD(j2).y = XD_MINUS_INF;
                    }
                    endj = new_endj;
                }
            }
            MD(j) = make_float2(m_new, d_new);
            TB[tb_row + j] = TraceBits;
        }
        XDW_T(t_tail);
        XDW_ADD(1, t_tail - t_body); XDW_ADD(6, jhi - jb_end + 1);
        if (jhi < LB) {                                             // end of Drow[]
            const uint32_t jhi1 = jhi + 1;
            uint8_t t = 0;
            const float md = M0 + Open;
            float d = MD(jhi1).y;
            d += Ext;
            if (md >= d) { d = md; t = XD_MD; }
            MD(jhi1).y = d;
            TB[tb_row + jhi1] = t;
        }
        if (next_jlo == U) break;
        prev_jlo = jlo; prev_jhi = jhi;
        jlo = next_jlo; jhi = next_jhi;
        if (jlo > LB) jlo = LB;
        if (jhi > LB) jhi = LB;
        if (jlo == prev_jlo) { M0 = XD_MINUS_INF; MD(jlo).y = XD_MINUS_INF; }
        else M0 = MD(jlo - 1).x;
    }
    if (BestScore <= 0.0f) { a.score[e] = 0.0f; return true; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    XDW_T(t_tb0);
    // traceback (xdropfwd.cpp:10-67).  The walk is a chain of dependent byte reads; the wave fetches an 8 x 8 window of trace
    // cells (one per lane) whose bottom-right corner is the cell asked for -- the walk moves up and left -- and serves the
    // following steps from registers: one memory round trip per ~6 steps instead of one per step.
    uint32_t wi = 0, wj = 0, win = 0;
    bool have = false;
    auto tbyte = [&](uint32_t r, uint32_t c) -> uint32_t {
        if (!have || r < wi || r > wi + 7 || c < wj || c > wj + 7) {
            wi = r >= 7 ? r - 7 : 0;
            wj = c >= 7 ? c - 7 : 0;
            win = TB[(size_t) (wi + (lane >> 3)) * Cols + wj + (lane & 7)];      // rows <= max(r, 7), columns <= max(c, 7): inside the slot
            have = true;
        }
        return (uint32_t) __builtin_amdgcn_readlane((int) win, (int) ((r - wi) * 8 + (c - wj)));
    };
    uint32_t i = Besti, j = Bestj, n = 0;
    char State = 'M';
    for (;;) {
        if (dir) slot[n] = State;
        else slot[cap - 1 - n] = State;
        ++n;
        if (i == 1 || j == 1) break;
        char Next;
        if (State == 'M') {
            const uint32_t c = tbyte(i, j);
            Next = (c & XD_DM) ? 'D' : (c & XD_IM) ? 'I' : 'M';
            --i; --j;
        } else if (State == 'D') {
            Next = (tbyte(i, j + 1) & XD_MD) ? 'M' : 'D';
            --i;
        } else {
            Next = (tbyte(i + 1, j) & XD_MI) ? 'M' : 'I';
            --j;
        }
        State = Next;
    }
    XDW_T(t_tb1);
    XDW_ADD(2, t_tb1 - t_tb0); XDW_ADD(7, n); XDW_ADD(8, 1);
    a.path_start[e] = dir ? 0 : cap - n;
    a.path_len[e] = n;
    a.score[e] = BestScore;
    return true;
#undef MD
}

template <bool EXPLICIT>
__global__ __launch_bounds__(64 * XDW_WAVES) void k_xdrop_wave(xd_args a)
{
    __shared__ float tab[XD_TABLE_FLOATS];
    __shared__ float2 ring[XDW_WAVES][XDW_RING];
    if (!EXPLICIT) {
        for (int i = threadIdx.x; i < XD_TABLE_FLOATS; i += blockDim.x) tab[i] = c_xd_tables.t[i];
        __syncthreads();
    }
    // the wave's extension: one value for its 64 lanes (readfirstlane tells the compiler so: counts and loop bounds derived
    // from it stay in SGPRs, loop tests are scalar)
    const uint32_t wv = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t e = blockIdx.x * XDW_WAVES + wv;
    const uint32_t lane = threadIdx.x & 63;
    if (e >= 2 * a.nreq) return;
    if (!xdw_extend<true, EXPLICIT>(a, tab, ring[wv], e, lane)) xdw_extend<false, EXPLICIT>(a, tab, nullptr, e, lane);
}

static void xd_launch(rsk_ctx *ctx, const xd_args &xa, size_t nreq)
{
    hipLaunchKernelGGL(k_xdrop_wave<false>, dim3((unsigned) ((2 * nreq + XDW_WAVES - 1) / XDW_WAVES)), dim3(64 * XDW_WAVES), 0, ctx->stream, xa);
}

// ---------------------------------------------------------------------------------------------------------------------
// rsk_xdrop_fwd / rsk_xdrop_bwd: ONE extension on an explicit LA x LB score matrix -- the form the reference's own
// self-test drives XDropFwd / XDropBwd in (test_xdrop.cpp:81-175), so its vectors pin this kernel directly.
// ---------------------------------------------------------------------------------------------------------------------
static int xd_explicit(rsk_ctx *ctx, const char *who, const float *S, uint32_t LA, uint32_t LB, float X, float open, float ext, uint32_t lo_a,
                       uint32_t lo_b, int dir, float *score, char *path, size_t path_cap, uint32_t *path_len)
{
    if (!ctx || !S || !score || !path) { rsk_set_error("%s: NULL argument", who); return RSK_E_INVALID; }
    if (open > 0 || ext > 0) { rsk_set_error("%s: gap penalties must be <= 0", who); return RSK_E_INVALID; }
    // forward: extents LA - lo_a, LB - lo_b >= 1; backward: lo_a, lo_b >= 1 (the reference asserts the same, xdropbwd.cpp:37-38)
    if (lo_a > LA || lo_b > LB || (dir == 0 && (lo_a >= LA || lo_b >= LB)) || (dir == 1 && (lo_a == 0 || lo_b == 0))) {
        rsk_set_error("%s: start (%u, %u) outside the %u x %u matrix", who, lo_a, lo_b, LA, LB);
        return RSK_E_INVALID;
    }
    RSK_HIP(hipSetDevice(ctx->device));
    const uint64_t EA = dir ? lo_a : LA - lo_a, EB = dir ? lo_b : LB - lo_b;
    const size_t tb_bytes = (size_t) ((EA + 9) * (EB + 9) + 15) & ~(size_t) 15, row_floats = 2 * (EB + 9), slot = EA + EB + 2;
    struct ws_t {
        rsk_ctx *ctx;
        std::vector<void *> all;
        ~ws_t() { for (void *p : all) rsk_pool_free(ctx, p); }
    } ws{ ctx, {} };
    auto dalloc = [&](void **p, size_t bytes) -> int {
        int r = rsk_pool_alloc(ctx, p, std::max<size_t>(bytes, 16));
        if (r == RSK_OK) ws.all.push_back(*p);
        return r;
    };
    float *d_S, *d_rows, *d_score;
    uint32_t *d_u;                    // a_len, b_len, ia = ib = 0, lo_a, lo_b, path_start[2], path_len[2]
    uint64_t *d_off;                  // row_off[2], tb_off[2], path_off[2] (both directions share the slots: only one runs)
    uint8_t *d_tb;
    char *d_paths;
    int rc;
    if ((rc = dalloc((void **) &d_S, (size_t) LA * LB * 4)) || (rc = dalloc((void **) &d_rows, row_floats * 4)) || (rc = dalloc((void **) &d_score, 8)) ||
        (rc = dalloc((void **) &d_u, 9 * 4)) || (rc = dalloc((void **) &d_off, 6 * 8)) || (rc = dalloc((void **) &d_tb, tb_bytes)) ||
        (rc = dalloc((void **) &d_paths, slot)))
        return rc;
    const uint32_t h_u[9] = { LA, LB, 0, lo_a, lo_b, 0, 0, 0, 0 };
    const uint64_t h_off[6] = { 0, 0, 0, 0, 0, 0 };
    RSK_HIP(hipMemcpyAsync(d_S, S, (size_t) LA * LB * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_u, h_u, sizeof(h_u), hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_tb, 0, tb_bytes, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_rows, 0, row_floats * 4, ctx->stream));
    xd_args a = {};
    a.a_len = d_u; a.b_len = d_u + 1; a.ia = d_u + 2; a.ib = d_u + 2; a.lo_a = d_u + 3; a.lo_b = d_u + 4;
    a.path_start = d_u + 5; a.path_len = d_u + 7;
    a.nreq = 1; a.X = X; a.open = open; a.ext = ext;
    a.rows = d_rows; a.row_off = d_off; a.tb = d_tb; a.tb_off = d_off + 2; a.score = d_score; a.paths = d_paths; a.path_off = d_off + 4;
    a.smx = d_S; a.smx_ld = LB; a.only_dir = dir;
    hipLaunchKernelGGL(k_xdrop_wave<true>, dim3(1), dim3(64 * XDW_WAVES), 0, ctx->stream, a);
    RSK_HIP(hipGetLastError());
    float h_score[2];
    uint32_t h_res[4];
    RSK_HIP(hipMemcpyAsync(h_score, d_score, 8, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(h_res, d_u + 5, 16, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t len = h_res[2 + dir];
    *score = h_score[dir];
    if (path_len) *path_len = len;
    if ((size_t) len + 1 > path_cap) { rsk_set_error("%s: path buffer too small (%u needed)", who, len + 1); return RSK_E_INVALID; }
    if (len) {
        RSK_HIP(hipMemcpyAsync(path, d_paths + h_res[dir], len, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipStreamSynchronize(ctx->stream));
    }
    path[len] = 0;
    return RSK_OK;
}

extern "C" int rsk_xdrop_fwd(rsk_ctx *ctx, const float *S, uint32_t LA, uint32_t LB, float X, float open, float ext, uint32_t lo_a, uint32_t lo_b,
                             float *score, char *path, size_t path_cap, uint32_t *path_len)
{
    return xd_explicit(ctx, "rsk_xdrop_fwd", S, LA, LB, X, open, ext, lo_a, lo_b, 0, score, path, path_cap, path_len);
}

// XDropBwd(HiA, HiB) extends from (hi_a, hi_b) towards the starts = the backward extension of the start (hi_a + 1, hi_b + 1)
extern "C" int rsk_xdrop_bwd(rsk_ctx *ctx, const float *S, uint32_t LA, uint32_t LB, float X, float open, float ext, uint32_t hi_a, uint32_t hi_b,
                             float *score, char *path, size_t path_cap, uint32_t *path_len)
{
    if (hi_a >= LA || hi_b >= LB) { rsk_set_error("rsk_xdrop_bwd: (%u, %u) outside the %u x %u matrix", hi_a, hi_b, LA, LB); return RSK_E_INVALID; }
    return xd_explicit(ctx, "rsk_xdrop_bwd", S, LA, LB, X, open, ext, hi_a + 1, hi_b + 1, 1, score, path, path_cap, path_len);
}

// XDropHSP's two extensions (xdrophsp.cpp:97-108) for a list of seeded pairs.  Host arrays in, host arrays out.
// The path of an extension fills a small part of its worst-case slot: the paths are packed back to back on the device
// (sizes scanned, one wave copies one path) and only the packed bytes cross PCIe.
__global__ void k_xd_sizes(const uint32_t *path_len, uint32_t n2, uint64_t *sizes)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= n2) sizes[e] = e < n2 ? path_len[e] : 0;
}
__global__ void k_xd_pack(const char *paths, const uint64_t *path_off, const uint32_t *path_start, const uint32_t *path_len, const uint64_t *out_off,
                          uint32_t n2, char *out)
{
    const uint32_t e = blockIdx.x * (blockDim.x >> 6) + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // wave-uniform
    if (e >= n2) return;
    const uint32_t len = path_len[e];
    const char *src = paths + path_off[e] + path_start[e];
    char *dst = out + out_off[e];
    for (uint32_t c = threadIdx.x & 63; c < len; c += 64) dst[c] = src[c];
}

extern "C" int rsk_xdrop_pairs(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, const uint32_t *ia, const uint32_t *ib,
                               const uint32_t *lo_a, const uint32_t *lo_b, size_t n, float X, float gap_open, float gap_ext,
                               float *score_fwd, float *score_bwd, char *paths, size_t paths_bytes, uint64_t *fwd_off,
                               uint32_t *fwd_len, uint64_t *bwd_off, uint32_t *bwd_len)
{
    if (!ctx || !dba || !dbb || (n && (!ia || !ib || !lo_a || !lo_b || !score_fwd || !score_bwd || !paths || !fwd_off || !fwd_len || !bwd_off || !bwd_len))) {
        rsk_set_error("rsk_xdrop_pairs: NULL argument");
        return RSK_E_INVALID;
    }
    if (!dba->d_prof_ra || !dbb->d_prof_cb) { rsk_set_error("rsk_xdrop_pairs: chain set has no profiles"); return RSK_E_INVALID; }
    if (gap_open > 0 || gap_ext > 0) { rsk_set_error("rsk_xdrop_pairs: gap penalties must be <= 0"); return RSK_E_INVALID; }
    if (n == 0) return RSK_OK;
    if (n > 0x3FFFFFFFull) { rsk_set_error("rsk_xdrop_pairs: too many pairs in one call"); return RSK_E_RANGE; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = xd_upload_tables(ctx);
    if (rc != RSK_OK) return rc;
    // per-extension slots; the forward extension needs lo < L on both chains, the backward one lo >= 1 (xdropbwd.cpp:37-38)
    size_t need = 0;
    for (size_t r = 0; r < n; ++r) {
        if (ia[r] >= dba->n || ib[r] >= dbb->n) { rsk_set_error("rsk_xdrop_pairs: pair %zu out of range", r); return RSK_E_INVALID; }
        const uint32_t LA = dba->len[ia[r]], LB = dbb->len[ib[r]];
        if (lo_a[r] == 0 || lo_b[r] == 0 || lo_a[r] >= LA || lo_b[r] >= LB) {
            rsk_set_error("rsk_xdrop_pairs: start (%u, %u) of pair %zu outside 1..L-1", lo_a[r], lo_b[r], r);
            return RSK_E_INVALID;
        }
        need += (size_t) LA + LB + 4;
    }
    if (paths_bytes < need) { rsk_set_error("rsk_xdrop_pairs: paths buffer too small (%zu < %zu)", paths_bytes, need); return RSK_E_INVALID; }

    struct ws_t {
        std::vector<void *> all;
        ~ws_t() { for (void *p : all) (void) hipFree(p); }
        int alloc(void **p, size_t bytes)
        {
            const int rc_ = rsk_dev_malloc(nullptr, p, bytes ? bytes : 16);
            if (rc_ == RSK_OK) all.push_back(*p);
            return rc_;
        }
    };
    // sub-batches bounded by the trace scratch (the rest is small)
    const uint64_t TB_BUDGET = 12ull << 30;            // below the block size from which a fresh hipMalloc costs ~30 ms per GB (cold calls)
    const bool trace = getenv("RSK_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // One workspace for the whole call, sized by the largest sub-batch: hipFree + hipMalloc of tens of GB between
    // sub-batches cost up to 0.8 s on some boxes.
    uint64_t max_ro = 0, max_to = 0, max_po = 0;
    size_t max_m = 0;
    for (size_t q0 = 0; q0 < n;) {
        uint64_t ro = 0, to = 0, po = 0;
        size_t q1 = q0;
        while (q1 < n) {
            const uint32_t LA = dba->len[ia[q1]], LB = dbb->len[ib[q1]];
            const uint32_t ext[2][2] = { { LA - lo_a[q1], LB - lo_b[q1] }, { lo_a[q1], lo_b[q1] } };
            const uint64_t t_need = (uint64_t) (ext[0][0] + 9) * (ext[0][1] + 9) + (uint64_t) (ext[1][0] + 9) * (ext[1][1] + 9) + 64;
            if (q1 > q0 && to + t_need > TB_BUDGET) break;
            for (int d = 0; d < 2; ++d) {
                ro += 2ull * (ext[d][1] + 9);
                to += ((uint64_t) (ext[d][0] + 9) * (ext[d][1] + 9) + 15) & ~15ull;
                po += ext[d][0] + ext[d][1] + 2;
            }
            ++q1;
        }
        max_ro = std::max(max_ro, ro); max_to = std::max(max_to, to); max_po = std::max(max_po, po);
        max_m = std::max(max_m, q1 - q0);
        q0 = q1;
    }
    const auto t_alloc0 = now();
    ws_t ws;
    uint32_t *d_ia, *d_ib, *d_la, *d_lb, *d_pstart, *d_plen;
    uint64_t *d_rowoff, *d_tboff, *d_pathoff;
    float *d_rows, *d_score;
    uint8_t *d_tb;
    char *d_paths, *d_packed;
    uint64_t *d_sizes, *d_outoff;
    void *d_scan = nullptr;
    size_t scan_bytes = 0;
    RSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint64_t *) nullptr, (uint64_t *) nullptr, (int) (2 * max_m + 1), ctx->stream));
    {
        const size_t m = max_m;
        if ((rc = ws.alloc((void **) &d_packed, max_po + 16)) || (rc = ws.alloc((void **) &d_sizes, (2 * m + 1) * 8)) ||
            (rc = ws.alloc((void **) &d_outoff, (2 * m + 1) * 8)) || (rc = ws.alloc(&d_scan, scan_bytes)))
            return rc;
        if ((rc = ws.alloc((void **) &d_ia, m * 4)) || (rc = ws.alloc((void **) &d_ib, m * 4)) || (rc = ws.alloc((void **) &d_la, m * 4)) ||
            (rc = ws.alloc((void **) &d_lb, m * 4)) || (rc = ws.alloc((void **) &d_pstart, 2 * m * 4)) || (rc = ws.alloc((void **) &d_plen, 2 * m * 4)) ||
            (rc = ws.alloc((void **) &d_rowoff, 2 * m * 8)) || (rc = ws.alloc((void **) &d_tboff, 2 * m * 8)) ||
            (rc = ws.alloc((void **) &d_pathoff, 2 * m * 8)) || (rc = ws.alloc((void **) &d_rows, max_ro * 4)) || (rc = ws.alloc((void **) &d_score, 2 * m * 4)) ||
            (rc = ws.alloc((void **) &d_tb, max_to)) || (rc = ws.alloc((void **) &d_paths, max_po)))
            return rc;
    }
    if (trace) fprintf(stderr, "[rsk_xdrop_pairs] workspace %.2f GB allocated in %.1f ms\n", (max_to + max_ro * 4 + max_po) / 1073741824.0, ms(t_alloc0, now()));
    size_t r0 = 0;
    uint64_t poff = 0;                // running offset into the caller's paths buffer
    while (r0 < n) {
        std::vector<uint64_t> row_off, tb_off, path_off;
        uint64_t ro = 0, to = 0, po = 0;
        size_t r1 = r0;
        while (r1 < n) {
            const uint32_t LA = dba->len[ia[r1]], LB = dbb->len[ib[r1]];
            const uint32_t ext[2][2] = { { LA - lo_a[r1], LB - lo_b[r1] }, { lo_a[r1], lo_b[r1] } };
            const uint64_t t_need = (uint64_t) (ext[0][0] + 9) * (ext[0][1] + 9) + (uint64_t) (ext[1][0] + 9) * (ext[1][1] + 9) + 64;
            if (r1 > r0 && to + t_need > TB_BUDGET) break;
            for (int d = 0; d < 2; ++d) {
                row_off.push_back(ro); ro += 2ull * (ext[d][1] + 9);
                tb_off.push_back(to); to += ((uint64_t) (ext[d][0] + 9) * (ext[d][1] + 9) + 15) & ~15ull;
                path_off.push_back(po); po += ext[d][0] + ext[d][1] + 2;
            }
            ++r1;
        }
        const size_t m = r1 - r0;
        const auto t_a = now();
        const auto t_b = t_a;
        RSK_HIP(hipMemcpyAsync(d_ia, ia + r0, m * 4, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_ib, ib + r0, m * 4, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_la, lo_a + r0, m * 4, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_lb, lo_b + r0, m * 4, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_rowoff, row_off.data(), 2 * m * 8, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_tboff, tb_off.data(), 2 * m * 8, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_pathoff, path_off.data(), 2 * m * 8, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemsetAsync(d_tb, 0, to, ctx->stream));          // unwritten trace cells read as 0 (XDPMem zeroes its matrix)
        RSK_HIP(hipMemsetAsync(d_rows, 0, ro * 4, ctx->stream));    // as XDPMem::Alloc leaves its rows
        xd_args a = {};
        a.a_ra = dba->d_prof_ra; a.b_cb = dbb->d_prof_cb;
        a.a_off = dba->d_off; a.b_off = dbb->d_off; a.a_len = dba->d_len; a.b_len = dbb->d_len;
        a.ia = d_ia; a.ib = d_ib; a.lo_a = d_la; a.lo_b = d_lb;
        a.nreq = (uint32_t) m;
        a.X = X; a.open = gap_open; a.ext = gap_ext;
        a.rows = d_rows; a.row_off = d_rowoff; a.tb = d_tb; a.tb_off = d_tboff;
        a.score = d_score; a.paths = d_paths; a.path_off = d_pathoff; a.path_start = d_pstart; a.path_len = d_plen;
        a.only_dir = -1;
        if (trace) RSK_HIP(hipStreamSynchronize(ctx->stream));
        const auto t_c = now();
        xd_launch(ctx, a, m);
        RSK_HIP(hipGetLastError());
        if (trace) RSK_HIP(hipStreamSynchronize(ctx->stream));
        const auto t_d = now();
        const uint32_t n2 = (uint32_t) (2 * m);
        hipLaunchKernelGGL(k_xd_sizes, dim3((n2 + 256) / 256), dim3(256), 0, ctx->stream, d_plen, n2, d_sizes);
        RSK_HIP(hipcub::DeviceScan::ExclusiveSum(d_scan, scan_bytes, d_sizes, d_outoff, (int) (n2 + 1), ctx->stream));
        hipLaunchKernelGGL(k_xd_pack, dim3((n2 + 3) / 4), dim3(256), 0, ctx->stream, d_paths, d_pathoff, d_pstart, d_plen, d_outoff, n2, d_packed);
        RSK_HIP(hipGetLastError());
        std::vector<float> h_score(2 * m);
        std::vector<uint32_t> h_plen(2 * m);
        std::vector<uint64_t> h_outoff(2 * m + 1);
        RSK_HIP(hipMemcpyAsync(h_score.data(), d_score, 2 * m * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipMemcpyAsync(h_plen.data(), d_plen, 2 * m * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipMemcpyAsync(h_outoff.data(), d_outoff, (2 * m + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipStreamSynchronize(ctx->stream));
        const uint64_t packed = h_outoff[2 * m];
        if (poff + packed > paths_bytes) { rsk_set_error("rsk_xdrop_pairs: paths buffer too small"); return RSK_E_INVALID; }
        if (packed) {
            RSK_HIP(hipMemcpyAsync(paths + poff, d_packed, packed, hipMemcpyDeviceToHost, ctx->stream));   // (a null-stream copy would wait for every other stream)
            RSK_HIP(hipStreamSynchronize(ctx->stream));
        }
        for (size_t k = 0; k < m; ++k) {
            score_fwd[r0 + k] = h_score[2 * k];
            score_bwd[r0 + k] = h_score[2 * k + 1];
            fwd_off[r0 + k] = poff + h_outoff[2 * k];
            fwd_len[r0 + k] = h_plen[2 * k];
            bwd_off[r0 + k] = poff + h_outoff[2 * k + 1];
            bwd_len[r0 + k] = h_plen[2 * k + 1];
        }
        po = packed;
        if (trace)
            fprintf(stderr, "[rsk_xdrop_pairs] %zu pairs: trace scratch %.2f GB, alloc %.1f ms, h2d + clear %.1f ms, kernel %.1f ms, d2h %.1f ms\n", m,
                    to / 1073741824.0, ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d), ms(t_d, now()));
        poff += po;
        r0 = r1;
    }
    return RSK_OK;
}


// ---------------------------------------------------------------------------------------------
// rsk_mkf_align_pairs: everything of the long-chain path after the chaining of the seed HSPs in ONE device batch
//   PostAlignMKF dssaligner.cpp:1395-1430 (GetMegaHSPScore :488 of every chained HSP, the MinMegaHSPScore gate, best HSP),
//   XDropHSP xdrophsp.cpp:42-117 (best 8-mer of that HSP = start, XDropFwd + XDropBwd, TotalScore < 10 => no alignment),
//   MergeFwdBwd mergefwdback.cpp:6, CalcEvalue dssaligner.cpp:852 (LDDT, test statistic, E-value).
// The chaining of the seed HSPs (MuKmerFilter::ChainHSPs mukmerfilter.cpp:391 -> Chainer::Chain chainer.cpp:31) runs on the
// device as well (k_mkf_chain, rsk_mkf_chain_align_pairs) for every pair whose outcome is defined: the reference sorts the
// interval end points with libc qsort on a comparator that returns 0 for two end points of the same kind at the same
// position (chainer.cpp:11-29), and the sweep takes the FIRST of two intervals that end at one position with equal chain
// scores -- whichever qsort left in front.  Those pairs (status 3) go back to the caller, which chains them with the
// same libc qsort on the host (rsk_mkf_align_pairs takes chained lists).
// ---------------------------------------------------------------------------------------------
struct mkfa_args {
    const uint16_t *a_ra, *b_cb;
    const uint32_t *a_off, *b_off, *a_len, *b_len;
    const uint32_t *ia, *ib;
    const uint32_t *hsp_first;       // [npairs + 1]
    const uint32_t *hsp_cnt;         // optional: HSPs of pair p = hsp_cnt[p] entries from hsp_first[p] (lists chained on the device)
    const int32_t *hsp_lo_a, *hsp_lo_b, *hsp_len;
    uint32_t npairs;
    float min_mega;
    uint8_t *valid;                  // 1 = extensions wanted, 0 = no alignment, 2 = start outside 1..L-1 (host decides)
    uint32_t *lo_a, *lo_b;           // start of the gapped extensions
};

// Chainer::Chain (chainer.cpp:31-176) over the seed HSPs of one pair, one thread per pair (a pair has at most MKF_CHAIN_MAX
// HSPs: the seeding keeps 32).  Intervals [Lo_i, Lo_i + Len - 1] on the query with the HSP's integer score: sweep over
// the 2N end points in order of position, starts before ends at one position; at a start the interval's chain score is
// its score plus the best chain that has ended; at an end it becomes the best ended chain if strictly better.  The
// chain is reported end -> start (the order ChainHSPs hands it to PostAlignMKF, whose float sums follow it).
// The pair's HSP list is rewritten in place to that chain, cnt[p] = its length, 0 if the chain's total score is <= 0
// (PostAlignMKF dssaligner.cpp:1397).  valid[p] = 3 marks a sweep whose outcome depends on the order of two equal end points.
#define MKF_CHAIN_MAX 64
__global__ __launch_bounds__(64) void k_mkf_chain(const uint32_t *first, int32_t *lo_a, int32_t *lo_b, int32_t *len, const int32_t *score,
                                                 uint32_t npairs, uint32_t *cnt, uint8_t *valid)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const uint32_t h0 = first[p], N = first[p + 1] - h0;
    valid[p] = 0;
    cnt[p] = 0;
    if (N == 0) return;
    if (N > MKF_CHAIN_MAX) { valid[p] = 3; return; }
    // end points as sortable keys: position, then starts before ends, then the HSP's index
    uint32_t key[2 * MKF_CHAIN_MAX];
    for (uint32_t i = 0; i < N; ++i) {
        const uint32_t lo = (uint32_t) lo_a[h0 + i], hi = lo + (uint32_t) len[h0 + i] - 1;
        key[2 * i] = (lo << 8) | i;
        key[2 * i + 1] = (hi << 8) | 0x80u | i;
    }
    for (uint32_t i = 1; i < 2 * N; ++i) {          // insertion sort: the lists are a few entries long
        const uint32_t k = key[i];
        uint32_t j = i;
        for (; j > 0 && key[j - 1] > k; --j) key[j] = key[j - 1];
        key[j] = k;
    }
    float cs[MKF_CHAIN_MAX];
    uint32_t tb[MKF_CHAIN_MAX], endpos[MKF_CHAIN_MAX];
    const uint32_t NONE = 0xFFFFFFFFu;
    uint32_t best_end = NONE;
    bool tied = false;
    for (uint32_t e = 0; e < 2 * N; ++e) {
        const uint32_t i = key[e] & 0x7Fu, pos = key[e] >> 8;
        if (!(key[e] & 0x80u)) {
            tb[i] = best_end;
            cs[i] = best_end == NONE ? (float) score[h0 + i] : cs[best_end] + (float) score[h0 + i];
        } else {
            endpos[i] = pos;
            if (best_end == NONE || cs[i] > cs[best_end]) best_end = i;
            else if (cs[i] == cs[best_end] && endpos[best_end] == pos) tied = true;      // which of the two qsort leaves first decides
        }
    }
    if (tied) { valid[p] = 3; return; }
    // chain end -> start, then the list in place (the entries are read before they are overwritten: copy out first)
    int32_t ca[MKF_CHAIN_MAX], cb[MKF_CHAIN_MAX], cl[MKF_CHAIN_MAX];
    uint32_t n = 0;
    float total = 0;
    for (uint32_t i = best_end; i != NONE; i = tb[i]) {
        total += (float) score[h0 + i];
        ca[n] = lo_a[h0 + i]; cb[n] = lo_b[h0 + i]; cl[n] = len[h0 + i];
        ++n;
    }
    if ((int) total <= 0) return;
    for (uint32_t k = 0; k < n; ++k) { lo_a[h0 + k] = ca[k]; lo_b[h0 + k] = cb[k]; len[h0 + k] = cl[k]; }
    cnt[p] = n;
}

// one thread per pair; every sum in the reference's order
__global__ __launch_bounds__(256) void k_mkf_start(mkfa_args a)
{
    __shared__ float tab[XD_TABLE_FLOATS];
    for (int i = threadIdx.x; i < XD_TABLE_FLOATS; i += blockDim.x) tab[i] = c_xd_tables.t[i];
    __syncthreads();
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.npairs) return;
    const uint32_t A = a.ia[p], B = a.ib[p];
    const uint16_t *RA = a.a_ra + (size_t) a.a_off[A] * 8, *CB = a.b_cb + (size_t) a.b_off[B] * 8;
    const char *tabb = (const char *) tab;
    const uint32_t toffb[8] = { 0 * 4, 400 * 4, 656 * 4, 912 * 4, 1168 * 4, 1424 * 4, 1680 * 4, 1936 * 4 };
    const uint32_t h0 = a.hsp_first[p], h1 = a.hsp_cnt ? h0 + a.hsp_cnt[p] : a.hsp_first[p + 1];
    if (a.hsp_cnt && a.valid[p] == 3) return;     // k_mkf_chain: tied chain, the caller decides
    a.valid[p] = 0; a.lo_a[p] = 0; a.lo_b[p] = 0;
    if (h1 == h0) return;
    float MegaTotal = 0, BestMega = 0;
    uint32_t BestIdx = h0;
    for (uint32_t h = h0; h < h1; ++h) {
        // GetMegaHSPScore dssaligner.cpp:488: Total += M_f[a][b] for f = 0..7 (outer), k = 0..Len-1 (inner)
        const uint32_t Li = (uint32_t) a.hsp_lo_a[h], Lj = (uint32_t) a.hsp_lo_b[h], Len = (uint32_t) a.hsp_len[h];
        float Total = 0;
        for (int f = 0; f < 8; ++f)
            for (uint32_t k = 0; k < Len; ++k)
                Total += *(const float *) (tabb + toffb[f] + RA[(size_t) (Li + k) * 8 + f] + CB[(size_t) (Lj + k) * 8 + f]);
        if (Total > BestMega) { BestMega = Total; BestIdx = h; }
        MegaTotal += Total;
    }
    if (MegaTotal < a.min_mega) return;
    // XDropHSP xdrophsp.cpp:42-95: start = the highest-scoring 8-mer of the best HSP (first one on ties), else its middle
    const uint32_t Li = (uint32_t) a.hsp_lo_a[BestIdx], Lj = (uint32_t) a.hsp_lo_b[BestIdx], Len = (uint32_t) a.hsp_len[BestIdx];
    uint32_t LoA = Li + Len / 2, LoB = Lj + Len / 2;
    auto sub = [&](uint32_t c) {
        float Total = 0.0f;
#pragma unroll
        for (int f = 0; f < 8; ++f) Total += *(const float *) (tabb + toffb[f] + RA[(size_t) (Li + c) * 8 + f] + CB[(size_t) (Lj + c) * 8 + f]);
        return Total;
    };
    if (Len >= 8) {
        float w0 = sub(0), w1 = sub(1), w2 = sub(2), w3 = sub(3), w4 = sub(4), w5 = sub(5), w6 = sub(6), w7;
        float BestMer = 0;
        for (uint32_t ms = 0; ms + 8 <= Len; ++ms) {
            w7 = sub(ms + 7);
            float Mer = 0;
            Mer += w0; Mer += w1; Mer += w2; Mer += w3; Mer += w4; Mer += w5; Mer += w6; Mer += w7;
            if (Mer > BestMer) { BestMer = Mer; LoA = Li + ms; LoB = Lj + ms; }
            w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6; w6 = w7;
        }
    }
    if (min(LoA, LoB) < 4) { LoA += 4; LoB += 4; }
    a.lo_a[p] = LoA; a.lo_b[p] = LoB;
    a.valid[p] = (LoA >= 1 && LoB >= 1 && LoA < a.a_len[A] && LoB < a.b_len[B]) ? 1 : 2;
}

struct mkfm_args {
    const uint32_t *a_len, *b_len, *ia, *ib;
    const uint8_t *valid;
    const uint32_t *req_lo_a, *req_lo_b;
    const float *xscore;             // [2 * npairs]
    const char *xpaths; const uint64_t *xpath_off; const uint32_t *xpath_start, *xpath_len;
    uint32_t npairs;
    char *mpaths; const uint64_t *mpath_off;     // merged paths: slot of LA + LB + 5 per pair
    float *score; uint32_t *lo_a, *lo_b; uint64_t *pstart; uint32_t *plen;
};

// MergeFwdBwd mergefwdback.cpp:6-26 + the TotalScore gate of XDropHSP (xdrophsp.cpp:111-115); one wave per pair
__global__ __launch_bounds__(256) void k_mkf_merge(mkfm_args a)
{
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // wave-uniform
    if (p >= a.npairs) return;
    const int lane = threadIdx.x & 63;
    const uint64_t mo = a.mpath_off[p];
    bool ok = a.valid[p] == 1;
    float total = 0;
    if (ok) {
        total = a.xscore[2 * p] + a.xscore[2 * p + 1];       // ScoreFwd + ScoreBwd
        ok = !(total < 10.0f);
    }
    if (!ok) {
        if (lane == 0) { a.score[p] = 0.0f; a.lo_a[p] = RSK_NO_POS; a.lo_b[p] = RSK_NO_POS; a.pstart[p] = mo; a.plen[p] = 0; a.mpaths[mo] = 0; }
        return;
    }
    const uint32_t A = a.ia[p], B = a.ib[p];
    const uint32_t LoA = a.req_lo_a[p], LoB = a.req_lo_b[p];
    const uint64_t FA = a.a_len[A] - LoA, FB = a.b_len[B] - LoB;
    const uint64_t fo = a.xpath_off[p], bo = fo + FA + FB + 2;                  // per-pair slot: forward share, then backward share
    const char *fp = a.xpaths + fo + a.xpath_start[2 * p], *bp = a.xpaths + bo + a.xpath_start[2 * p + 1];
    const uint32_t fl = a.xpath_len[2 * p], bl = a.xpath_len[2 * p + 1];
    uint32_t nMD = 0, nMI = 0;                                                   // columns of the backward path that consume A / B
    for (uint32_t c0 = 0; c0 < bl; c0 += 64) {
        const uint32_t c = c0 + lane;
        const char ch = c < bl ? bp[c] : 0;
        nMD += (uint32_t) __popcll(__ballot(ch == 'M' || ch == 'D'));
        nMI += (uint32_t) __popcll(__ballot(ch == 'M' || ch == 'I'));
        if (c < bl) a.mpaths[mo + c] = ch;
    }
    for (uint32_t c = lane; c < fl; c += 64) a.mpaths[mo + bl + c] = fp[c];
    if (lane == 0) {
        a.mpaths[mo + bl + fl] = 0;
        a.score[p] = total;
        a.lo_a[p] = LoA - nMD;                                                   // BwdHiA + 1 - (M + D), BwdHiA = LoA - 1 (= FwdLoA if the backward path is empty)
        a.lo_b[p] = LoB - nMI;
        a.pstart[p] = mo;
        a.plen[p] = bl + fl;
    }
}

static int mkf_batch(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, const uint32_t *ia, const uint32_t *ib, size_t npairs,
                     const uint32_t *hsp_first, const int32_t *hsp_lo_a, const int32_t *hsp_lo_b, const int32_t *hsp_len,
                     const int32_t *hsp_score /* non-NULL: unchained seed HSPs, chained on the device */, float x2, float gap_open, float gap_ext,
                     float min_mega_score, float min_fwd_score, rsk_aln *out, uint8_t *status, char *paths, size_t paths_bytes);

extern "C" int rsk_mkf_align_pairs(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, const uint32_t *ia, const uint32_t *ib, size_t npairs,
                                   const uint32_t *hsp_first, const int32_t *hsp_lo_a, const int32_t *hsp_lo_b, const int32_t *hsp_len, float x2,
                                   float gap_open, float gap_ext, float min_mega_score, float min_fwd_score, rsk_aln *out, uint8_t *status,
                                   char *paths, size_t paths_bytes)
{
    return mkf_batch(ctx, dba, dbb, ia, ib, npairs, hsp_first, hsp_lo_a, hsp_lo_b, hsp_len, nullptr, x2, gap_open, gap_ext, min_mega_score,
                     min_fwd_score, out, status, paths, paths_bytes);
}

extern "C" int rsk_mkf_chain_align_pairs(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, const uint32_t *ia, const uint32_t *ib, size_t npairs,
                                         const uint32_t *hsp_first, const int32_t *hsp_lo_a, const int32_t *hsp_lo_b, const int32_t *hsp_len,
                                         const int32_t *hsp_score, float x2, float gap_open, float gap_ext, float min_mega_score,
                                         float min_fwd_score, rsk_aln *out, uint8_t *status, char *paths, size_t paths_bytes)
{
    if (npairs && hsp_first && hsp_first[npairs] && !hsp_score) { rsk_set_error("rsk_mkf_chain_align_pairs: NULL HSP scores"); return RSK_E_INVALID; }
    static const int32_t none = 0;
    return mkf_batch(ctx, dba, dbb, ia, ib, npairs, hsp_first, hsp_lo_a, hsp_lo_b, hsp_len, hsp_score ? hsp_score : &none, x2, gap_open, gap_ext,
                     min_mega_score, min_fwd_score, out, status, paths, paths_bytes);
}

static int mkf_batch(rsk_ctx *ctx, const rsk_db *dba, const rsk_db *dbb, const uint32_t *ia, const uint32_t *ib, size_t npairs,
                     const uint32_t *hsp_first, const int32_t *hsp_lo_a, const int32_t *hsp_lo_b, const int32_t *hsp_len, const int32_t *hsp_score,
                     float x2, float gap_open, float gap_ext, float min_mega_score, float min_fwd_score, rsk_aln *out, uint8_t *status,
                     char *paths, size_t paths_bytes)
{
    if (!ctx || !dba || !dbb || (npairs && (!ia || !ib || !hsp_first || !out || !status || !paths))) { rsk_set_error("rsk_mkf_align_pairs: NULL argument"); return RSK_E_INVALID; }
    if (!dba->d_prof_ra || !dbb->d_prof_cb || !dba->d_x || !dbb->d_x) { rsk_set_error("rsk_mkf_align_pairs: chain sets need profiles and coordinates"); return RSK_E_INVALID; }
    if (gap_open > 0 || gap_ext > 0) { rsk_set_error("rsk_mkf_align_pairs: gap penalties must be <= 0"); return RSK_E_INVALID; }
    if (npairs == 0) return RSK_OK;
    if (npairs > 0x3FFFFFFFull) { rsk_set_error("rsk_mkf_align_pairs: too many pairs in one call"); return RSK_E_RANGE; }
    const size_t nh = hsp_first[npairs];
    if (nh && (!hsp_lo_a || !hsp_lo_b || !hsp_len)) { rsk_set_error("rsk_mkf_align_pairs: NULL HSP arrays"); return RSK_E_INVALID; }
    size_t need = 0;
    for (size_t p = 0; p < npairs; ++p) {
        if (ia[p] >= dba->n || ib[p] >= dbb->n || hsp_first[p + 1] < hsp_first[p]) { rsk_set_error("rsk_mkf_align_pairs: pair %zu out of range", p); return RSK_E_INVALID; }
        const uint32_t LA = dba->len[ia[p]], LB = dbb->len[ib[p]];
        for (uint32_t h = hsp_first[p]; h < hsp_first[p + 1]; ++h)
            if (hsp_lo_a[h] < 0 || hsp_lo_b[h] < 0 || hsp_len[h] < 1 || (uint64_t) hsp_lo_a[h] + hsp_len[h] > LA || (uint64_t) hsp_lo_b[h] + hsp_len[h] > LB) {
                rsk_set_error("rsk_mkf_align_pairs: HSP %u of pair %zu outside its chains", h, p);
                return RSK_E_INVALID;
            }
        need += (size_t) LA + LB + 1;
    }
    if (paths_bytes < need) { rsk_set_error("rsk_mkf_align_pairs: paths buffer too small (%zu < %zu)", paths_bytes, need); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = xd_upload_tables(ctx);
    if (rc != RSK_OK) return rc;
    const bool trace = getenv("RSK_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
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
    // sub-batches bounded by the trace scratch: per pair (LA + 18)(LB + 18) bytes cover any split into the two extensions
    const uint64_t TB_BUDGET = 12ull << 30;            // below the block size from which a fresh hipMalloc costs ~30 ms per GB (cold calls)
    size_t done = 0, poff = 0;
    while (done < npairs) {
        const auto t0 = now();
        std::vector<uint64_t> row_off, tb_off, path_off, mp_off;
        uint64_t ro = 0, to = 0, po = 0, mo = 0;
        size_t p1 = done;
        while (p1 < npairs) {
            const uint64_t LA = dba->len[ia[p1]], LB = dbb->len[ib[p1]];
            const uint64_t t_need = (LA + 18) * (LB + 18) + 32;
            if (p1 > done && to + t_need > TB_BUDGET) break;
            row_off.push_back(ro); ro += 2 * (LB + 18);
            tb_off.push_back(to); to += (t_need + 15) & ~15ull;
            path_off.push_back(po); po += LA + LB + 4;
            mp_off.push_back(mo); mo += LA + LB + 5;
            ++p1;
        }
        const size_t m = p1 - done;
        const uint32_t h_lo = hsp_first[done], h_hi = hsp_first[p1];
        std::vector<uint32_t> first(m + 1);
        for (size_t k = 0; k <= m; ++k) first[k] = hsp_first[done + k] - h_lo;
        uint32_t *d_ia, *d_ib, *d_first, *d_loa, *d_lob, *d_pstart, *d_plen, *d_mloa, *d_mlob, *d_mplen, *d_hcnt = nullptr;
        int32_t *d_hla, *d_hlb, *d_hlen, *d_hsc = nullptr;
        uint8_t *d_valid, *d_tb;
        uint64_t *d_rowoff, *d_tboff, *d_pathoff, *d_mpoff, *d_mpstart;
        float *d_rows, *d_xscore, *d_mscore;
        char *d_paths, *d_mpaths;
        const size_t nhb = std::max<size_t>(h_hi - h_lo, 1);
        if ((rc = dalloc((void **) &d_ia, m * 4)) || (rc = dalloc((void **) &d_ib, m * 4)) || (rc = dalloc((void **) &d_first, (m + 1) * 4)) ||
            (rc = dalloc((void **) &d_hla, nhb * 4)) || (rc = dalloc((void **) &d_hlb, nhb * 4)) || (rc = dalloc((void **) &d_hlen, nhb * 4)) ||
            (rc = dalloc((void **) &d_valid, m)) || (rc = dalloc((void **) &d_loa, m * 4)) || (rc = dalloc((void **) &d_lob, m * 4)) ||
            (rc = dalloc((void **) &d_pstart, 2 * m * 4)) || (rc = dalloc((void **) &d_plen, 2 * m * 4)) || (rc = dalloc((void **) &d_xscore, 2 * m * 4)) ||
            (rc = dalloc((void **) &d_rowoff, m * 8)) || (rc = dalloc((void **) &d_tboff, m * 8)) || (rc = dalloc((void **) &d_pathoff, m * 8)) ||
            (rc = dalloc((void **) &d_mpoff, m * 8)) || (rc = dalloc((void **) &d_rows, ro * 4)) || (rc = dalloc((void **) &d_tb, to)) ||
            (rc = dalloc((void **) &d_paths, po)) || (rc = dalloc((void **) &d_mpaths, mo)) || (rc = dalloc((void **) &d_mscore, m * 4)) ||
            (rc = dalloc((void **) &d_mloa, m * 4)) || (rc = dalloc((void **) &d_mlob, m * 4)) || (rc = dalloc((void **) &d_mpstart, m * 8)) ||
            (rc = dalloc((void **) &d_mplen, m * 4)))
            return rc;
        if (hsp_score && ((rc = dalloc((void **) &d_hsc, nhb * 4)) || (rc = dalloc((void **) &d_hcnt, m * 4)))) return rc;
        RSK_HIP(hipMemcpyAsync(d_ia, ia + done, m * 4, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_ib, ib + done, m * 4, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_first, first.data(), (m + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        if (h_hi > h_lo) {
            RSK_HIP(hipMemcpyAsync(d_hla, hsp_lo_a + h_lo, (size_t) (h_hi - h_lo) * 4, hipMemcpyHostToDevice, ctx->stream));
            RSK_HIP(hipMemcpyAsync(d_hlb, hsp_lo_b + h_lo, (size_t) (h_hi - h_lo) * 4, hipMemcpyHostToDevice, ctx->stream));
            RSK_HIP(hipMemcpyAsync(d_hlen, hsp_len + h_lo, (size_t) (h_hi - h_lo) * 4, hipMemcpyHostToDevice, ctx->stream));
            if (hsp_score) RSK_HIP(hipMemcpyAsync(d_hsc, hsp_score + h_lo, (size_t) (h_hi - h_lo) * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        RSK_HIP(hipMemcpyAsync(d_rowoff, row_off.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_tboff, tb_off.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_pathoff, path_off.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemcpyAsync(d_mpoff, mp_off.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
        RSK_HIP(hipMemsetAsync(d_tb, 0, to, ctx->stream));          // unwritten trace cells read as 0 (XDPMem zeroes its matrix)
        RSK_HIP(hipMemsetAsync(d_rows, 0, ro * 4, ctx->stream));    // as XDPMem::Alloc leaves its rows
        mkfa_args sa = {};
        sa.a_ra = dba->d_prof_ra; sa.b_cb = dbb->d_prof_cb; sa.a_off = dba->d_off; sa.b_off = dbb->d_off; sa.a_len = dba->d_len; sa.b_len = dbb->d_len;
        sa.ia = d_ia; sa.ib = d_ib; sa.hsp_first = d_first; sa.hsp_lo_a = d_hla; sa.hsp_lo_b = d_hlb; sa.hsp_len = d_hlen;
        sa.npairs = (uint32_t) m; sa.min_mega = min_mega_score; sa.valid = d_valid; sa.lo_a = d_loa; sa.lo_b = d_lob;
        if (hsp_score) {
            hipLaunchKernelGGL(k_mkf_chain, dim3((unsigned) ((m + 63) / 64)), dim3(64), 0, ctx->stream, d_first, d_hla, d_hlb, d_hlen, d_hsc, (uint32_t) m,
                               d_hcnt, d_valid);
            sa.hsp_cnt = d_hcnt;
        }
        hipLaunchKernelGGL(k_mkf_start, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, ctx->stream, sa);
        xd_args xa = {};
        xa.a_ra = dba->d_prof_ra; xa.b_cb = dbb->d_prof_cb; xa.a_off = dba->d_off; xa.b_off = dbb->d_off; xa.a_len = dba->d_len; xa.b_len = dbb->d_len;
        xa.ia = d_ia; xa.ib = d_ib; xa.lo_a = d_loa; xa.lo_b = d_lob; xa.nreq = (uint32_t) m;
        xa.X = x2; xa.open = gap_open; xa.ext = gap_ext;
        xa.rows = d_rows; xa.row_off = d_rowoff; xa.tb = d_tb; xa.tb_off = d_tboff;
        xa.score = d_xscore; xa.paths = d_paths; xa.path_off = d_pathoff; xa.path_start = d_pstart; xa.path_len = d_plen;
        xa.per_pair = 1; xa.valid = d_valid; xa.only_dir = -1;
        RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        xd_launch(ctx, xa, m);
        RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        mkfm_args ma = {};
        ma.a_len = dba->d_len; ma.b_len = dbb->d_len; ma.ia = d_ia; ma.ib = d_ib; ma.valid = d_valid; ma.req_lo_a = d_loa; ma.req_lo_b = d_lob;
        ma.xscore = d_xscore; ma.xpaths = d_paths; ma.xpath_off = d_pathoff; ma.xpath_start = d_pstart; ma.xpath_len = d_plen;
        ma.npairs = (uint32_t) m; ma.mpaths = d_mpaths; ma.mpath_off = d_mpoff;
        ma.score = d_mscore; ma.lo_a = d_mloa; ma.lo_b = d_mlob; ma.pstart = d_mpstart; ma.plen = d_mplen;
        hipLaunchKernelGGL(k_mkf_merge, dim3((unsigned) ((m + 3) / 4)), dim3(256), 0, ctx->stream, ma);
        RSK_HIP(hipGetLastError());
        RSK_HIP(hipMemcpyAsync(status + done, d_valid, m, hipMemcpyDeviceToHost, ctx->stream));
        const auto t1 = now();
        if (trace) RSK_HIP(hipStreamSynchronize(ctx->stream));
        const auto t2 = now();
        if ((rc = rsk_paths_stats_pack(ctx, dba, dbb, m, ia + done, ib + done, d_ia, d_ib, d_mpaths, d_mpstart, d_mplen, d_mloa, d_mlob, d_mscore,
                                       min_fwd_score, out + done, paths + poff, paths_bytes - poff)) != RSK_OK)
            return rc;
        // path offsets of this sub-batch are relative to its part of the caller's buffer
        size_t used = 0;
        for (size_t k = 0; k < m; ++k) { used = std::max<size_t>(used, (size_t) out[done + k].path_off + out[done + k].path_len + 1); out[done + k].path_off += poff; }
        poff += used;
        if (trace)
            fprintf(stderr, "[rsk_mkf_align_pairs] %zu pairs, %u HSPs: trace scratch %.2f GB; setup %.1f ms, start + X-drop + merge %.1f ms, statistics + d2h %.1f ms\n",
                    m, h_hi - h_lo, to / 1073741824.0, ms(t0, t1), ms(t1, t2), ms(t2, now()));
        for (void *q : ws.all) rsk_pool_free(ctx, q);                // the next sub-batch reuses the blocks
        ws.all.clear();
        done = p1;
    }
    return RSK_OK;
}

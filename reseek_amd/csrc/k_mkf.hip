// k_mkf.hip -- seeding stage of the long-chain path on gfx950 (first half):
//   MuKmerFilter::SetHashTable mukmerfilter.cpp:208-225  query 3-mer table, first HASHW = 4 positions per k-mer
//   MuKmerFilter::Align        mukmerfilter.cpp:316-389  for each target 3-mer, each stored query position:
//                              MuXDrop :105-175 (ungapped X-drop on IntScoreMx_Mu, both directions);
//                              HSPs with score >= MinHSPScore that STRICTLY improve on the best so far and
//                              start at a new query position are kept, in (PosT, slot) order.
// Every pair of the MKF path runs this stage; only the few pairs with a kept HSP go on to chaining and the
// gapped float X-drop (host/dssaligner.cpp).  MI355X layout: one compact hash table per query chain
// that occurs in the batch (same content as the reference's 46656 x 4 table, 32-64 bytes per residue),
// one wave per pair: the lanes take 64 consecutive target positions, look up their four candidate
// query positions with one 16-byte probe (load factor <= 0.5) and extend them; the order-dependent keep rule is resolved by walking the (rare) lanes that beat
// the running best in lane order.  Integer work, L2-resident letters; bound: table-probe latency.
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include "rsk_dev_tables.h"

#define MKF_HASHW 4
#define MKF_WAVES 4
#define MKF_QUEUE 320               // seeds waiting for a full wave: < 64 + 64 target positions x 4 slots

// Table of one query chain: open addressing, 2^bits >= 2 * (L - 2) slots of 16 bytes
//   { k-mer, pos0 | pos1 << 16, pos2 | pos3 << 16, unused },   empty slot: k-mer = MKF_EMPTY, unused positions 0xFFFF.
// The content equals the reference's 46656 x 4 table (first four positions of each k-mer, in chain order) but takes 32-64
// bytes per residue instead of 373 KB per chain, so the tables of a whole database batch stay in L2.
#define MKF_EMPTY 0xFFFFFFFFu
#define MKF_MIN_BITS 6
#define MKF_LDS_BITS 11            // tables up to 2^11 slots (32 KB) are assembled in LDS and copied out

__device__ __forceinline__ uint32_t mkf_hash(uint32_t k, uint32_t bits) { return (k * 2654435761u) >> (32 - bits); }

template <typename TAB>
__device__ __forceinline__ void mkf_insert_all(TAB *T, uint32_t bits, const uint8_t *Q, uint32_t L)
{
    const uint32_t mask = (1u << bits) - 1;
    for (uint32_t p = 0; p + 3 <= L; ++p) {
        const uint32_t k = ((uint32_t) Q[p] * 36 + Q[p + 1]) * 36 + Q[p + 2];
        uint32_t h = mkf_hash(k, bits);
        for (;;) {
            uint4 e = T[h];
            if (e.x == MKF_EMPTY) { e.x = k; e.y = 0xFFFF0000u | p; T[h] = e; break; }
            if (e.x == k) {
                if ((e.y >> 16) == 0xFFFFu) e.y = (e.y & 0xFFFFu) | (p << 16);
                else if ((e.z & 0xFFFFu) == 0xFFFFu) e.z = 0xFFFF0000u | p;
                else if ((e.z >> 16) == 0xFFFFu) e.z = (e.z & 0xFFFFu) | (p << 16);
                else break;                                       // already four positions: first come, first kept
                T[h] = e;
                break;
            }
            h = (h + 1) & mask;
        }
    }
}

// one workgroup per listed query: clear its table, then insert positions in order (first come, <= 4 per k-mer)
__global__ __launch_bounds__(256) void k_mkf_build(const uint8_t *q_mu, const uint32_t *q_off, const uint32_t *q_len, const uint32_t *qlist,
                                                   const uint32_t *tab_off, const uint8_t *tab_bits, uint4 *tables)
{
    const uint32_t q = qlist[blockIdx.x];
    const uint32_t bits = tab_bits[blockIdx.x];
    uint4 *T = tables + tab_off[blockIdx.x];
    const uint32_t H = 1u << bits;
    const uint4 ff = make_uint4(MKF_EMPTY, 0xFFFFFFFFu, 0xFFFFFFFFu, 0);
    const uint8_t *Q = q_mu + q_off[q];
    const uint32_t L = q_len[q];
    __shared__ uint4 S[1u << MKF_LDS_BITS];
    if (bits <= MKF_LDS_BITS) {
        for (uint32_t i = threadIdx.x; i < H; i += blockDim.x) S[i] = ff;
        __syncthreads();
        if (threadIdx.x == 0) mkf_insert_all(S, bits, Q, L);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < H; i += blockDim.x) T[i] = S[i];
    } else {
        for (uint32_t i = threadIdx.x; i < H; i += blockDim.x) T[i] = ff;
        __syncthreads();
        if (threadIdx.x == 0) mkf_insert_all(T, bits, Q, L);
    }
}

template <typename MAT>
__device__ __forceinline__ int mkf_xdrop(const MAT c_mu_int, const uint8_t *Q, int LQ, const uint8_t *T, int LT, int PosQ, int PosT, int X, int &Loi, int &Loj,
                                         int &Len)
{
    Loi = PosQ; Loj = PosT;
    int i = PosQ, j = PosT, fs = 0, bf = 0, flen = 0;
    while (i < LQ && j < LT) {
        fs += c_mu_int[36 * Q[i++] + T[j++]];
        if (fs > bf) { flen = i - PosQ; bf = fs; }
        else if (fs + X < bf) break;
    }
    int rs = 0, br = 0, rlen = 0;
    i = PosQ - 1; j = PosT - 1;
    while (i >= 0 && j >= 0) {
        rs += c_mu_int[36 * Q[i] + T[j]];
        if (rs > br) { br = rs; Loi = i; Loj = j; rlen = PosQ - i; }
        else if (rs + X < br) break;
        --i; --j;
    }
    Len = flen + rlen;
    return bf + br;
}

struct mkf_args {
    const uint8_t *q_mu; const uint32_t *q_off; const uint32_t *q_len;
    const uint8_t *t_mu; const uint32_t *t_off; const uint32_t *t_len;
    const uint32_t *iq, *it;           // pairs
    const uint32_t *qslot;             // table index of each pair's query
    const uint32_t *tab_off;           // per table: first slot in `tables`
    const uint8_t *tab_bits;           // per table: log2 of its slot count
    const uint4 *tables;
    uint32_t pair_lo, pair_hi;         // this launch handles the pairs [pair_lo, pair_hi) of the call
    int X, min_score;
    uint32_t cap;                      // kept HSPs stored per record (<= the kernel instance's CAPMAX)
    uint8_t *found;                    // per pair: any seed with score >= min_score
    // compact records of the found pairs (appended with one atomic each)
    uint32_t *nrec; uint32_t max_rec;
    uint32_t *rec_pair, *rec_nkept;    // nkept may exceed cap: the list is then truncated and the host redoes the pair
    int4 *rec_kept;                    // kept HSPs (Loi, Loj, Len, Score) of the records, back to back: record r owns
                                       // min(nkept, cap) entries from rec_first[r] (only the entries written cross PCIe)
    uint32_t *rec_first, *nent;
};
#define MKF_CAP_MAX 32                 // the search's instance: a pair keeps a handful of HSPs (strictly improving scores)
#define MKF_CAP_BIG 1024               // the instance that redoes a pair whose list did not fit (r04: was a host re-seeding)

template <int CAPMAX>
__global__ __launch_bounds__(64 * MKF_WAVES) void k_mkf_seed(mkf_args a)
{
    __shared__ int4 skept[MKF_WAVES][CAPMAX];
    __shared__ uint32_t sseeds[MKF_WAVES][MKF_QUEUE];
    __shared__ signed char smat[1296];                            // the integer Mu matrix: one LDS read per extension step
    for (int i = threadIdx.x; i < 1296; i += blockDim.x) smat[i] = (signed char) c_mu_int[i];
    __syncthreads();
    // the wave's pair: one value for its 64 lanes (readfirstlane tells the compiler so: lengths, table geometry and loop
    // bounds derived from it stay in SGPRs)
    const uint32_t wv = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t p = a.pair_lo + blockIdx.x * MKF_WAVES + wv;
    if (p >= a.pair_hi) return;
    const int lane = threadIdx.x & 63;
    const uint32_t q = a.iq[p], t = a.it[p];
    const uint8_t *Q = a.q_mu + a.q_off[q], *T = a.t_mu + a.t_off[t];
    const int LQ = (int) a.q_len[q], LT = (int) a.t_len[t];
    const uint32_t slot = a.qslot[p], bits = a.tab_bits[slot], hmask = (1u << bits) - 1;
    const uint4 *tab = a.tables + a.tab_off[slot];
    int4 *kept = skept[wv];
    uint32_t *queue = sseeds[wv];
    int best = 0;
    uint32_t nk = 0, qn = 0;
    bool found = false;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // Two phases so that the extensions run on full waves: the lanes look up the (<= 4) query positions of 64 target
    // positions and append them as seeds (PosQ | PosT << 16) to an LDS queue in (PosT, slot) order -- only a few per cent
    // of the 3-mers occur in the query, so seed by seed the wave would sit in the extension loop with one or two lanes
    // active; whenever 64 seeds wait each lane extends one.  The order-dependent keep rule then walks, in queue order,
    // the lanes that beat the running best.
    auto process = [&](uint32_t n) {                              // the first n <= 64 seeds of the queue
        int sc = 0, loi = 0, loj = 0, len = 0;
        if ((uint32_t) lane < n) {
            const uint32_t sd = queue[lane];
            const int v = mkf_xdrop(smat, Q, LQ, T, LT, (int) (sd & 0xFFFFu), (int) (sd >> 16), a.X, loi, loj, len);
            if (v >= a.min_score) sc = v;
        }
        if (__ballot(sc > 0)) found = true;
        unsigned long long m = __ballot(sc > best);
        while (m) {
            const int l = __builtin_ctzll(m);
            const int v = __shfl(sc, l, 64);
            if (v > best) {                                       // strictly improving (mukmerfilter.cpp:362)
                best = v;
                const int Li = __shfl(loi, l, 64), Lj = __shfl(loj, l, 64), Ln = __shfl(len, l, 64);
                // "Old" test: an HSP with this Loi was kept before (:364-371); the list is short
                bool old = false;
                for (uint32_t k = lane; k < min(nk, a.cap); k += 64) old |= kept[k].x == Li;
                if (!__ballot(old)) {
                    if (lane == 0 && nk < a.cap) kept[nk] = make_int4(Li, Lj, Ln, v);
                    ++nk;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
            }
            m &= m - 1;
            m &= __ballot(sc > best);                            // later seeds must still beat the new best
        }
        // the seeds behind the first n move to the front
        const uint32_t rest = qn - n;
        for (uint32_t k0 = 0; k0 < rest; k0 += 64) {
            uint32_t v = 0;
            if (k0 + lane < rest) v = queue[n + k0 + lane];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (k0 + lane < rest) queue[k0 + lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        qn = rest;
    };
    for (int base = 0; base + 3 <= LT; base += 64) {
        const int PosT = base + lane;
        uint32_t pos[4] = { 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu };
        if (PosT + 3 <= LT) {
            const uint32_t k = ((uint32_t) T[PosT] * 36 + T[PosT + 1]) * 36 + T[PosT + 2];
            uint32_t h = mkf_hash(k, bits);
            uint4 s = tab[h];
            while (s.x != k && s.x != MKF_EMPTY) { h = (h + 1) & hmask; s = tab[h]; }
            if (s.x == k) { pos[0] = s.y & 0xFFFFu; pos[1] = s.y >> 16; pos[2] = s.z & 0xFFFFu; pos[3] = s.z >> 16; }
        }
        // positions fill the slots from the front (first come), so the count says which are valid
        const uint32_t c = (pos[0] != 0xFFFFu) + (pos[1] != 0xFFFFu) + (pos[2] != 0xFFFFu) + (pos[3] != 0xFFFFu);
        const unsigned long long b0 = __ballot(c & 1u), b1 = __ballot(c & 2u), b2 = __ballot(c & 4u);
        if (b0 | b1 | b2) {
            const uint32_t offs = (uint32_t) __popcll(b0 & lt) + 2u * (uint32_t) __popcll(b1 & lt) + 4u * (uint32_t) __popcll(b2 & lt);
#pragma unroll
            for (int w = 0; w < MKF_HASHW; ++w)
                if ((uint32_t) w < c) queue[qn + offs + w] = pos[w] | ((uint32_t) PosT << 16);
            qn += (uint32_t) __popcll(b0) + 2u * (uint32_t) __popcll(b1) + 4u * (uint32_t) __popcll(b2);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            while (qn >= 64) process(64);
        }
    }
    if (qn) process(qn);
    if (lane == 0) a.found[p] = found ? 1 : 0;
    if (found) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(a.nrec, 1u);
        r = (uint32_t) __shfl((int) r, 0, 64);
        if (r < a.max_rec) {
            const uint32_t ne = min(nk, a.cap);
            uint32_t base = 0;
            if (lane == 0) { base = atomicAdd(a.nent, ne); a.rec_pair[r] = p; a.rec_nkept[r] = nk; a.rec_first[r] = base; }
            base = (uint32_t) __shfl((int) base, 0, 64);
            for (uint32_t k = lane; k < ne; k += 64) a.rec_kept[(size_t) base + k] = kept[k];
        }
    }
}

extern "C" int rsk_mkf_seed_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it, size_t npairs,
                                  int x1, int min_hsp_score, uint32_t cap, uint8_t *found, size_t max_records, size_t *nrecords,
                                  uint32_t *rec_pair, uint32_t *rec_nkept, int32_t *rec_kept)
{
    if (!ctx || !q || !t || (npairs && (!iq || !it || !found)) || !nrecords || cap == 0 || cap > MKF_CAP_BIG ||
        (max_records && (!rec_pair || !rec_nkept || !rec_kept))) {
        rsk_set_error("rsk_mkf_seed_pairs: bad argument (cap must be 1..%d)", MKF_CAP_BIG);
        return RSK_E_INVALID;
    }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mkf_seed_pairs: chain set has no Mu letters"); return RSK_E_INVALID; }
    *nrecords = 0;
    if (npairs == 0) return RSK_OK;
    if (npairs > 0x7FFFFFFFull || max_records > 0x7FFFFFFFull) { rsk_set_error("rsk_mkf_seed_pairs: too many pairs in one call"); return RSK_E_RANGE; }
    for (size_t p = 0; p < npairs; ++p)
        if (iq[p] >= q->n || it[p] >= t->n) { rsk_set_error("rsk_mkf_seed_pairs: pair %zu out of range", p); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    int rc = rsk_upload_mu_tables(ctx);
    if (rc != RSK_OK) return rc;
    // distinct queries -> tables.  A table takes 16 bytes x 2^bits slots (2^bits >= 2 (L - 2)); the pairs are cut into
    // consecutive chunks whose tables fit a byte budget (and a table-count budget, RSK_MKF_MAX_TABLES) and reuse one block,
    // so a call whose query side is a whole database (-db mode with a long query: every DB chain is a distinct iq) is
    // bounded whatever the database size.
    const size_t budget = getenv("RSK_MKF_MAX_TABLES") ? (size_t) std::max(1, atoi(getenv("RSK_MKF_MAX_TABLES"))) : (size_t) 1 << 22;
    const size_t slot_budget = (size_t) 1 << 29;                 // 8 GB of 16-byte slots
    struct chunk_t { size_t p0, p1, q0, q1; };
    std::vector<chunk_t> chunks;
    std::vector<uint32_t> slot_of(q->n, 0xFFFFFFFFu), qlist, qslot(npairs), tab_off;
    std::vector<uint8_t> tab_bits;
    size_t max_slots = 0;
    {
        size_t p0 = 0, q0 = 0, slots = 0;
        for (size_t p = 0; p < npairs; ++p) {
            if (slot_of[iq[p]] == 0xFFFFFFFFu) {
                const uint32_t L = q->len[iq[p]], nk = L >= 3 ? L - 2 : 0;
                uint32_t bits = MKF_MIN_BITS;
                while ((1u << bits) < 2 * nk) ++bits;
                if (qlist.size() - q0 == budget || (slots && slots + ((size_t) 1 << bits) > slot_budget)) {   // close the chunk before this pair
                    chunks.push_back({ p0, p, q0, qlist.size() });
                    for (size_t k = q0; k < qlist.size(); ++k) slot_of[qlist[k]] = 0xFFFFFFFFu;
                    p0 = p; q0 = qlist.size(); slots = 0;
                }
                slot_of[iq[p]] = (uint32_t) qlist.size();
                qlist.push_back(iq[p]);
                tab_off.push_back((uint32_t) slots);
                tab_bits.push_back((uint8_t) bits);
                slots += (size_t) 1 << bits;
                max_slots = std::max(max_slots, slots);
            }
            qslot[p] = slot_of[iq[p]];
        }
        chunks.push_back({ p0, npairs, q0, qlist.size() });
    }
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
    uint32_t *d_iq, *d_it, *d_qslot, *d_qlist, *d_toff, *d_nrec, *d_rpair, *d_rnk;
    uint4 *d_tab;
    uint8_t *d_found, *d_tbits;
    int4 *d_rkept;
    uint32_t *d_rfirst, *d_nent;
    if ((rc = dalloc((void **) &d_iq, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_it, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_qslot, npairs * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_qlist, qlist.size() * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_toff, qlist.size() * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_tbits, qlist.size())) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_tab, max_slots * sizeof(uint4))) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_found, npairs)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_nrec, 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_rpair, max_records * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_rnk, max_records * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_rkept, max_records * (size_t) cap * sizeof(int4))) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_rfirst, max_records * 4)) != RSK_OK) return rc;
    if ((rc = dalloc((void **) &d_nent, 4)) != RSK_OK) return rc;
    RSK_HIP(hipMemcpyAsync(d_iq, iq, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_it, it, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_qslot, qslot.data(), npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_qlist, qlist.data(), qlist.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_toff, tab_off.data(), qlist.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_tbits, tab_bits.data(), qlist.size(), hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_nrec, 0, 4, ctx->stream));
    RSK_HIP(hipMemsetAsync(d_nent, 0, 4, ctx->stream));
    RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    mkf_args a = {};
    a.q_mu = q->d_mu; a.q_off = q->d_off; a.q_len = q->d_len;
    a.t_mu = t->d_mu; a.t_off = t->d_off; a.t_len = t->d_len;
    a.iq = d_iq; a.it = d_it; a.qslot = d_qslot; a.tab_off = d_toff; a.tab_bits = d_tbits; a.tables = d_tab;
    a.X = x1; a.min_score = min_hsp_score; a.cap = cap;
    a.found = d_found; a.nrec = d_nrec; a.max_rec = (uint32_t) max_records;
    a.rec_pair = d_rpair; a.rec_nkept = d_rnk; a.rec_kept = d_rkept; a.rec_first = d_rfirst; a.nent = d_nent;
    for (const chunk_t &c : chunks) {                       // same stream: a chunk's tables are rebuilt after its seeding kernel is done
        hipLaunchKernelGGL(k_mkf_build, dim3((unsigned) (c.q1 - c.q0)), dim3(256), 0, ctx->stream, q->d_mu, q->d_off, q->d_len, d_qlist + c.q0,
                           d_toff + c.q0, d_tbits + c.q0, d_tab);
        a.pair_lo = (uint32_t) c.p0; a.pair_hi = (uint32_t) c.p1;
        const dim3 grid((unsigned) ((c.p1 - c.p0 + MKF_WAVES - 1) / MKF_WAVES));
        if (cap <= MKF_CAP_MAX) hipLaunchKernelGGL(k_mkf_seed<MKF_CAP_MAX>, grid, dim3(64 * MKF_WAVES), 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_mkf_seed<MKF_CAP_BIG>, grid, dim3(64 * MKF_WAVES), 0, ctx->stream, a);
    }
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    uint32_t nrec = 0;
    RSK_HIP(hipMemcpyAsync(found, d_found, npairs, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(&nrec, d_nrec, 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    *nrecords = nrec;
    if (getenv("RSK_TRACE")) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        fprintf(stderr, "[rsk_mkf_seed_pairs] %zu pairs, %zu tables in %zu chunk(s), %.1f MB of tables, kernels %.2f ms\n", npairs, qlist.size(),
                chunks.size(), max_slots * 16 / 1048576.0, ms);
    }
    const size_t m = std::min<size_t>(nrec, max_records);
    if (m) {
        // the records' HSP lists sit back to back on the device (a few entries per record, not cap): they are copied
        // compact and spread into the caller's [record][cap] layout here
        uint32_t nent = 0;
        RSK_HIP(hipMemcpyAsync(&nent, d_nent, 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipMemcpyAsync(rec_pair, d_rpair, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipMemcpyAsync(rec_nkept, d_rnk, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        std::unique_ptr<uint32_t[]> first(new uint32_t[m]);
        RSK_HIP(hipMemcpyAsync(first.get(), d_rfirst, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipStreamSynchronize(ctx->stream));
        std::unique_ptr<int4[]> ent(new int4[(size_t) nent + 1]);
        if (nent) RSK_HIP(hipMemcpy(ent.get(), d_rkept, (size_t) nent * sizeof(int4), hipMemcpyDeviceToHost));
        rsk_parallel_for(m, 4096, [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; ++r) {
                const uint32_t ne = std::min(rec_nkept[r], cap);
                memcpy(rec_kept + r * (size_t) cap * 4, ent.get() + first[r], (size_t) ne * sizeof(int4));
            }
        });
    }
    return RSK_OK;
}

// runpairs.cpp -- the pair space of a search in GPU batches (runself.cpp:13-99, runquery.cpp:18-125: where the reference
// hands one pair at a time to one DSSAligner per thread):
//   Mu filter (rsk_mu_filter_dev) -> survivors -> alignment batches (rsk_align_pairs: AlignBatch) -> hit records replayed
//   through DSSAligner + BaseOnAln (ReplayBatch), several batches in flight on helper contexts;
//   long-chain pairs (DoMKF dssaligner.cpp:715) through RunMKFPairs: seeding, chaining, gapped X-drop, merge, statistics in
//   device batches, started before the filter on a context of its own.
#include "host_internal.h"

namespace reseek_amd {
// Align a batch of (ia, ib) pairs of one chain set on the GPU and replay the hits.
// One batch of (ia, ib) pairs: the GPU stage (AlignBatch: rsk_align_pairs) and the host stage (ReplayBatch: hit
// records -> Reject -> TSV lines).  RunPairs runs the GPU stage of batch k + 1 while batch k is replayed.
// Page-locked host buffers for the packed paths of a batch (hundreds of MB; the device-to-host copy into pageable
// memory was ~25 % of the GPU stage): two or three buffers are recycled between the batches of a run.
struct AlignedBatch {
    std::vector<uint32_t> ia, ib;
    std::vector<rsk_aln> out;
    PinnedPool *pool = nullptr;
    char *paths = nullptr;
    size_t paths_cap = 0;
    ~AlignedBatch() { if (paths) pool->Put(paths, paths_cap); }
};

static std::unique_ptr<AlignedBatch> AlignBatch(const DSSParams &P, rsk_ctx *ctx, PinnedPool &Pool, DBSearcher &SrcA, DBSearcher &SrcB,
                                                std::vector<uint32_t> ia, std::vector<uint32_t> ib)
{
    std::unique_ptr<AlignedBatch> B(new AlignedBatch);
    B->ia = std::move(ia);
    B->ib = std::move(ib);
    const size_t n = B->ia.size();
    if (n == 0) return B;
    B->out.resize(n);
    const size_t bytes = rsk_align_paths_bytes(SrcA.m_Db, SrcB.m_Db, B->ia.data(), B->ib.data(), n);
    B->pool = &Pool;
    B->paths = Pool.Get(bytes + 1, B->paths_cap);
    check(rsk_align_pairs(ctx, SrcA.m_Db, SrcB.m_Db, B->ia.data(), B->ib.data(), n, P.m_GapOpen, P.m_GapExt, P.m_MinFwdScore, B->out.data(),
                          B->paths, bytes),
          "rsk_align_pairs");
    return B;
}

static void ReplayBatch(DBSearcher &S, DBSearcher &SrcA, DBSearcher &SrcB, const std::vector<uint32_t> &ia, const std::vector<uint32_t> &ib,
                        const std::vector<rsk_aln> &out, const char *paths, bool Self, uint joff = 0)
{
    const DSSParams &P = *S.m_Params;
    const size_t n = ia.size();
    if (n == 0) return;
    S.m_SWCount += n;
    {
        // rsk_path_counters: the pairs whose score reached m_MinFwdScore (CalcEvalue ran and set an E-value, dssaligner.cpp:852-861)
        uint64_t scored = 0;
        for (size_t p = 0; p < n; ++p) scored += out[p].evalue != FLT_MAX;
        g_rsk_counters.sw_pairs += n;
        g_rsk_counters.sw_pairs_scored += scored;
    }
    // one pair's hit record -> DSSAligner result fields -> Reject / hit line(s), as runself.cpp:61-66 / runquery.cpp:72-73
    auto replay = [&](DSSAligner &DA, size_t p, auto &&OnHit) {
        if (out[p].path_len == 0) return;                                    // runself.cpp:61 / runquery.cpp:72
        // Reject (dbsearcher.cpp:258) on the batch record itself: both orientations carry the same E-value / TS, and a
        // plain DBSearcher does nothing with a rejected hit -- skip the string work for the (many) rejected pairs
        if (!S.m_HasOnAlnOverride) {
            const float ev = out[p].evalue, ts = out[p].evalue != FLT_MAX ? out[p].ts : -FLT_MAX;
            if (!S.m_Opts.scores_are_not_evalues && ev > S.m_MaxEvalue) return;
            if (S.m_Opts.mints_set && ts < S.m_Opts.mints) return;
        }
        const uint i = ia[p], j = ib[p];
        DA.m_ChainA = SrcA.m_DBChains[i]; DA.m_ProfileA = SrcA.m_DBProfiles[i];
        DA.m_ChainB = SrcB.m_DBChains[j]; DA.m_ProfileB = SrcB.m_DBProfiles[j];
        DA.m_SelfRevScoreA = SrcA.m_DBSelfRevScores[i]; DA.m_SelfRevScoreB = SrcB.m_DBSelfRevScores[j];
        DA.SetFromAln(out[p], paths + out[p].path_off);
        if (Self) {
            OnHit(DA, true);
            if (i != joff + j) OnHit(DA, false);
        } else
            OnHit(DA, false);                                                // runquery.cpp:73: A = DB chain, B = query
    };
    const unsigned T = (unsigned) std::min<size_t>(HostThreads(64), n / 2048 + 1);
    if (S.m_HasOnAlnOverride || T < 2) {
        // subclasses see every hit through OnAln in pair order, one at a time (the reference's m_Lock semantics)
        for (size_t p = 0; p < n; ++p) replay(S.m_DA, p, [&](DSSAligner &DA, bool Up) { S.BaseOnAln(DA, Up); });
        return;
    }
    // plain DBSearcher: BaseOnAln = Reject + hit count + one TSV line.  Threads format contiguous slices of the batch
    // into strings, which are then appended to the output in slice order (= the sequential row order).
    struct slice { std::string buf; uint64_t hits = 0; std::string err; };
    std::vector<slice> sl(T);
    PhaseTimer rt("ReplayBatch");
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < T; ++t)
        ts.emplace_back([&, t]() {
            slice &me = sl[t];
            try {
                const bool want = S.m_fTsv != nullptr;
                DSSAligner DA;
                DA.SetParams(P);
                DA.m_UFs = S.m_DA.m_UFs;
                const size_t lo = n * t / T, hi = n * (t + 1) / T;
                if (want) me.buf.reserve((hi - lo) * 56);
                // A hit line touches two chain objects, their labels and sequences (percent identity) and the path: five
                // or six cache misses that cost more than the formatting itself.  They are requested a few pairs ahead.
                auto touch = [&](size_t q, bool deep) {
                    const PDBChain *a = SrcA.m_DBChains[ia[q]], *b = SrcB.m_DBChains[ib[q]];
                    if (!deep) { __builtin_prefetch(a); __builtin_prefetch(b); return; }
                    if (out[q].path_len == 0) return;
                    __builtin_prefetch(a->m_Label.data()); __builtin_prefetch(b->m_Label.data());
                    if (out[q].lo_a != RSK_NO_POS) { __builtin_prefetch(a->m_Seq.data() + out[q].lo_a); __builtin_prefetch(a->m_Seq.data() + out[q].lo_a + 64); }
                    if (out[q].lo_b != RSK_NO_POS) { __builtin_prefetch(b->m_Seq.data() + out[q].lo_b); __builtin_prefetch(b->m_Seq.data() + out[q].lo_b + 64); }
                    __builtin_prefetch(paths + out[q].path_off);
                };
                for (size_t p = lo; p < hi; ++p) {
                    if (p + 24 < hi) touch(p + 24, false);
                    if (p + 12 < hi) touch(p + 12, true);
                    replay(DA, p, [&](DSSAligner &D, bool Up) {
                        if (S.Reject(D, Up)) return;
                        ++me.hits;
                        if (want && !(S.m_Opts.noself && D.m_ChainA->m_Label == D.m_ChainB->m_Label)) D.AppendTsv(me.buf, Up);
                    });
                }
                DA.UnsetQuery();
            } catch (const std::exception &e) { me.err = e.what(); }
        });
    for (auto &t : ts) t.join();
    rt.lap("format (threads)");
    for (slice &me : sl) {
        if (me.err.empty() && !me.buf.empty() && fwrite(me.buf.data(), 1, me.buf.size(), S.m_fTsv) != me.buf.size())
            me.err = "short write to the hits file";
        S.m_HitCount += me.hits;
    }
    rt.lap("append to the hits file");
    for (slice &me : sl)
        if (!me.err.empty()) throw std::runtime_error("hit replay: " + me.err);
}

// Shared body of RunSelf / RunQuery: A-side chains come from SrcA, B-side from *this.
std::vector<std::pair<size_t, size_t> > AlignBatches(const SearchOptions &O, const DBSearcher &A, const DBSearcher &B,
                                                     const std::vector<uint32_t> &ia, const std::vector<uint32_t> &ib)
{
    std::vector<std::pair<size_t, size_t> > out;
    // RSK_BATCH_PAIRS lowers the batch size so that tests reach the multi-batch pipeline with small inputs
    const size_t maxp = std::max<size_t>(1, getenv("RSK_BATCH_PAIRS") ? (size_t) atoll(getenv("RSK_BATCH_PAIRS")) : O.batch_pairs);
    const uint64_t maxc = getenv("RSK_BATCH_CELLS") ? std::max<uint64_t>(1, (uint64_t) atoll(getenv("RSK_BATCH_CELLS"))) : O.batch_cells;
    size_t b = 0;
    uint64_t cells = 0;
    // flat length tables: the loop below runs over tens of millions of pairs (two pointer chases per pair took 0.1 s)
    std::vector<uint32_t> la(A.m_DBChains.size()), lb(B.m_DBChains.size());
    for (size_t i = 0; i < la.size(); ++i) la[i] = A.m_DBChains[i]->GetSeqLength();
    for (size_t j = 0; j < lb.size(); ++j) lb[j] = B.m_DBChains[j]->GetSeqLength();
    // The greedy cut below is sequential (a batch closes when the next pair would take it over either limit), and a streamed
    // -verysensitive pass hands it 33 M pairs: the cells of chunks of 65,536 pairs are summed on the host threads first, and the
    // sequential walk only enters the chunks a cut falls into (a chunk that fits whole is added whole: no pair of it can trigger
    // either test) -- the same cuts, a fifth of the serial work.
    const size_t n = ia.size(), CH = 65536, nch = (n + CH - 1) / CH;
    std::vector<uint64_t> chsum(nch, 0);
    rsk_parallel_for(nch, 4, [&](size_t lo, size_t hi) {
        for (size_t ch = lo; ch < hi; ++ch) {
            uint64_t sum = 0;
            for (size_t k = ch * CH, e = std::min(n, k + CH); k < e; ++k) sum += (uint64_t) la[ia[k]] * lb[ib[k]];
            chsum[ch] = sum;
        }
    });
    for (size_t ch = 0; ch < nch; ++ch) {
        const size_t k0 = ch * CH, k1 = std::min(n, k0 + CH);
        if (k1 - 1 - b < maxp && cells + chsum[ch] <= maxc && (k0 > b || k0 == 0)) { cells += chsum[ch]; continue; }
        for (size_t k = k0; k < k1; ++k) {
            const uint64_t c = (uint64_t) la[ia[k]] * lb[ib[k]];
            if (k > b && (k - b >= maxp || cells + c > maxc)) { out.emplace_back(b, k); b = k; cells = 0; }
            cells += c;
        }
    }
    if (b < ia.size()) out.emplace_back(b, ia.size());
    return out;
}

// rsk_align_pairs over the batches of a pair list with the GPU stage of batch k + 1 running while `OnBatch` consumes
// batch k on the calling thread (RunPairs: hit replay; PostMuFilter: Accept + hit lines).
typedef std::function<void(const std::vector<uint32_t> &, const std::vector<uint32_t> &, const std::vector<rsk_aln> &, const char *)> OnBatchFn;
typedef std::function<void(size_t, size_t, std::vector<uint32_t> &, std::vector<uint32_t> &)> FillPairsFn;      // pairs [b, e) of the list -> ia, ib
static void ForEachBatchOf(const DSSParams &P, rsk_ctx *ctx, DBSearcher &SrcA, DBSearcher &SrcB, const std::vector<std::pair<size_t, size_t> > &batches,
                           const FillPairsFn &fill, const OnBatchFn &OnBatch);

void ForEachAlignedBatch(const DSSParams &P, rsk_ctx *ctx, const SearchOptions &O, DBSearcher &SrcA, DBSearcher &SrcB,
                         const std::vector<uint32_t> &ia, const std::vector<uint32_t> &ib, const OnBatchFn &OnBatch)
{
    ForEachBatchOf(P, ctx, SrcA, SrcB, AlignBatches(O, SrcA, SrcB, ia, ib),
                   [&](size_t b, size_t e, std::vector<uint32_t> &ba, std::vector<uint32_t> &bb) {
                       ba.assign(ia.begin() + (ptrdiff_t) b, ia.begin() + (ptrdiff_t) e);
                       bb.assign(ib.begin() + (ptrdiff_t) b, ib.begin() + (ptrdiff_t) e);
                   },
                   OnBatch);
}

// Every pair of a rectangle SrcA x SrcB in row-major order -- what a -verysensitive search (no Mu filter) aligns of a streamed
// database batch: 33 M pairs per 32,768 chains against 1,000 queries -- WITHOUT the pair list: the cells of the first k pairs are
// F(k) = (sum of B's lengths) x (lengths of A's rows before i) + la[i] x (B's lengths before j), i = k / NB, j = k % NB, so the
// batch cuts of AlignBatches (close a batch when the next pair would take it over the pair or the cell limit) are binary
// searches over F, and a stage writes its own slice of (i, j) when it starts.  (r01-r05 filled two 131-MB index arrays per
// database batch, walked them for the cuts and released them: ~90 ms of every 1.6-s pass with the GPU idle.)
void ForEachAlignedBatchDense(const DSSParams &P, rsk_ctx *ctx, const SearchOptions &O, DBSearcher &SrcA, DBSearcher &SrcB, const OnBatchFn &OnBatch)
{
    const size_t NA = SrcA.m_DBChains.size(), NB = SrcB.m_DBChains.size(), n = NA * NB;
    const size_t maxp = std::max<size_t>(1, getenv("RSK_BATCH_PAIRS") ? (size_t) atoll(getenv("RSK_BATCH_PAIRS")) : O.batch_pairs);
    const uint64_t maxc = getenv("RSK_BATCH_CELLS") ? std::max<uint64_t>(1, (uint64_t) atoll(getenv("RSK_BATCH_CELLS"))) : O.batch_cells;
    std::vector<uint64_t> pla(NA + 1, 0), clb(NB + 1, 0);
    std::vector<uint32_t> la(NA + 1, 0);
    for (size_t i = 0; i < NA; ++i) { la[i] = SrcA.m_DBChains[i]->GetSeqLength(); pla[i + 1] = pla[i] + la[i]; }
    for (size_t j = 0; j < NB; ++j) clb[j + 1] = clb[j] + SrcB.m_DBChains[j]->GetSeqLength();
    const uint64_t SB = clb[NB];
    auto F = [&](size_t k) -> uint64_t { const size_t i = NB ? k / NB : 0, j = NB ? k % NB : 0; return SB * pla[i] + (uint64_t) la[i] * clb[j]; };
    std::vector<std::pair<size_t, size_t> > batches;
    for (size_t b = 0; b < n;) {
        size_t lo = b + 1, hi = std::min(n, b + maxp);                 // the batch ends at the largest e in [lo, hi] with F(e) - F(b) <= maxc (at least one pair)
        const uint64_t Fb = F(b);
        while (lo < hi) {
            const size_t mid = lo + (hi - lo + 1) / 2;
            if (F(mid) - Fb <= maxc) lo = mid; else hi = mid - 1;
        }
        batches.emplace_back(b, lo);
        b = lo;
    }
    ForEachBatchOf(P, ctx, SrcA, SrcB, batches,
                   [NB](size_t b, size_t e, std::vector<uint32_t> &ba, std::vector<uint32_t> &bb) {
                       ba.resize(e - b); bb.resize(e - b);
                       size_t i = b / NB, j = b % NB;
                       for (size_t k = b; k < e; ++k) {
                           ba[k - b] = (uint32_t) i; bb[k - b] = (uint32_t) j;
                           if (++j == NB) { j = 0; ++i; }
                       }
                   },
                   OnBatch);
}

static void ForEachBatchOf(const DSSParams &P, rsk_ctx *ctx, DBSearcher &SrcA, DBSearcher &SrcB, const std::vector<std::pair<size_t, size_t> > &batches,
                           const FillPairsFn &fill, const OnBatchFn &OnBatch)
{
    PinnedPool &Pool = PinnedPool::Shared();                             // the process's pool: outlives every batch, and the call
    // Several GPU stages in flight while batch k is replayed, each on a context of its own (device pool, staging buffers,
    // stream): the host part of rsk_align_pairs (grouping the pairs, work items, statistics) is a quarter of a stage, with a
    // single stage in flight the GPU idles through it.  The chain sets are read-only here.
    // Up to three stages in flight: with two, both were launched at the same moment and stayed in phase (host part, then
    // kernels, on both contexts at once), so the GPU idled through every host part.
    SecondaryCtx second, third;
    const int inflight = getenv("RSK_ALIGN_INFLIGHT") ? std::max(1, std::min(3, atoi(getenv("RSK_ALIGN_INFLIGHT")))) : 3;
    if (inflight > 1 && batches.size() >= 3) second.Create(ctx->device, "align");
    if (inflight > 2 && batches.size() >= 4) third.Create(ctx->device, "align");
    const size_t nctx = 1 + (second.c ? 1 : 0) + (third.c ? 1 : 0);
    rsk_ctx *const ring[3] = { ctx, second.c ? second.c : ctx, third.c ? third.c : (second.c ? second.c : ctx) };
    auto launch = [&](size_t k) {
        const auto be = batches[k];
        rsk_ctx *c = ring[k % nctx];
        return std::async(std::launch::async, [&, be, c]() {
            std::vector<uint32_t> ba, bb;
            fill(be.first, be.second, ba, bb);
            return AlignBatch(P, c, Pool, SrcA, SrcB, std::move(ba), std::move(bb));
        });
    };
    std::deque<std::future<std::unique_ptr<AlignedBatch> > > q;
    size_t launched = 0;
    auto drain = [&]() { for (auto &f : q) if (f.valid()) f.wait(); };   // stages in flight reference this frame
    // the first stage of a process runs alone (one-time table uploads); later calls (the next batch of a streamed
    // database) start with both stages
    static std::atomic<bool> tables_up{false};
    if (!batches.empty()) q.push_back(launch(launched++));
    while (tables_up.load() && q.size() < nctx && launched < batches.size()) q.push_back(launch(launched++));
    for (size_t k = 0; k < batches.size(); ++k) {
        std::unique_ptr<AlignedBatch> cur;
        try {
            cur = q.front().get();                                       // rethrows a failed GPU stage
        } catch (...) {
            q.pop_front();
            drain();
            throw;
        }
        tables_up.store(true);
        q.pop_front();
        while (launched < batches.size() && q.size() < nctx) q.push_back(launch(launched++));
        try {
            OnBatch(cur->ia, cur->ib, cur->out, cur->paths);
        } catch (...) {
            drain();
            throw;
        }
    }
}

void RunMKFPairs(rsk_ctx *Ctx, const DSSParams &P, const std::string &Columns, DBSearcher &SrcA, DBSearcher &SrcB,
                 const std::vector<std::pair<uint32_t, uint32_t> > &Pairs, const std::function<void(DSSAligner &, uint, uint)> &OnHit,
                 const std::function<void(DSSAligner &, uint, uint, unsigned)> *OnHitOfWorker)
{
    const size_t n = Pairs.size();
    if (n == 0) return;
    // seed HSPs per pair returned by the device (RSK_MKF_CAP lowers it so that tests reach the truncated-list path)
    const uint32_t CAP = getenv("RSK_MKF_CAP") ? (uint32_t) std::max(1, std::min(32, atoi(getenv("RSK_MKF_CAP")))) : 32;
    // records of the pairs that have a seed HSP (everything else has no alignment: mukmerfilter.cpp:387, dssaligner.cpp:1397)
    struct Rec { uint32_t pair, nkept; std::vector<int32_t> kept; };
    std::vector<Rec> recs;
    const size_t BATCH = 1u << 22;
    for (size_t b = 0; b < n; b += BATCH) {
        const size_t m = std::min(n, b + BATCH) - b;
        std::vector<uint32_t> iq(m), it(m);
        for (size_t k = 0; k < m; ++k) { iq[k] = Pairs[b + k].first; it[k] = Pairs[b + k].second; }
        std::vector<uint8_t> found(m);
        // Room for a record of every pair up to 2 M pairs: a self search seeds a few per cent of its long-chain pairs, a -db
        // search with long queries 95 %, and a record list that overflows costs a second run of the whole seeding kernel.
        // The arrays are not value-initialised (512 B per record; only the records returned are ever touched).
        size_t maxrec = m <= ((size_t) 2 << 20) ? m : std::max<size_t>((size_t) 2 << 20, m / 4), nrec = 0;
        std::unique_ptr<uint32_t[]> rp, rn;
        std::unique_ptr<int32_t[]> rk;
        for (;;) {
            rp.reset(new uint32_t[maxrec]); rn.reset(new uint32_t[maxrec]); rk.reset(new int32_t[maxrec * CAP * 4]);
            check(rsk_mkf_seed_pairs(Ctx, SrcA.m_Db, SrcB.m_Db, iq.data(), it.data(), m, P.m_MKF_X1, P.m_MKF_MinHSPScore, CAP, found.data(), maxrec,
                                     &nrec, rp.get(), rn.get(), rk.get()),
                  "rsk_mkf_seed_pairs");
            if (nrec <= maxrec) break;
            maxrec = nrec;
        }
        for (size_t r = 0; r < nrec; ++r) {
            Rec R;
            R.pair = (uint32_t) (b + rp[r]);
            R.nkept = rn[r];
            R.kept.assign(rk.get() + r * CAP * 4, rk.get() + r * CAP * 4 + 4 * (size_t) std::min(rn[r], CAP));
            recs.push_back(std::move(R));
        }
    }
    std::sort(recs.begin(), recs.end(), [](const Rec &x, const Rec &y) { return x.pair < y.pair; });
    if (getenv("RSK_TRACE")) fprintf(stderr, "[RunMKFPairs] %zu pairs, %zu with a seed HSP\n", n, recs.size());
    {
        // a seed list that did not fit the record (more than CAP strictly improving HSPs): the same kernel again with room
        // for 1024 (r01-r03 re-seeded such a pair on the host with a copy of MuKmerFilter::Align)
        std::vector<size_t> redo;
        for (size_t r = 0; r < recs.size(); ++r)
            if (recs[r].nkept > CAP) redo.push_back(r);
        if (!redo.empty()) {
            const uint32_t BIG = 1024;
            const size_t m = redo.size();
            std::vector<uint32_t> iq(m), it(m), rp(m), rn(m);
            std::vector<uint8_t> found(m);
            std::vector<int32_t> rk(m * (size_t) BIG * 4);
            for (size_t k = 0; k < m; ++k) { iq[k] = Pairs[recs[redo[k]].pair].first; it[k] = Pairs[recs[redo[k]].pair].second; }
            size_t nrec = 0;
            check(rsk_mkf_seed_pairs(Ctx, SrcA.m_Db, SrcB.m_Db, iq.data(), it.data(), m, P.m_MKF_X1, P.m_MKF_MinHSPScore, BIG, found.data(), m, &nrec,
                                     rp.data(), rn.data(), rk.data()),
                  "rsk_mkf_seed_pairs");
            if (nrec != m) throw std::runtime_error("RunMKFPairs: the re-seeded pairs lost their seed HSPs");
            for (size_t r = 0; r < nrec; ++r) {
                if (rn[r] > BIG) throw std::runtime_error("RunMKFPairs: a pair keeps more than 1024 seed HSPs");
                Rec &R = recs[redo[rp[r]]];
                R.nkept = rn[r];
                R.kept.assign(rk.data() + r * (size_t) BIG * 4, rk.data() + r * (size_t) BIG * 4 + 4 * (size_t) rn[r]);
            }
            if (getenv("RSK_TRACE")) fprintf(stderr, "[RunMKFPairs] %zu pairs re-seeded with room for %u HSPs\n", m, BIG);
        }
    }
    const auto t_host0 = std::chrono::steady_clock::now();
    const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>((size_t) HostThreads(128), recs.size() / 8 + 1));
    auto parallel = [&](const std::function<void(DSSAligner &, size_t, unsigned)> &fn) {
        std::atomic<size_t> next{0};
        auto body = [&](unsigned worker) {
            DSSAligner DA;
            DA.SetParams(P);
            DA.SetColumns(Columns);
            for (;;) {
                const size_t r = next.fetch_add(1);
                if (r >= recs.size()) break;
                fn(DA, r, worker);
            }
            DA.UnsetQuery();
        };
        if (T == 1) body(0);
        else {
            std::vector<std::thread> ts;
            std::vector<std::string> errs(T);
            for (unsigned t = 0; t < T; ++t)
                ts.emplace_back([&, t]() { try { body(t); } catch (const std::exception &e) { errs[t] = e.what(); } });
            for (auto &t : ts) t.join();
            for (auto &e : errs)
                if (!e.empty()) throw std::runtime_error(e);
        }
    };
    // stage 1 + 2 (GPU, one batch): the seed HSPs of every record -> chain (Chainer::Chain), mega-HSP scores + gates, start of
    // the gapped extensions, both extensions, merge, statistics (rsk_mkf_chain_align_pairs).  One kind of pair makes a second,
    // small batch after a host step: pairs whose chain depends on libc qsort's order of equal end points (status 3) are
    // chained by ChainHSPs (the reference's own outcome there is its qsort's) here.
    struct Chained { std::vector<int32_t> lo_a, lo_b, len; };
    std::vector<size_t> slot(recs.size(), (size_t) -1);
    std::vector<uint32_t> xa, xb, first(1, 0);
    std::vector<int32_t> hla, hlb, hlen, hsc;
    std::vector<size_t> host_recs;                                      // records of the second batch
    size_t xbytes = 0;
    for (size_t r = 0; r < recs.size(); ++r) {
        const Rec &R = recs[r];
        slot[r] = xa.size();
        const uint i = Pairs[R.pair].first, j = Pairs[R.pair].second;
        xa.push_back(i); xb.push_back(j);
        for (uint32_t k = 0; k < R.nkept; ++k) {
            hla.push_back(R.kept[4 * k]); hlb.push_back(R.kept[4 * k + 1]); hlen.push_back(R.kept[4 * k + 2]); hsc.push_back(R.kept[4 * k + 3]);
        }
        first.push_back((uint32_t) hla.size());
        xbytes += (size_t) SrcA.m_DBChains[i]->GetSeqLength() + SrcB.m_DBChains[j]->GetSeqLength() + 1;
    }
    size_t nx = xa.size();
    std::vector<rsk_aln> xout(nx);
    std::vector<uint8_t> xstatus(nx);
    std::unique_ptr<char[]> xpaths_mem(new char[xbytes + 16]);         // hundreds of MB: not value-initialised
    char *xpaths = xpaths_mem.get();
    if (nx)
        check(rsk_mkf_chain_align_pairs(Ctx, SrcA.m_Db, SrcB.m_Db, xa.data(), xb.data(), nx, first.data(), hla.data(), hlb.data(), hlen.data(),
                                        hsc.data(), float(P.m_MKF_X2), P.m_GapOpen, P.m_GapExt, P.m_MKF_MinMegaHSPScore, P.m_MinFwdScore, xout.data(),
                                        xstatus.data(), xpaths, xbytes + 16),
              "rsk_mkf_chain_align_pairs");
    for (size_t r = 0; r < recs.size(); ++r)
        if (slot[r] != (size_t) -1 && xstatus[slot[r]] == 3) { host_recs.push_back(r); slot[r] = (size_t) -1; }
    // second batch: chained on the host threads
    std::vector<rsk_aln> yout;
    std::vector<uint8_t> ystatus;
    std::unique_ptr<char[]> ypaths_mem;
    std::vector<size_t> yslot(recs.size(), (size_t) -1);
    if (!host_recs.empty()) {
        std::vector<Chained> chains(host_recs.size());
        const size_t saved_T = recs.size();
        (void) saved_T;
        std::atomic<size_t> nexth{0};
        auto body = [&]() {
            DSSAligner DA;
            DA.SetParams(P);
            for (;;) {
                const size_t h = nexth.fetch_add(1);
                if (h >= host_recs.size()) break;
                const size_t r = host_recs[h];
                const Rec &R = recs[r];
                DA.m_MKF.SetSeedHSPs(R.kept.data(), R.nkept);
                if (DA.m_MKF.m_BestChainScore <= 0) continue;             // PostAlignMKF dssaligner.cpp:1397
                Chained &C = chains[h];
                C.lo_a.assign(DA.m_MKF.m_ChainHSPLois.begin(), DA.m_MKF.m_ChainHSPLois.end());
                C.lo_b.assign(DA.m_MKF.m_ChainHSPLojs.begin(), DA.m_MKF.m_ChainHSPLojs.end());
                C.len.assign(DA.m_MKF.m_ChainHSPLens.begin(), DA.m_MKF.m_ChainHSPLens.end());
            }
            DA.UnsetQuery();
        };
        {
            const unsigned TH = (unsigned) std::max<size_t>(1, std::min<size_t>(T, host_recs.size() / 8 + 1));
            std::vector<std::thread> ts;
            std::vector<std::string> errs(TH);
            for (unsigned t = 0; t < TH; ++t)
                ts.emplace_back([&, t]() { try { body(); } catch (const std::exception &e) { errs[t] = e.what(); } });
            for (auto &t : ts) t.join();
            for (auto &e : errs)
                if (!e.empty()) throw std::runtime_error(e);
        }
        std::vector<uint32_t> ya, yb, yfirst(1, 0);
        std::vector<int32_t> yla, ylb, ylen;
        size_t ybytes = 0;
        for (size_t h = 0; h < host_recs.size(); ++h) {
            if (chains[h].len.empty()) continue;
            const size_t r = host_recs[h];
            yslot[r] = ya.size();
            const uint i = Pairs[recs[r].pair].first, j = Pairs[recs[r].pair].second;
            ya.push_back(i); yb.push_back(j);
            yla.insert(yla.end(), chains[h].lo_a.begin(), chains[h].lo_a.end());
            ylb.insert(ylb.end(), chains[h].lo_b.begin(), chains[h].lo_b.end());
            ylen.insert(ylen.end(), chains[h].len.begin(), chains[h].len.end());
            yfirst.push_back((uint32_t) yla.size());
            ybytes += (size_t) SrcA.m_DBChains[i]->GetSeqLength() + SrcB.m_DBChains[j]->GetSeqLength() + 1;
        }
        yout.resize(ya.size());
        ystatus.resize(ya.size());
        ypaths_mem.reset(new char[ybytes + 16]);
        if (!ya.empty())
            check(rsk_mkf_align_pairs(Ctx, SrcA.m_Db, SrcB.m_Db, ya.data(), yb.data(), ya.size(), yfirst.data(), yla.data(), ylb.data(), ylen.data(),
                                      float(P.m_MKF_X2), P.m_GapOpen, P.m_GapExt, P.m_MKF_MinMegaHSPScore, P.m_MinFwdScore, yout.data(), ystatus.data(),
                                      ypaths_mem.get(), ybytes + 16),
                  "rsk_mkf_align_pairs");
    }
    if (getenv("RSK_TRACE") && !host_recs.empty())
        fprintf(stderr, "[RunMKFPairs] %zu pairs chained on the host (chains tied under qsort)\n", host_recs.size());
    const auto t_host1 = std::chrono::steady_clock::now();
    // stage 3 (host threads): the aligned pairs become DSSAligner results and go to the caller
    std::mutex lock;
    parallel([&](DSSAligner &DA, size_t r, unsigned worker) {
        const Rec &R = recs[r];
        const uint i = Pairs[R.pair].first, j = Pairs[R.pair].second;
        // status 2 (the start XDropHSP derives lies outside a chain: only possible for chains shorter than 8, where the
        // reference's own extents wrap around) counts as "no alignment"
        const rsk_aln *aln = nullptr;
        const char *path = nullptr;
        if (slot[r] != (size_t) -1) {
            const size_t k = slot[r];
            if (xstatus[k] == 1 && xout[k].path_len) { aln = &xout[k]; path = xpaths + xout[k].path_off; }
        } else if (yslot[r] != (size_t) -1) {
            const size_t k = yslot[r];
            if (ystatus[k] == 1 && yout[k].path_len) { aln = &yout[k]; path = ypaths_mem.get() + yout[k].path_off; }
        }
        if (!aln) return;                                                    // nothing to report (m_Path empty)
        DA.ClearAlign();
        DA.m_ChainA = SrcA.m_DBChains[i]; DA.m_ProfileA = SrcA.m_DBProfiles[i];
        DA.m_ChainB = SrcB.m_DBChains[j]; DA.m_ProfileB = SrcB.m_DBProfiles[j];
        DA.m_SelfRevScoreA = SrcA.m_DBSelfRevScores[i]; DA.m_SelfRevScoreB = SrcB.m_DBSelfRevScores[j];
        DA.SetFromAln(*aln, path);
        if (OnHitOfWorker) { (*OnHitOfWorker)(DA, i, j, worker); return; }
        std::lock_guard<std::mutex> g(lock);
        OnHit(DA, i, j);
    });
    if (getenv("RSK_TRACE"))
        fprintf(stderr, "[RunMKFPairs] %zu pairs with chained HSPs through the device batch: chaining + batch %.3f ms, replay %.3f ms (%u threads)\n", nx,
                std::chrono::duration<double, std::milli>(t_host1 - t_host0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host1).count(), T);
}

// Self with SelfOffset >= 0 is (part of) one SHARD of a self search: B = the chains [SelfOffset, SelfOffset + NB)
// of the set, A = its chains [0, NA) with NA = SelfOffset (the rectangle above the shard's triangle) or up to
// SelfOffset + NB; the pairs i <= SelfOffset + j are scored.
// Win (r06; Self, SrcA = S, a Mu filter): one shard of the WHOLE set's triangle as a window of its length order.  The reference
// deals the pairs of RunSelf to its threads one at a time through a locked counter (runself.cpp:72-99): any thread takes any
// pair.  Here a rank takes (a) the pairs whose longer member stands in its window of the length order -- the order the
// triangle mode of the Mu filter walks its targets in, so the shard's filter pass is ONE launch of the whole triangle's shape
// against fewer targets, windows of equal DP cells -- and (b) one contiguous ShardCount-th of the long-chain list, whose pairs all
// hold one of the few longest chains and would otherwise all fall to the last window.  (r01-r05: a target range of the chain
// ORDER, as a rectangle + a small triangle, two passes each with its own filter, survivor lists and long-chain job: measured
// r06 on the 11,211-chain set, N = 8: slowest shard 2.7 x the fastest, efficiency 0.33.)
void RunPairs(DBSearcher &S, DBSearcher &SrcA, bool Self, int64_t SelfOffset, const SelfWindow *Win)
{
    PhaseTimer tm;
    const DSSParams &P = *S.m_Params;
    rsk_ctx *ctx = S.m_Ctx;
    const uint NA = SrcA.GetDBChainCount(), NB = S.GetDBChainCount();
    const bool UseMu = P.m_Omega > 0;            // LoadDB keeps Mu letters only when Omega > 0 (dbsearcher.cpp:249-251)
    const bool Tri = Self && SelfOffset < 0;     // the whole triangle in one call
    const uint joff = SelfOffset > 0 ? (uint) SelfOffset : 0;
    if (Win && !(Tri && UseMu && &SrcA == &S && Win->RankLo <= Win->RankHi && Win->RankHi <= NB && Win->ShardCount >= 1))
        throw std::runtime_error("RunPairs: a window needs the whole set's triangle and a Mu filter");
    const std::vector<uint32_t> *Rank = nullptr; // length rank of the chains (window mode)
    if (Win) {
        check(rsk_build_len_perm(S.m_Db), "rsk_build_len_perm");
        Rank = &S.m_Db->h_len_rank;
    }
    auto InShard = [&](uint i, uint j) { return !Self || i <= joff + j; };
    auto IsMKF = [&](uint i, uint j) {           // DSSAligner::DoMKF dssaligner.cpp:715-732
        if (!UseMu) return false;
        if (SrcA.m_DBMuKmersVec[i]->empty() || S.m_DBMuKmersVec[j]->empty()) return false;
        return SrcA.m_DBChains[i]->GetSeqLength() >= P.m_MKFL || S.m_DBChains[j]->GetSeqLength() >= P.m_MKFL;
    };
    auto Skip = [&](uint i, uint j) {
        if (!S.m_Opts.noself) return false;
        return Self ? (i == joff + j) : (SrcA.m_DBChains[i]->m_Label == S.m_DBChains[j]->m_Label);
    };
    uint64_t SelfTotal = 0;                      // pairs of this (shard of the) triangle
    if (Self && !Win) for (uint j = 0; j < NB; ++j) SelfTotal += std::min<uint64_t>(NA, (uint64_t) joff + j + 1);
    if (Win) SelfTotal = ((uint64_t) Win->RankHi * (Win->RankHi + 1) - (uint64_t) Win->RankLo * (Win->RankLo + 1)) / 2;      // position p closes p + 1 pairs
    std::vector<uint32_t> ia, ib;                // pairs for the full alignment
    std::vector<std::pair<uint32_t, uint32_t> > mkf;
    uint64_t npairs = 0;
    bool Dense = false;                          // no Mu filter, no skipped pairs, not a self search: every pair of SrcA x S in row-major order
    // long-chain pairs: MKF path (dssaligner.cpp:809-813), one aligner per host thread as in the reference
    // (dbsearcher.cpp:98-106); BaseOnAln serialises the output under m_Lock.
    auto each_orientation = [&](DSSAligner &DA, uint i, uint j, auto &&fn) {
        if (DA.m_Path.empty()) return;
        if (Self) {
            fn(DA, true);
            if (i != joff + j) fn(DA, false);
        } else
            fn(DA, false);
    };
    // A plain DBSearcher runs its two jobs side by side: the long-chain job on a context of its own, its hit lines collected
    // in memory (one buffer per worker thread: formatting under one lock was a tenth of the job) and appended after the
    // alignment job's -- the order of the output file stays: Smith-Waterman hits, then long-chain hits.
    struct sink { std::string lines; uint64_t hits = 0; char pad[64]; };
    std::vector<sink> sinks(HostThreads(128));
    const std::function<void(DSSAligner &, uint, uint, unsigned)> on_hit = [&](DSSAligner &DA, uint i, uint j, unsigned worker) {
        sink &me = sinks[worker];
        each_orientation(DA, i, j, [&](DSSAligner &D, bool Up) {
            if (S.Reject(D, Up)) return;
            ++me.hits;
            if (S.m_fTsv && !(S.m_Opts.noself && D.m_ChainA->m_Label == D.m_ChainB->m_Label)) D.AppendTsv(me.lines, Up);
        });
    };
    const bool may_overlap = !S.m_HasOnAlnOverride && !(getenv("RSK_MKF_OVERLAP") && atoi(getenv("RSK_MKF_OVERLAP")) == 0);
    SecondaryCtx own;
    std::future<void> job;
    auto start_mkf_job = [&]() {
        if (!may_overlap || mkf.empty()) return;
        own.Create(ctx->device, "mkf");
        job = std::async(std::launch::async, [&]() { RunMKFPairs(own.c, P, S.m_Opts.columns, SrcA, S, mkf, [](DSSAligner &, uint, uint) {}, &on_hit); });
    };
    struct JobJoin { std::future<void> &j; ~JobJoin() { if (j.valid()) j.wait(); } } join_on_exit{ job };      // an exception below must not leave the job running
    if (UseMu) {
        // Mu filter over the whole enumerated pair space on the GPU
        // The kernel keeps one chain's profile in LDS and streams the other set past it.  The Mu matrix is symmetric and
        // SW(rev(A), B) = SW(A, rev(B)) = SW(rev(B), A), so fwd, rev and the saturation flags do not depend on which chain
        // plays which role: with a small query set against a large database the queries take the profile side
        // (125,000 profiles x 256 partners each would rebuild a profile per 26 wave passes).
        const bool Swap = !Self && NA > NB;
        rsk_db *FilterQ = Swap ? S.m_Db : SrcA.m_Db, *FilterT = Swap ? SrcA.m_Db : S.m_Db;
        const size_t ldo = Swap ? NA : NB;
        const uint64_t total = Self ? SelfTotal : (uint64_t) NA * NB;
        uint64_t nmkf = 0, nskip = 0;
        // MKF pairs = either chain >= m_MKFL (both with k-mers): enumerated from the list of long chains,
        // not by walking the whole pair space
        // (the list -- 0.67 M pairs, ~5 ms of one host thread on the SCOP40-sized self search -- is not needed before the filter
        // has run: it is built on a thread of its own under the filter kernels, r05)
        uint64_t mkf_in_window = 0;              // window mode: long-chain pairs whose longer member stands in the window (they pass through the filter launch unused)
        // ... and the whole list, of which the shard takes one contiguous Count-th (the list is ordered by its first chain: a
        // contiguous piece names an N-th of the chains as seeding queries, and the k-mer tables of the job -- one per query chain
        // of the batch, 85 MB / 7-9 ms of kernels for all 11,211 chains -- shrink with it; dealt round-robin every shard built
        // every table)
        std::vector<std::pair<uint32_t, uint32_t> > mkf_all;
        auto take_mkf = [&](uint i, uint j) {
            if (!Win) { mkf.emplace_back(i, j); ++nmkf; return; }
            const uint32_t rmax = std::max((*Rank)[i], (*Rank)[j]);
            if (rmax >= Win->RankLo && rmax < Win->RankHi) ++mkf_in_window;
            mkf_all.emplace_back(i, j);
        };
        auto deal_mkf = [&]() {
            if (!Win) return;
            // pairs whose FIRST chain is short (its partner is the long one): whole rows of the list (below); pairs of two long chains -- a few
            // per cent of the list, but the ones whose seeds extend and chain -- every Count-th (as one contiguous piece they all fell
            // to the last shard: 93 ms against 58-65 for the others at N = 8)
            std::vector<std::pair<uint32_t, uint32_t> > both_long;
            size_t w = 0;
            for (size_t k = 0; k < mkf_all.size(); ++k) {
                if (SrcA.m_DBChains[mkf_all[k].first]->GetSeqLength() >= P.m_MKFL) both_long.push_back(mkf_all[k]);
                else mkf_all[w++] = mkf_all[k];
            }
            mkf_all.resize(w);
            // (blocks of 16 consecutive first chains, dealt cyclically: a shard still names only an N-th of the chains as seeding
            // queries, and rows that happen to be expensive -- a first chain with seeds against many long chains -- are spread: as
            // ONE contiguous piece per shard two of 8 shards of the look-alike .bca set took 0.46 / 0.57 s against 0.27)
            {
                uint32_t prev = UINT32_MAX;
                uint64_t row = UINT64_MAX;          // ordinal of the current first chain among the list's first chains
                for (const auto &pr : mkf_all) {
                    if (pr.first != prev) { prev = pr.first; ++row; }
                    if ((row / 16) % Win->ShardCount == Win->ShardIndex) mkf.push_back(pr);
                }
            }
            for (size_t k = Win->ShardIndex; k < both_long.size(); k += Win->ShardCount) mkf.push_back(both_long[k]);
            nmkf = mkf.size();
            std::vector<std::pair<uint32_t, uint32_t> >().swap(mkf_all);
        };
        auto build_mkf_list = [&]() {
            std::vector<uint32_t> longB;
            for (uint j = 0; j < NB; ++j)
                if (!S.m_DBMuKmersVec[j]->empty() && S.m_DBChains[j]->GetSeqLength() >= P.m_MKFL) longB.push_back(j);
            for (uint i = 0; i < NA; ++i) {
                if (SrcA.m_DBMuKmersVec[i]->empty()) continue;
                const uint j0 = Self ? (i > joff ? i - joff : 0) : 0;
                if (SrcA.m_DBChains[i]->GetSeqLength() >= P.m_MKFL) {
                    for (uint j = j0; j < NB; ++j) {
                        if (S.m_DBMuKmersVec[j]->empty() || Skip(i, j)) continue;
                        take_mkf(i, j);
                    }
                } else {
                    for (auto it = std::lower_bound(longB.begin(), longB.end(), j0); it != longB.end(); ++it) {
                        if (Skip(i, *it)) continue;
                        take_mkf(i, *it);
                    }
                }
            }
            deal_mkf();
        };
        // (r06, window shards, N = 8 on the 11,211-chain set: with the job under the filter a shard took 82.9 ms against 80.0 ms behind it
        // -- the seeding kernels take 22 ms instead of 7 beside the filter's and the filter is no shorter: the default stays)
        const bool early = getenv("RSK_MKF_EARLY") && atoi(getenv("RSK_MKF_EARLY")) == 1;
        std::future<void> mkf_list;
        if (early) build_mkf_list();
        else mkf_list = std::async(std::launch::async, build_mkf_list);      // (a future of std::async joins in its destructor: no exit path leaves the thread behind)
        if (S.m_Opts.noself) {
            if (Win) nskip = Win->RankHi - Win->RankLo;                               // the diagonal pairs of the window's positions
            else if (Self) nskip = NA > joff ? std::min<uint64_t>(NB, NA - joff) : 0;      // the diagonal pairs this pass holds
            else {
                std::unordered_map<std::string, uint32_t> cntB;
                for (uint j = 0; j < NB; ++j) ++cntB[S.m_DBChains[j]->m_Label];
                for (uint i = 0; i < NA; ++i) {
                    auto it = cntB.find(SrcA.m_DBChains[i]->m_Label);
                    if (it != cntB.end()) nskip += it->second;
                }
            }
        }
        // The long-chain job does not depend on the filter (its pairs are known from the chain lengths), so it COULD start now,
        // on a context of its own, under the filter kernels.  Measured (r04): no consistent gain -- 256 x 1,000,000: 15.3-17.4 s
        // against 14.8-16.8 s, the filter and the seeding kernels are both VALU-bound and share the SIMDs -- so the default
        // stays "beside the alignment job" (below); RSK_MKF_EARLY=1 starts it here.
        if (early) start_mkf_job();
        tm.lap("  long-chain pair list (started)");
        // survivor lists: sized for 1/6 of the pairs (the presets pass 0.3 % of real SCOP40 pairs, 15 % of look-alike synthetic
        // structures; 8 bytes per slot, 5.3 GB for the largest filter tile), re-run with the exact count on overflow
        // (the kernel counts every survivor; it only stops storing at `cap`)
        const uint64_t dense = Tri ? total : (uint64_t) NA * NB;
        size_t cap = (size_t) std::min<uint64_t>(dense, std::max<uint64_t>(1u << 22, dense / 6));
        auto hipok = [](hipError_t e, const char *w) { if (e != hipSuccess) throw std::runtime_error(std::string(w) + ": " + hipGetErrorString(e)); };
        DeviceBuffer Fwd(ctx, (size_t) (Swap ? NB : NA) * ldo, "filter score matrix"), Count(ctx, 4, "survivor counter"), ListQ, ListT;
        uint32_t ns = 0;
        for (;;) {
            ListQ.Make(ctx, cap * 4, "survivor list");
            ListT.Make(ctx, cap * 4, "survivor list");
            if (Win)
                check(rsk_mu_filter_window_dev(ctx, S.m_Db, Win->RankLo, Win->RankHi, P.m_ParaMuGapOpen, P.m_ParaMuGapExt, P.m_Omega, P.m_OmegaFwd,
                                               Fwd.As<uint8_t>(), ldo, ListQ.As<uint32_t>(), ListT.As<uint32_t>(), nullptr, nullptr, cap, Count.As<uint32_t>()),
                      "rsk_mu_filter_window_dev");
            else
            check(rsk_mu_filter_dev(ctx, FilterQ, FilterT, Tri ? 1 : 0, P.m_ParaMuGapOpen, P.m_ParaMuGapExt, P.m_Omega, P.m_OmegaFwd, Fwd.As<uint8_t>(), ldo,
                                    ListQ.As<uint32_t>(), ListT.As<uint32_t>(), nullptr, nullptr, cap, Count.As<uint32_t>()),
                  "rsk_mu_filter_dev");
            check(rsk_ctx_sync(ctx), "rsk_ctx_sync");                  // the filter is queued on the context's stream; the copies below are not
            hipok(hipMemcpy(&ns, Count.As<uint32_t>(), 4, hipMemcpyDeviceToHost), "copy n");
            if (ns <= cap) break;
            cap = ns;
        }
        if (mkf_list.valid()) mkf_list.get();
        tm.lap("  Mu filter kernels");
        // deterministic order (the device list is unordered): by A-side chain, then B-side chain -- the order the reference walks
        // its pairs in (runself.cpp:72-99, runquery.cpp:82) -- sorted on the device (8.7 M survivors through a host counting
        // sort + per-chain sorts were 0.15 s), the two columns arrive ordered
        uint32_t *const dA = Swap ? ListT.As<uint32_t>() : ListQ.As<uint32_t>(), *const dB = Swap ? ListQ.As<uint32_t>() : ListT.As<uint32_t>();
        const bool sort_on_device = ns <= 0x7FFFFFFFu;                // the device sort's item count is 31 bits; beyond it the host orders the list
        if (sort_on_device) check(rsk_pairs_sort_dev(ctx, dA, dB, ns, (uint32_t) NA), "rsk_pairs_sort_dev");
        std::vector<uint32_t> pa(ns), pb(ns);
        hipok(hipMemcpy(pa.data(), dA, (size_t) ns * 4, hipMemcpyDeviceToHost), "copy pairs");
        hipok(hipMemcpy(pb.data(), dB, (size_t) ns * 4, hipMemcpyDeviceToHost), "copy pairs");
        if (!sort_on_device) {
            std::vector<uint64_t> key(ns);
            rsk_parallel_for(ns, 1 << 20, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) key[k] = ((uint64_t) pa[k] << 32) | pb[k]; });
            std::sort(key.begin(), key.end());
            rsk_parallel_for(ns, 1 << 20, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) { pa[k] = (uint32_t) (key[k] >> 32); pb[k] = (uint32_t) key[k]; } });
        }
        Fwd.Free(); ListQ.Free(); ListT.Free(); Count.Free();
        tm.lap("  survivors: device sort + d2h");
        {
            // the pairs this pass aligns: survivors of its shard that are neither skipped (-noself) nor long-chain pairs
            // (slices on the host threads, concatenated in order)
            const size_t nsl = std::max<size_t>(1, std::min<size_t>(64, ns / 65536 + 1));
            std::vector<std::vector<uint32_t> > sa(nsl), sb(nsl);
            rsk_parallel_for(nsl, 1, [&](size_t lo, size_t hi) {
                for (size_t sl = lo; sl < hi; ++sl) {
                    const size_t k0 = (size_t) ns * sl / nsl, k1 = (size_t) ns * (sl + 1) / nsl;
                    sa[sl].reserve(k1 - k0); sb[sl].reserve(k1 - k0);
                    for (size_t k = k0; k < k1; ++k) {
                        const uint i = pa[k], j = pb[k];
                        if (!InShard(i, j) || Skip(i, j) || IsMKF(i, j)) continue;
                        sa[sl].push_back(i); sb[sl].push_back(j);
                    }
                }
            });
            size_t tot = 0;
            for (size_t sl = 0; sl < nsl; ++sl) tot += sa[sl].size();
            ia.reserve(tot); ib.reserve(tot);
            for (size_t sl = 0; sl < nsl; ++sl) { ia.insert(ia.end(), sa[sl].begin(), sa[sl].end()); ib.insert(ib.end(), sb[sl].begin(), sb[sl].end()); }
        }
        tm.lap("  alignment pair list");
        // window mode: the shard's pairs = its window's pairs that are not long-chain pairs + its share of the long-chain list
        // (the shards' counts add up to the unsharded call's)
        npairs = Win ? total - nskip - mkf_in_window + nmkf : total - nskip;
        S.m_MKFPairCount = nmkf;
        S.m_MuFilterInputCount = npairs - nmkf;
        S.m_MuFilterDiscardCount = S.m_MuFilterInputCount - ia.size();
    } else if (!Self && !S.m_Opts.noself && !(getenv("RSK_DENSE_PAIR_LISTS") && atoi(getenv("RSK_DENSE_PAIR_LISTS")) == 1)) {
        // the whole rectangle, row-major: no pair list (ForEachAlignedBatchDense)
        Dense = true;
        npairs = (uint64_t) NA * NB;
    } else {
        // every pair of the enumerated space (tens of millions for a query batch against a DB batch): row starts by a
        // prefix sum, rows filled on the host threads
        std::vector<uint64_t> first((size_t) NA + 1, 0);
        rsk_parallel_for(NA, 4096, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const uint j0 = Self ? ((uint) i > joff ? (uint) i - joff : 0) : 0;
                uint64_t c = NB > j0 ? NB - j0 : 0;
                if (S.m_Opts.noself)
                    for (uint j = j0; j < NB; ++j) c -= Skip((uint) i, j) ? 1 : 0;
                first[i + 1] = c;
            }
        });
        for (uint i = 0; i < NA; ++i) first[i + 1] += first[i];
        npairs = first[NA];
        ia.resize(npairs); ib.resize(npairs);
        rsk_parallel_for(NA, 256, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                uint64_t k = first[i];
                for (uint j = Self ? ((uint) i > joff ? (uint) i - joff : 0) : 0; j < NB; ++j) {
                    if (Skip((uint) i, j)) continue;
                    ia[k] = (uint32_t) i; ib[k] = j;
                    ++k;
                }
            }
        });
    }
    S.m_ProcessedPairCount = npairs;
    S.m_AlnCount = npairs - mkf.size();
    tm.lap("filter + pair lists");
    auto align = [&]() {
        const OnBatchFn on_batch = [&](const std::vector<uint32_t> &bia, const std::vector<uint32_t> &bib, const std::vector<rsk_aln> &out, const char *paths) {
            ReplayBatch(S, SrcA, S, bia, bib, out, paths, Self, joff);
        };
        if (Dense) ForEachAlignedBatchDense(P, ctx, S.m_Opts, SrcA, S, on_batch);
        else ForEachAlignedBatch(P, ctx, S.m_Opts, SrcA, S, ia, ib, on_batch);
    };
    if (!job.valid()) start_mkf_job();             // the default: beside the alignment job
    if (job.valid()) {
        align();
        job.get();
        tm.lap("align + replay | long-chain job side by side");
        for (sink &me : sinks) {
            S.m_HitCount += me.hits;
            if (!me.lines.empty() && fwrite(me.lines.data(), 1, me.lines.size(), S.m_fTsv) != me.lines.size()) throw std::runtime_error("short write to the hits file");
        }
        return;
    }
    align();
    tm.lap("align + replay");
    RunMKFPairs(ctx, P, S.m_Opts.columns, SrcA, S, mkf, [&](DSSAligner &DA, uint i, uint j) {
        each_orientation(DA, i, j, [&](DSSAligner &D, bool Up) { S.BaseOnAln(D, Up); });
    });
    tm.lap("MKF (GPU seeds + host)");
}

uint64_t RunMKFPairsBeside(rsk_ctx *Ctx, const DSSParams &Params, const std::string &Columns, DBSearcher &SrcA, DBSearcher &SrcB,
                           const std::vector<std::pair<uint32_t, uint32_t> > &Pairs, const std::function<void()> &AlignJob,
                           const std::function<bool(const DSSAligner &)> &Keep, bool Up, FILE *fTsv)
{
    struct sink { std::string lines; uint64_t hits = 0; char pad[64]; };
    std::vector<sink> sinks(HostThreads(128));
    const std::function<void(DSSAligner &, uint, uint, unsigned)> on_hit = [&](DSSAligner &DA, uint, uint, unsigned worker) {
        if (!Keep(DA)) return;
        sink &me = sinks[worker];
        ++me.hits;
        if (fTsv) DA.AppendTsv(me.lines, Up);
    };
    const bool beside = !Pairs.empty() && !(getenv("RSK_MKF_OVERLAP") && atoi(getenv("RSK_MKF_OVERLAP")) == 0);
    if (beside) {
        SecondaryCtx own;
        own.Create(Ctx->device, "mkf");
        std::future<void> job = std::async(std::launch::async, [&]() {
            RunMKFPairs(own.c, Params, Columns, SrcA, SrcB, Pairs, [](DSSAligner &, uint, uint) {}, &on_hit);
        });
        try {
            AlignJob();
        } catch (...) {
            job.wait();
            throw;
        }
        job.get();
    } else {
        AlignJob();
        RunMKFPairs(Ctx, Params, Columns, SrcA, SrcB, Pairs, [](DSSAligner &, uint, uint) {}, &on_hit);
    }
    uint64_t hits = 0;
    for (sink &me : sinks) {
        hits += me.hits;
        if (fTsv && !me.lines.empty() && fwrite(me.lines.data(), 1, me.lines.size(), fTsv) != me.lines.size()) throw std::runtime_error("short write to the hits file");
    }
    return hits;
}

}   // namespace reseek_amd

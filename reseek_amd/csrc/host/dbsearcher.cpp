// dbsearcher.cpp -- host mirror of DBSearcher's drivers (dbsearcher.cpp, runself.cpp:101, runquery.cpp:82): RunSelf,
// RunQuery (a loaded set, or a .bca file streamed in batches with a loader thread), shards of the pair space
// and the fan-out over a device list; DSSAligner::AlignQueryTarget (the per-pair entry point = a batch of one).
// The pair-space work itself is in runpairs.cpp, loading in loaddb.cpp, the C-ABI search entry points in search_api.cpp.
#include "host_internal.h"

namespace reseek_amd {
// Largest dense pair block one Mu-filter pass may cover: the forward-score matrix is one byte per pair in HBM and the
// survivor counter is 32 bits.  RSK_FILTER_TILE_PAIRS lowers it (tests).
static uint64_t FilterTilePairs()
{
    uint64_t v = 4ull << 30;
    if (const char *e = getenv("RSK_FILTER_TILE_PAIRS")) { const long long x = atoll(e); if (x > 0) v = (uint64_t) x; }
    return std::min<uint64_t>(v, 0xFFFFFFFFull);
}

struct PairCounters {
    uint64_t pairs = 0, alns = 0, fin = 0, fdis = 0, mkf = 0;
    void add(const DBSearcher &S) { pairs += S.m_ProcessedPairCount; alns += S.m_AlnCount; fin += S.m_MuFilterInputCount; fdis += S.m_MuFilterDiscardCount; mkf += S.m_MKFPairCount; }
    void store(DBSearcher &S) const { S.m_ProcessedPairCount = pairs; S.m_AlnCount = alns; S.m_MuFilterInputCount = fin; S.m_MuFilterDiscardCount = fdis; S.m_MKFPairCount = mkf; }
};

// The triangle of the chains [Lo, Hi) of `Set` plus the rectangle chains[0, Lo) x chains[Lo, Hi) above it, in target
// blocks whose filter passes stay below the tile size.  (0, N) = the whole self search; a shard passes its own range.
static void RunSelfRange(DBSearcher &Set, uint Lo, uint Hi, FILE *fTsv, PairCounters &C, uint64_t &Hits, uint64_t &SW)
{
    const uint64_t tile = FilterTilePairs();
    uint b0 = Lo;
    while (b0 < Hi) {
        // widest block [b0, b1) with b1 * (b1 - b0) <= tile pairs (at least one target)
        uint b1 = b0 + 1;
        {
            uint lo = b0 + 1, hi = Hi;
            while (lo < hi) {
                const uint mid = lo + (hi - lo + 1) / 2;
                if ((uint64_t) mid * (mid - b0) <= tile) lo = mid; else hi = mid - 1;
            }
            b1 = std::max(b0 + 1, lo);
        }
        DBSearcher B;
        B.MakeView(Set, b0, b1);
        B.Setup();
        B.m_fTsv = fTsv;
        B.UploadToGpu();
        if (b0 > 0) {
            DBSearcher A;
            A.MakeView(Set, 0, b0);
            A.UploadToGpu();
            RunPairs(B, A, true, (int64_t) b0);
            C.add(B);
        }
        RunPairs(B, B, true);
        C.add(B);
        Hits += B.m_HitCount;
        SW += B.m_SWCount;
        b0 = b1;
    }
}

// ---------------------------------------------------------------------------------------------
// Several devices behind one DBSearcher.  The reference fans the pair space out over threads, one DSSAligner
// each (dbsearcher.cpp:98-106, runself.cpp:72-101, runquery.cpp:82-125); here the fan-out is over DEVICES: one context and
// one host thread per entry of m_Devices, each running one shard of the pair space -- a target range of the triangle
// balanced by DP cells (self search) or a contiguous range of the -db chains balanced by residues, the other side
// replicated -- through the same single-device code.  No collective on the data path: every shard appends its hit lines
// to the searcher's hits file (whole lines per fwrite) and the counters are summed.
// ---------------------------------------------------------------------------------------------
std::vector<int> DBSearcher::ParseDeviceList(const char *Str)
{
    std::vector<int> v;
    if (!Str) return v;
    const char *p = Str;
    while (*p) {
        while (*p == ',' || *p == ' ') ++p;
        if (!*p) break;
        char *end = nullptr;
        const long d = strtol(p, &end, 10);
        if (end == p || d < 0 || d > 1023) throw std::runtime_error(std::string("RSK_DEVICES: cannot parse \"") + Str + "\"");
        v.push_back((int) d);
        p = end;
    }
    return v;
}

void DBSearcher::SelfShardRange(const uint32_t *Lens, uint64_t N, uint Index, uint Count, uint64_t &Lo, uint64_t &Hi)
{
    // cells of the pairs (i <= j) with target j < t, for every t: the shard bounds cut this curve into equal parts
    std::vector<double> cum(N + 1, 0.0);
    double pre = 0;
    for (uint64_t j = 0; j < N; ++j) { pre += Lens[j]; cum[j + 1] = cum[j] + pre * Lens[j]; }
    auto bound = [&](uint r) { return r >= Count ? N : (uint64_t) (std::lower_bound(cum.begin(), cum.end(), cum[N] * r / Count) - cum.begin()); };
    Lo = std::min(N, bound(Index));
    Hi = std::max(Lo, std::min(N, bound(Index + 1)));
}

void DBSearcher::ResidueShardRange(const uint32_t *Lens, uint64_t N, uint Index, uint Count, uint64_t &Lo, uint64_t &Hi)
{
    std::vector<uint64_t> cum(N + 1, 0);
    for (uint64_t i = 0; i < N; ++i) cum[i + 1] = cum[i] + Lens[i];
    auto bound = [&](uint r) { return r >= Count ? N : (uint64_t) (std::lower_bound(cum.begin(), cum.end(), cum[N] * r / Count) - cum.begin()); };
    Lo = std::min(N, bound(Index));
    Hi = std::max(Lo, std::min(N, bound(Index + 1)));
}

namespace {

// a searcher over the same chains as `Of` that runs on another context: borrowed chains, own device arrays
void MakeReplica(DBSearcher &R, const DBSearcher &Of, rsk_ctx *Ctx)
{
    R.MakeView(Of, 0, Of.GetDBChainCount());
    R.m_Ctx = Ctx;
    R.m_fTsv = Of.m_fTsv;
    R.Setup();
    R.m_MaxEvalue = Of.m_MaxEvalue;
    R.m_StreamBatchChains = Of.m_StreamBatchChains;
    R.m_StreamBatchResidues = Of.m_StreamBatchResidues;
}

// runs Shard(k, replica k) for every device on a thread of its own; counters of the replicas are summed into `Into`
void OnEveryDevice(DBSearcher &Into, const std::function<void(uint, DBSearcher &)> &Shard)
{
    const uint D = (uint) Into.m_Devices.size();
    DeviceTeam Team(Into.m_Devices);
    std::vector<std::unique_ptr<DBSearcher> > Rep(D);
    std::vector<std::exception_ptr> Err(D);
    std::vector<std::thread> Th;
    for (uint k = 0; k < D; ++k)
        Th.emplace_back([&, k]() {
            try {
                (void) hipSetDevice(Into.m_Devices[k]);
                Rep[k].reset(new DBSearcher);
                MakeReplica(*Rep[k], Into, Team.ctx(k));
                Shard(k, *Rep[k]);
            } catch (...) {
                Err[k] = std::current_exception();
            }
        });
    for (auto &t : Th) t.join();
    for (uint k = 0; k < D; ++k)
        if (Err[k]) std::rethrow_exception(Err[k]);
    uint64_t pairs = 0, alns = 0, fin = 0, fdis = 0, mkf = 0, hits = 0, sw = 0;
    for (uint k = 0; k < D; ++k) {
        const DBSearcher &R = *Rep[k];
        pairs += R.m_ProcessedPairCount; alns += R.m_AlnCount; fin += R.m_MuFilterInputCount; fdis += R.m_MuFilterDiscardCount;
        mkf += R.m_MKFPairCount; hits += R.m_HitCount; sw += R.m_SWCount;
    }
    Into.m_ProcessedPairCount = pairs; Into.m_AlnCount = alns; Into.m_MuFilterInputCount = fin; Into.m_MuFilterDiscardCount = fdis;
    Into.m_MKFPairCount = mkf; Into.m_HitCount += hits; Into.m_SWCount += sw;
}
}   // namespace

bool DBSearcher::OnSeveralDevices() const { return m_Devices.size() > 1 && !m_HasOnAlnOverride; }

void DBSearcher::RunSelf()
{
    if (!m_fTsv) m_fTsv = g_fTsv;
    if (OnSeveralDevices()) {
        const uint D = (uint) m_Devices.size();
        OnEveryDevice(*this, [D](uint k, DBSearcher &Replica) { Replica.RunSelfShard(k, D); });
        return;
    }
    const uint N = GetDBChainCount();
    if (m_HasOnAlnOverride || (uint64_t) N * N <= FilterTilePairs()) {
        UploadToGpu();
        RunPairs(*this, *this, true);
        return;
    }
    // a set whose dense pair matrix exceeds one filter pass (~65 k chains): target blocks, as the shards of a multi-GPU run
    PairCounters C;
    uint64_t Hits = 0, SW = 0;
    if (m_Db) { rsk_db_destroy(m_Db); m_Db = nullptr; }      // as in RunSelfShard: the blocks upload their own views
    RunSelfRange(*this, 0, N, m_fTsv, C, Hits, SW);
    C.store(*this);
    m_HitCount += Hits;
    m_SWCount += SW;
}

// chains [Lo, Hi) of Src as a searcher of their own (borrowed pointers)
void DBSearcher::MakeView(const DBSearcher &Src, uint Lo, uint Hi)
{
    m_OwnsChains = false;
    m_Params = Src.m_Params; m_Opts = Src.m_Opts; m_Ctx = Src.m_Ctx;
    m_DBChains.assign(Src.m_DBChains.begin() + Lo, Src.m_DBChains.begin() + Hi);
    m_DBProfiles.assign(Src.m_DBProfiles.begin() + Lo, Src.m_DBProfiles.begin() + Hi);
    m_DBMuLettersVec.assign(Src.m_DBMuLettersVec.begin() + Lo, Src.m_DBMuLettersVec.begin() + Hi);
    m_DBMuKmersVec.assign(Src.m_DBMuKmersVec.begin() + Lo, Src.m_DBMuKmersVec.begin() + Hi);
    m_DBSelfRevScores.assign(Src.m_DBSelfRevScores.begin() + Lo, Src.m_DBSelfRevScores.begin() + Hi);
}

// Windows of the LENGTH ORDER of a set (stable sort by length = rsk_len_rank's order) with equal DP cells of the triangle: position
// p of that order closes the pairs {chain at p, every chain at a position <= p}.
void DBSearcher::SelfWindowRange(const uint32_t *Lens, uint64_t N, uint Index, uint Count, uint64_t &Lo, uint64_t &Hi)
{
    std::vector<uint32_t> Sorted(Lens, Lens + N);
    std::stable_sort(Sorted.begin(), Sorted.end());
    SelfShardRange(Sorted.data(), N, Index, Count, Lo, Hi);
}

// shard `Index` of `Count` of the self search (the reference: one locked pair counter for all threads, runself.cpp:72-99).
// With a Mu filter and a set whose dense pair matrix fits one filter pass: a window of the set's length order + one Count-th
// of the long-chain pair list (RunPairs, SelfWindow) -- the whole set stays resident, one pass.  Otherwise (no filter: -verysensitive, every
// pair costs its cells; or > ~65 k chains): targets [Lo, Hi) of the chain order with equal cells, as the rectangle
// chains[0, Lo) x chains[Lo, Hi) plus the triangle of chains[Lo, Hi).
void DBSearcher::RunSelfShard(uint Index, uint Count)
{
    const uint N = GetDBChainCount();
    std::vector<uint32_t> Lens(N);
    for (uint j = 0; j < N; ++j) Lens[j] = m_DBChains[j]->GetSeqLength();
    const bool Windows = m_Params->m_Omega > 0 && (uint64_t) N * N <= FilterTilePairs() && !m_HasOnAlnOverride &&
                         !(getenv("RSK_SELF_SHARD_RANGES") && atoi(getenv("RSK_SELF_SHARD_RANGES")) == 1);
    if (Windows) {
        uint64_t Lo64, Hi64;
        SelfWindowRange(Lens.data(), N, Index, Count, Lo64, Hi64);
        SelfWindow W;
        W.RankLo = (uint32_t) Lo64; W.RankHi = (uint32_t) Hi64; W.ShardIndex = Index; W.ShardCount = Count;
        UploadToGpu();
        const uint64_t Hits0 = m_HitCount, SW0 = m_SWCount;
        RunPairs(*this, *this, true, -1, &W);
        m_HitCount -= Hits0; m_SWCount -= SW0;      // (a shard reports its own counts, as the range form below)
        return;
    }
    uint64_t Lo64, Hi64;
    SelfShardRange(Lens.data(), N, Index, Count, Lo64, Hi64);
    const uint Lo = (uint) Lo64, Hi = (uint) Hi64;
    // this shard = the rectangle chains[0, Lo) x chains[Lo, Hi) plus the triangle of chains[Lo, Hi): no pair of the square
    // [Lo, Hi)^2 below its diagonal is ever scored (a single rectangular pass [0, Hi) x [Lo, Hi) made the first shard do
    // twice its share of the filter)
    PairCounters C;
    uint64_t Hits = 0, SW = 0;
    // the blocks below upload their own views: the whole set's device copy (left by the self-rev pass of LoadDB) would only
    // double the HBM the largest sets need; a later call re-uploads it on demand
    if (m_Db) { rsk_db_destroy(m_Db); m_Db = nullptr; }
    if (Hi > Lo) RunSelfRange(*this, Lo, Hi, m_fTsv, C, Hits, SW);
    C.store(*this);
    m_HitCount = Hits;
    m_SWCount = SW;
}

void DBSearcher::RunQuery(DBSearcher &DBChainsSource)
{
    PhaseTimer tm("RunQuery");
    if (!m_fTsv) m_fTsv = g_fTsv;
    if (OnSeveralDevices()) {
        // contiguous ranges of the source chains balanced by residues, our chains replicated on every device
        const uint D = (uint) m_Devices.size(), NS = DBChainsSource.GetDBChainCount();
        std::vector<uint32_t> Lens(NS);
        for (uint i = 0; i < NS; ++i) Lens[i] = DBChainsSource.m_DBChains[i]->GetSeqLength();
        OnEveryDevice(*this, [&](uint k, DBSearcher &Replica) {
            uint64_t Lo, Hi;
            ResidueShardRange(Lens.data(), NS, k, D, Lo, Hi);
            if (Hi <= Lo) return;
            DBSearcher Part;
            Part.MakeView(DBChainsSource, (uint) Lo, (uint) Hi);
            Replica.RunQuery(Part);
        });
        return;
    }
    UploadToGpu();
    DBChainsSource.m_Ctx = m_Ctx;
    const uint NA = DBChainsSource.GetDBChainCount(), NB = GetDBChainCount();
    const uint64_t tile = FilterTilePairs();
    if (m_HasOnAlnOverride || (uint64_t) NA * NB <= tile || NA <= 1) {
        DBChainsSource.UploadToGpu();
        tm.lap("upload both sets");
        RunPairs(*this, DBChainsSource, false);
        tm.lap("RunPairs");
        return;
    }
    // row blocks of the source so that one filter pass stays below the tile size
    const uint rows = (uint) std::max<uint64_t>(1, tile / std::max<uint>(NB, 1));
    PairCounters C;
    for (uint lo = 0; lo < NA; lo += rows) {
        DBSearcher View;
        View.MakeView(DBChainsSource, lo, std::min(NA, lo + rows));
        View.m_Ctx = m_Ctx;
        View.UploadToGpu();
        RunPairs(*this, View, false);
        C.add(*this);
    }
    C.store(*this);
    tm.lap("RunPairs (row blocks)");
}

// runquery.cpp:18-125.  The reference's threads pull one chain at a time from the reader, featurise it, compute its
// self-rev score and walk it past every loaded chain.  Here the reader's chains are taken in batches (bounded by chains and
// residues, so a -db file of any size needs a fixed amount of host RAM and HBM): a loader thread reads + featurises +
// self-rev-scores batch k + 1 (on a context of its own) while batch k goes through the filter / alignment kernels.
void DBSearcher::RunQuery(ChainReader2 &QCR)
{
    if (!m_Ctx) m_Ctx = DefaultCtx();
    if (!m_fTsv) m_fTsv = g_fTsv;
    if (OnSeveralDevices()) {
        // every device streams its own range of what the reader has left (balanced by residues) past a replica of our chains
        const uint D = (uint) m_Devices.size();
        const uint64_t First = QCR.m_ChainIdx_BCA, End = QCR.m_EndIdx_BCA;
        const std::string FN = QCR.m_CurrentFN;
        const uint32_t *Lens = QCR.m_BCA.m_SeqLengths.data() + First;
        OnEveryDevice(*this, [&](uint k, DBSearcher &Replica) {
            uint64_t Lo, Hi;
            ResidueShardRange(Lens, End - First, k, D, Lo, Hi);
            if (Hi <= Lo) return;
            ChainReader2 Part;
            Part.OpenRange(FN, First + Lo, First + Hi);
            Replica.RunQuery(Part);
        });
        QCR.m_ChainIdx_BCA = End;                          // consumed
        return;
    }
    UploadToGpu();
    SecondaryCtx loader;
    loader.Create(m_Ctx->device, "loader");
    // The first batch is a quarter of the others: nothing can run on the GPU before it is featurised, and the loader has
    // the second (full) batch ready by the time the first has gone through the kernels.
    size_t nloaded = 0;
    auto load = [&]() -> std::unique_ptr<DBSearcher> {
        const auto t_load = std::chrono::steady_clock::now();
        auto ns_since = [](std::chrono::steady_clock::time_point t) { return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); };
        std::vector<PDBChain *> Chains;
        uint64_t nres = 0;
        const size_t maxc = nloaded++ == 0 ? std::max<size_t>(std::min<size_t>(m_StreamBatchChains, 1024), m_StreamBatchChains / 4) : m_StreamBatchChains;
        while (Chains.size() < maxc && nres < m_StreamBatchResidues) {
            PDBChain *C = QCR.GetNext();
            if (!C) break;
            nres += C->GetSeqLength();
            Chains.push_back(C);
        }
        std::unique_ptr<DBSearcher> Src;
        if (Chains.empty()) return Src;
        Src.reset(new DBSearcher);
        Src->m_Params = m_Params;
        Src->m_SelfRevQueryFlavour = true;               // runquery.cpp:43-44: the search params themselves
        Src->m_Opts = m_Opts;
        Src->m_Ctx = loader.c;
        const auto t_feat = std::chrono::steady_clock::now();
        try {
            Src->LoadChains(Chains);
        } catch (...) {
            for (PDBChain *C : Chains) delete C;
            throw;
        }
        g_rsk_counters.featurise_ns += ns_since(t_feat);
        const auto t_up = std::chrono::steady_clock::now();
        Src->UploadToGpu();                              // here, beside the search of the previous batch
        g_rsk_counters.upload_ns += ns_since(t_up);
        g_rsk_counters.db_batches += 1;
        g_rsk_counters.loader_ns += ns_since(t_load);
        return Src;
    };
    uint64_t pairs = 0, alns = 0, mkf = 0, fin = 0, fdis = 0;
    std::future<std::unique_ptr<DBSearcher> > next = std::async(std::launch::async, load);
    std::future<void> teardown;                          // the previous batch (32,768 chains, its device arrays) is freed off this thread
    for (;;) {
        std::unique_ptr<DBSearcher> Src = next.get();
        if (!Src) break;
        next = std::async(std::launch::async, load);
        try {
            RunQuery(*Src);
        } catch (...) {
            next.wait();
            throw;
        }
        if (teardown.valid()) teardown.wait();
        teardown = std::async(std::launch::async, [dead = std::shared_ptr<DBSearcher>(Src.release())]() mutable { dead.reset(); });
        pairs += m_ProcessedPairCount; alns += m_AlnCount; mkf += m_MKFPairCount; fin += m_MuFilterInputCount; fdis += m_MuFilterDiscardCount;
    }
    if (teardown.valid()) teardown.wait();
    m_ProcessedPairCount = pairs; m_AlnCount = alns; m_MKFPairCount = mkf; m_MuFilterInputCount = fin; m_MuFilterDiscardCount = fdis;
}

// One pair through the same GPU kernels (the reference's per-pair entry point, dssaligner.cpp:793).
void DSSAligner::AlignQueryTarget()
{
    ClearAlign();
    if (DoMKF()) { AlignMKF(); return; }
    AlignPairOnGpu();
}

void DSSAligner::AlignPairOnGpu()
{
    if (!m_Ctx) m_Ctx = DefaultCtx();
    // a batch of one on the aligner's context; aligners of several threads may share it (the reference keeps one DSSAligner
    // per thread, dbsearcher.cpp:98-106; an rsk_ctx is not thread-safe)
    std::lock_guard<std::mutex> lock(CtxMutex(m_Ctx));
    auto mk = [&](const PDBChain &C, const std::vector<std::vector<byte> > &Prof, const std::vector<byte> *Mu, float SelfRev) {
        const uint32_t L = C.GetSeqLength();
        std::vector<uint8_t> prof((size_t) L * RSK_NFEAT);
        for (int f = 0; f < RSK_NFEAT; ++f) memcpy(&prof[(size_t) f * L], Prof[f].data(), L);
        rsk_db *db = nullptr;
        check(rsk_db_create(m_Ctx, 1, &L, Mu ? Mu->data() : nullptr, prof.data(), C.m_Xs.data(), C.m_Ys.data(), C.m_Zs.data(), &SelfRev, &db),
              "rsk_db_create");
        return db;
    };
    struct sets { rsk_db *a = nullptr, *b = nullptr; ~sets() { rsk_db_destroy(a); rsk_db_destroy(b); } } two;
    two.a = mk(*m_ChainA, *m_ProfileA, m_MuLettersA, m_SelfRevScoreA);
    two.b = mk(*m_ChainB, *m_ProfileB, m_MuLettersB, m_SelfRevScoreB);
    rsk_db *const a = two.a, *const b = two.b;
    bool pass = true;
    if (m_Params->m_Omega > 0 && m_MuLettersA && m_MuLettersB) {
        auto hipok = [](hipError_t e, const char *w) { if (e != hipSuccess) throw std::runtime_error(std::string(w) + ": " + hipGetErrorString(e)); };
        DeviceBuffer Fwd(m_Ctx, 16, "filter score"), List(m_Ctx, 16, "survivor list");
        uint32_t *d_p = List.As<uint32_t>();
        check(rsk_mu_filter_dev(m_Ctx, a, b, 0, m_Params->m_ParaMuGapOpen, m_Params->m_ParaMuGapExt, m_Params->m_Omega, m_Params->m_OmegaFwd,
                                Fwd.As<uint8_t>(), 1, d_p, d_p + 1, nullptr, nullptr, 1, d_p + 2),
              "rsk_mu_filter_dev");
        check(rsk_ctx_sync(m_Ctx), "rsk_ctx_sync");
        uint32_t n = 0;
        hipok(hipMemcpy(&n, d_p + 2, 4, hipMemcpyDeviceToHost), "hipMemcpy");
        pass = n > 0;
    }
    if (pass) {
        const uint32_t z = 0;
        rsk_aln out;
        std::vector<char> paths(m_ChainA->GetSeqLength() + m_ChainB->GetSeqLength() + 2);
        check(rsk_align_pairs(m_Ctx, a, b, &z, &z, 1, m_Params->m_GapOpen, m_Params->m_GapExt, m_Params->m_MinFwdScore, &out, paths.data(),
                              paths.size()),
              "rsk_align_pairs");
        SetFromAln(out, paths.data() + out.path_off);
    }
}

}   // namespace reseek_amd

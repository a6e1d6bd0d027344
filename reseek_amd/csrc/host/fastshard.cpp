// fastshard.cpp -- `reseek -search Q -db DB -fast` with the DB cut into target shards, one per GPU.
// The prefilter's per-query top-B (RankedScoresBag, rankedscoresbag.cpp:34-51) is a reduction over ALL targets, so the
// sharded form has one exchange in the middle:
//   rsk_fast_shard_open        queries loaded; k-mer prefilter over this rank's target range; local top-B per query
//   rsk_fast_shard_candidates  -> the local (query, global target, score) triples            [caller: all_gather]
//   rsk_fast_shard_finish      merge of every rank's triples, top-B per query, then PostMuFilter (AlignBags, Accept, ToTsv)
//                              of the candidates whose target lies in this rank's range -> this rank's hit table
// Two forms of the exchange:
//   exact  (rsk_fast_shard_triples / rsk_fast_shard_finish_exact, what reseek_amd/dist.py and the in-process device list
//          use): every rank hands over ALL its (query, global target, score) triples; their union goes through the same
//          replay of RankedScoresBag as the single-GPU path (rsk_rsb_select: truncation at 2B, the reference's quicksort
//          tie order) -> candidates, hand-off file and hit table are the reference's for any shard count;
//   top-B  (rsk_fast_shard_candidates / rsk_fast_shard_finish): B triples per query and rank.  Which equal-scoring
//          candidates survive the reference's cut depends on every element its quicksort saw, so this form settles ties by a
//          rule of its own -- higher score first, then lower target index; the two agree whenever no query has more than B
//          candidates or the B-th score is not tied, and in every case the kept set is a valid top-B.
#include <algorithm>
#include <exception>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "reseek_host.h"

void rsk_set_error(const char *fmt, ...);

namespace reseek_amd {
void MuPreFilterScan(rsk_ctx *ctx, const std::vector<uint32_t> &qlen, const std::vector<uint8_t> &qmu, rsk_db *tdb, uint NT, int idx_mode,
                     std::vector<uint32_t> &hq, std::vector<uint32_t> &ht, std::vector<uint32_t> &hs);
void PostMuFilterPairs(const DSSParams &Params, DBSearcher &Q, DBSearcher &DB, const std::vector<uint32_t> &pq, const std::vector<uint32_t> &pt,
                       const std::string &HitsFN);
}
using namespace reseek_amd;

// per query the rsb_size best (score desc, target asc) of the given triples; output grouped by query, best first
static void TopB(const uint32_t *q, const uint32_t *t, const uint32_t *s, size_t n, uint32_t nqueries, uint32_t B, std::vector<uint32_t> &oq,
                 std::vector<uint32_t> &ot, std::vector<uint32_t> &os)
{
    std::vector<size_t> first((size_t) nqueries + 1, 0);
    for (size_t k = 0; k < n; ++k) {
        if (q[k] >= nqueries) throw std::runtime_error("top-B: query index out of range");
        ++first[q[k] + 1];
    }
    for (uint32_t i = 0; i < nqueries; ++i) first[i + 1] += first[i];
    std::vector<uint64_t> key(n);                          // (65535 - score) << 32 | target: ascending = best first
    {
        std::vector<size_t> cur(first.begin(), first.end() - 1);
        for (size_t k = 0; k < n; ++k) key[cur[q[k]]++] = ((uint64_t) (0xFFFFu - std::min<uint32_t>(s[k], 0xFFFFu)) << 32) | t[k];
    }
    oq.clear(); ot.clear(); os.clear();
    for (uint32_t i = 0; i < nqueries; ++i) {
        uint64_t *b = key.data() + first[i], *e = key.data() + first[i + 1];
        std::sort(b, e);
        e = std::unique(b, e);                             // the same (target, score) reported by two ranks cannot happen; harmless
        const size_t keep = std::min<size_t>((size_t) (e - b), B);
        for (size_t k = 0; k < keep; ++k) { oq.push_back(i); ot.push_back((uint32_t) b[k]); os.push_back(0xFFFFu - (uint32_t) (b[k] >> 32)); }
    }
}

extern "C" int rsk_rsb_merge(const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, uint32_t nqueries, uint32_t rsb_size,
                             uint32_t *out_q, uint32_t *out_t, uint32_t *out_score, size_t *nout)
{
    if ((n && (!q || !t || !score)) || !nout || rsb_size == 0) { rsk_set_error("rsk_rsb_merge: bad argument"); return RSK_E_INVALID; }
    try {
        std::vector<uint32_t> oq, ot, os;
        TopB(q, t, score, n, nqueries, rsb_size, oq, ot, os);
        *nout = oq.size();
        if (out_q)
            for (size_t k = 0; k < oq.size(); ++k) { out_q[k] = oq[k]; out_t[k] = ot[k]; out_score[k] = os[k]; }
    } catch (const std::exception &e) {
        rsk_set_error("rsk_rsb_merge: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

struct rsk_fast_shard {
    rsk_ctx *ctx = nullptr;
    SearchOptions o, o2;
    DSSParams Params, Params2;
    DBSearcher Q;
    std::string db_path;
    uint64_t Lo = 0, Hi = 0, NT = 0;                       // this rank's target range [Lo, Hi) of NT DB chains
    std::vector<uint32_t> cq, ct, cs;                      // local top-B triples (global target indexes)
    std::vector<uint32_t> aq, at, as;                      // every triple of the shard (the exact exchange)
};

namespace reseek_amd { int ParseSearchOpts(const rsk_search_opts *opts, SearchOptions &o, const char *who); }

// stage 1 of one shard (throws): S->o holds the options incl. shard_index / shard_count
static void FastShardOpen(rsk_fast_shard *S, rsk_ctx *ctx, const char *query_path, const char *db_path)
{
    {
        S->ctx = ctx;
        S->db_path = db_path;
        if (!(S->db_path.size() >= 4 && S->db_path.compare(S->db_path.size() - 4, 4, ".bca") == 0))
            throw std::runtime_error(".bca format required for -db (search.cpp:79)");
        S->Params.SetDSSParams(S->o);
        S->o2 = S->o;
        S->o2.mode = AM_Sensitive;                          // DM_AlwaysSensitive search.cpp:104
        S->Params2.SetDSSParams(S->o2);
        S->Q.m_Params = &S->Params2;
        S->Q.m_SelfRevQueryFlavour = true;                  // postmufilter.cpp:79
        S->Q.m_Opts = S->o;
        S->Q.m_Ctx = ctx;
        S->Q.LoadDB(query_path);
        S->Q.Setup();
        for (USERFIELD u : S->Q.m_DA.m_UFs)
            if (u == UF_Undefined) throw std::runtime_error("invalid -columns field");
        // target Mu letters of this rank's range, balanced by residues (the scan's cost is per target residue)
        std::vector<std::vector<byte> > TSeqs;
        if (!S->o.dbmu.empty()) {
            std::vector<std::string> Labels;
            std::vector<std::vector<byte> > All;
            ReadMuFasta(S->o.dbmu, Labels, All);
            S->NT = All.size();
            std::vector<uint32_t> Lens(S->NT);
            for (uint64_t i = 0; i < S->NT; ++i) Lens[i] = (uint32_t) All[i].size();
            DBSearcher::ResidueShardRange(Lens.data(), S->NT, S->o.shard_index, S->o.shard_count, S->Lo, S->Hi);
            TSeqs.assign(All.begin() + S->Lo, All.begin() + S->Hi);
        } else {
            BCAData B;
            B.Open(S->db_path);
            S->NT = B.GetChainCount();
            DBSearcher::ResidueShardRange(B.m_SeqLengths.data(), S->NT, S->o.shard_index, S->o.shard_count, S->Lo, S->Hi);
            MuSeqSource SS;
            SS.m_IsFasta = false;
            SS.m_Params = &S->Params;
            SS.m_ASCII = false;                             // muprefilter.cpp:84
            SS.m_CR.OpenRange(S->db_path, S->Lo, S->Hi);
            std::vector<std::string> Labels;
            SS.GetAll(Labels, TSeqs);
        }
        const uint NTl = (uint) TSeqs.size();
        if (NTl) {
            std::vector<uint32_t> tlen(NTl);
            std::vector<uint8_t> tmu;
            for (uint i = 0; i < NTl; ++i) { tlen[i] = (uint32_t) TSeqs[i].size(); tmu.insert(tmu.end(), TSeqs[i].begin(), TSeqs[i].end()); }
            rsk_db *tdb = nullptr;
            if (rsk_db_create(ctx, NTl, tlen.data(), tmu.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &tdb) != RSK_OK)
                throw std::runtime_error(std::string("rsk_db_create: ") + rsk_last_error());
            struct guard { rsk_db *d; ~guard() { rsk_db_destroy(d); } } g{ tdb };
            // query letters with 10 / 11 exchanged, as SeqDB::ToLetters(g_CharToLetterMu) leaves them (search.cpp:91-98)
            const uint NQ = S->Q.GetDBChainCount();
            std::vector<uint32_t> qlen(NQ);
            std::vector<uint8_t> qmu;
            for (uint i = 0; i < NQ; ++i) {
                qlen[i] = S->Q.m_DBChains[i]->GetSeqLength();
                for (byte l : *S->Q.m_DBMuLettersVec[i]) qmu.push_back(l == 10 ? 11 : (l == 11 ? 10 : l));
            }
            std::vector<uint32_t> hq, ht, hs;
            MuPreFilterScan(ctx, qlen, qmu, tdb, NTl, S->o.idx_mode, hq, ht, hs);
            for (uint32_t &t : ht) t += (uint32_t) S->Lo;
            S->aq.swap(hq); S->at.swap(ht); S->as.swap(hs);
        }
    }
}

extern "C" int rsk_fast_shard_open(rsk_ctx *ctx, const char *query_path, const char *db_path, const rsk_search_opts *opts, rsk_fast_shard **out)
{
    if (!ctx || !query_path || !db_path || !opts || !out) { rsk_set_error("rsk_fast_shard_open: NULL argument"); return RSK_E_INVALID; }
    *out = nullptr;
    std::unique_ptr<rsk_fast_shard> S(new rsk_fast_shard);
    { const int rc = reseek_amd::ParseSearchOpts(opts, S->o, "rsk_fast_shard_open"); if (rc != RSK_OK) return rc; }
    if (S->o.mode != AM_Fast) { rsk_set_error("rsk_fast_shard_open: mode must be \"fast\" (the other modes shard through rsk_search)"); return RSK_E_INVALID; }
    if (!S->o.shard_count) S->o.shard_count = 1;
    if (S->o.shard_index >= S->o.shard_count) { rsk_set_error("rsk_fast_shard_open: shard_index >= shard_count"); return RSK_E_INVALID; }
    try {
        FastShardOpen(S.get(), ctx, query_path, db_path);
    } catch (const std::exception &e) {
        rsk_set_error("rsk_fast_shard_open: %s", e.what());
        return RSK_E_INVALID;
    }
    *out = S.release();
    return RSK_OK;
}

extern "C" int rsk_fast_shard_candidates(rsk_fast_shard *S, const uint32_t **q, const uint32_t **t, const uint32_t **score, size_t *n)
{
    if (!S || !q || !t || !score || !n) { rsk_set_error("rsk_fast_shard_candidates: NULL argument"); return RSK_E_INVALID; }
    try {
        if (S->cq.empty() && !S->aq.empty())
            TopB(S->aq.data(), S->at.data(), S->as.data(), S->aq.size(), S->Q.GetDBChainCount(), S->o.rsb_size, S->cq, S->ct, S->cs);
    } catch (const std::exception &e) {
        rsk_set_error("rsk_fast_shard_candidates: %s", e.what());
        return RSK_E_INVALID;
    }
    *q = S->cq.data(); *t = S->ct.data(); *score = S->cs.data(); *n = S->cq.size();
    return RSK_OK;
}

extern "C" int rsk_fast_shard_triples(rsk_fast_shard *S, const uint32_t **q, const uint32_t **t, const uint32_t **score, size_t *n)
{
    if (!S || !q || !t || !score || !n) { rsk_set_error("rsk_fast_shard_triples: NULL argument"); return RSK_E_INVALID; }
    *q = S->aq.data(); *t = S->at.data(); *score = S->as.data(); *n = S->aq.size();
    return RSK_OK;
}

// stage 2 of one shard (throws)
static void FastShardFinish(rsk_fast_shard *S, const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, const char *out_tsv,
                            const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8)
{
    {
        const uint NQ = S->Q.GetDBChainCount();
        std::vector<uint32_t> mq, mt, ms;
        TopB(q, t, score, n, NQ, S->o.rsb_size, mq, mt, ms);
        if (tmp_tsv && *tmp_tsv) {
            // the hand-off file of the merged bags (rankedscoresbag.cpp:185-231), for comparison with a single-GPU run
            RankedScoresBag RSB;
            RSB.m_B = S->o.rsb_size;
            RSB.Init(NQ);
            for (size_t k = 0; k < mq.size(); ++k) { RSB.m_QueryIdxToScoreVec[mq[k]].push_back((uint16_t) ms[k]); RSB.m_QueryIdxToTargetIdxVec[mq[k]].push_back(mt[k]); }
            FILE *f = fopen(tmp_tsv, "w");
            if (!f) throw std::runtime_error(std::string("cannot create ") + tmp_tsv);
            RSB.m_B = UINT32_MAX;                           // already truncated: ToTsv must not re-sort
            RSB.ToTsv(f);
            fclose(f);
        }
        // candidates of this rank's targets, target-major as PostMuFilter reads them from the hand-off file
        std::vector<std::pair<uint32_t, uint32_t> > mine;  // (target, query)
        for (size_t k = 0; k < mq.size(); ++k)
            if (mt[k] >= S->Lo && mt[k] < S->Hi) mine.emplace_back(mt[k], mq[k]);
        std::sort(mine.begin(), mine.end());
        std::vector<uint32_t> uniq;
        for (auto &p : mine)
            if (uniq.empty() || uniq.back() != p.first) uniq.push_back(p.first);
        FILE *f = fopen(out_tsv, "w");                      // a rank without candidates still leaves an (empty) table
        if (!f) throw std::runtime_error(std::string("cannot create ") + out_tsv);
        fclose(f);
        if (!mine.empty()) {
            BCAData B;
            B.Open(S->db_path);
            std::vector<PDBChain *> Chains;
            for (uint32_t ti : uniq) {
                PDBChain *C = new PDBChain;
                B.ReadChain(ti, *C);
                Chains.push_back(C);
            }
            DBSearcher DB;
            DB.m_Params = &S->Params2;
            DB.m_SelfRevQueryFlavour = true;                // postmufilter.cpp:171
            DB.m_Opts = S->o;
            DB.m_Ctx = S->ctx;
            DB.LoadChains(Chains);
            if (DB.GetDBChainCount() != uniq.size()) throw std::runtime_error("empty chain among the candidates");
            std::vector<uint32_t> pq, pt;
            for (auto &p : mine) {
                pq.push_back(p.second);
                pt.push_back((uint32_t) (std::lower_bound(uniq.begin(), uniq.end(), p.first) - uniq.begin()));
            }
            PostMuFilterPairs(S->Params2, S->Q, DB, pq, pt, out_tsv);
        }
        if (nhits) *nhits = S->Q.m_HitCount;
        if (stats8) {
            stats8[0] = S->Q.m_ProcessedPairCount; stats8[1] = S->Q.m_ProcessedPairCount - S->Q.m_MKFPairCount; stats8[2] = S->Q.m_MuFilterInputCount;
            stats8[3] = S->Q.m_MuFilterDiscardCount; stats8[4] = S->Q.m_MKFPairCount; stats8[5] = S->Q.m_SWCount;
            stats8[6] = S->Q.m_HitCount; stats8[7] = 1;
        }
    }
}

extern "C" int rsk_fast_shard_finish(rsk_fast_shard *S, const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n,
                                     const char *out_tsv, const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8)
{
    if (!S || !out_tsv || (n && (!q || !t || !score))) { rsk_set_error("rsk_fast_shard_finish: NULL argument"); return RSK_E_INVALID; }
    try {
        FastShardFinish(S, q, t, score, n, out_tsv, tmp_tsv, nhits, stats8);
    } catch (const std::exception &e) {
        rsk_set_error("rsk_fast_shard_finish: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

// The exact exchange: the union of every rank's triples -> the reference's bags (rsk_rsb_select) -> stage 2 of this shard.
static void FastShardFinishExact(rsk_fast_shard *S, const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, const char *out_tsv,
                                 const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8)
{
    std::vector<uint32_t> sq(n), st(n), ss(n);
    size_t nout = 0;
    if (rsk_rsb_select(q, t, score, n, S->Q.GetDBChainCount(), S->o.rsb_size, sq.data(), st.data(), ss.data(), &nout, tmp_tsv && *tmp_tsv ? tmp_tsv : nullptr) !=
        RSK_OK)
        throw std::runtime_error(std::string("rsk_rsb_select: ") + rsk_last_error());
    FastShardFinish(S, sq.data(), st.data(), ss.data(), nout, out_tsv, nullptr, nhits, stats8);
}

extern "C" int rsk_fast_shard_finish_exact(rsk_fast_shard *S, const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n,
                                           const char *out_tsv, const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8)
{
    if (!S || !out_tsv || (n && (!q || !t || !score))) { rsk_set_error("rsk_fast_shard_finish_exact: NULL argument"); return RSK_E_INVALID; }
    try {
        FastShardFinishExact(S, q, t, score, n, out_tsv, tmp_tsv, nhits, stats8);
    } catch (const std::exception &e) {
        rsk_set_error("rsk_fast_shard_finish_exact: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

// `-search -fast -db` on several devices of ONE process: one target shard per context, each stage on a host thread per
// shard, the exact exchange (top of the file) as a concatenation in host memory.  The shards' hit tables are appended to
// out_tsv in shard order.
namespace reseek_amd {
void FastDbOnContexts(const std::vector<rsk_ctx *> &Ctx, const char *query_path, const char *db_path, const SearchOptions &o, const char *out_tsv,
                      const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8)
{
    const uint D = (uint) Ctx.size();
    std::vector<std::unique_ptr<rsk_fast_shard> > Sh(D);
    std::vector<std::exception_ptr> Err(D);
    auto on_all = [&](const std::function<void(uint)> &fn) {
        std::vector<std::thread> Th;
        for (uint k = 0; k < D; ++k)
            Th.emplace_back([&, k]() { try { fn(k); } catch (...) { Err[k] = std::current_exception(); } });
        for (auto &t : Th) t.join();
        for (uint k = 0; k < D; ++k)
            if (Err[k]) std::rethrow_exception(Err[k]);
    };
    on_all([&](uint k) {
        Sh[k].reset(new rsk_fast_shard);
        Sh[k]->o = o;
        Sh[k]->o.mode = AM_Fast;
        Sh[k]->o.shard_index = k;
        Sh[k]->o.shard_count = D;
        FastShardOpen(Sh[k].get(), Ctx[k], query_path, db_path);
    });
    std::vector<uint32_t> aq, at, as;
    {
        std::vector<uint32_t> uq, ut, us;
        for (uint k = 0; k < D; ++k) {
            uq.insert(uq.end(), Sh[k]->aq.begin(), Sh[k]->aq.end());
            ut.insert(ut.end(), Sh[k]->at.begin(), Sh[k]->at.end());
            us.insert(us.end(), Sh[k]->as.begin(), Sh[k]->as.end());
            std::vector<uint32_t>().swap(Sh[k]->aq); std::vector<uint32_t>().swap(Sh[k]->at); std::vector<uint32_t>().swap(Sh[k]->as);
        }
        aq.resize(uq.size()); at.resize(uq.size()); as.resize(uq.size());
        size_t nout = 0;
        if (rsk_rsb_select(uq.data(), ut.data(), us.data(), uq.size(), Sh[0]->Q.GetDBChainCount(), o.rsb_size, aq.data(), at.data(), as.data(), &nout,
                           tmp_tsv && *tmp_tsv ? tmp_tsv : nullptr) != RSK_OK)
            throw std::runtime_error(std::string("rsk_rsb_select: ") + rsk_last_error());
        aq.resize(nout); at.resize(nout); as.resize(nout);
    }
    std::vector<uint64_t> Hits(D, 0);
    std::vector<std::vector<uint64_t> > Stats(D, std::vector<uint64_t>(8, 0));
    // the shards' part files are removed on EVERY exit path (a shard that throws must not leave the others' files behind)
    struct PartFiles {
        std::vector<std::string> names;
        ~PartFiles() { for (const std::string &n : names) remove(n.c_str()); }
    } Parts;
    for (uint k = 0; k < D; ++k) Parts.names.push_back(std::string(out_tsv) + ".shard" + std::to_string(k));
    on_all([&](uint k) {
        FastShardFinish(Sh[k].get(), aq.data(), at.data(), as.data(), aq.size(), Parts.names[k].c_str(), nullptr, &Hits[k], Stats[k].data());
    });
    FILE *f = fopen(out_tsv, "w");
    if (!f) throw std::runtime_error(std::string("cannot create ") + out_tsv);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{ f };
    std::vector<char> buf(1 << 20);
    for (uint k = 0; k < D; ++k) {
        FILE *g = fopen(Parts.names[k].c_str(), "r");
        if (!g) {
            // a shard without candidates writes no hits file (postmufilter.cpp:219-223: "No hits found"); anything else is an error
            if (Hits[k] == 0) continue;
            throw std::runtime_error("the hits of shard " + std::to_string(k) + " are missing (" + Parts.names[k] + ")");
        }
        size_t got;
        while ((got = fread(buf.data(), 1, buf.size(), g)) > 0)
            if (fwrite(buf.data(), 1, got, f) != got) { fclose(g); throw std::runtime_error("short write to the hits file"); }
        fclose(g);
    }
    closer.f = nullptr;
    if (fclose(f) != 0) throw std::runtime_error("short write to the hits file");
    if (nhits) { *nhits = 0; for (uint k = 0; k < D; ++k) *nhits += Hits[k]; }
    if (stats8) {
        for (int c = 0; c < 7; ++c) { stats8[c] = 0; for (uint k = 0; k < D; ++k) stats8[c] += Stats[k][c]; }
        stats8[7] = 1;
    }
}
}   // namespace reseek_amd

extern "C" void rsk_fast_shard_close(rsk_fast_shard *S) { delete S; }

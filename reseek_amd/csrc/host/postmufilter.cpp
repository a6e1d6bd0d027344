// postmufilter.cpp -- host mirror of the two stages of `reseek -search Q -db DB -fast`
// (cmd_search search.cpp:62-111):
//   MuPreFilter  muprefilter.cpp:70-133   Mu k-mer prefilter of every DB chain against the query index,
//                                         bounded top-B per query, hand-off TSV (rankedscoresbag.cpp:185)
//   PostMuFilter postmufilter.cpp:190-290 for each target line of that file: DSSAligner::AlignBags
//                                         (chainbag.cpp:44) of every listed query against the target,
//                                         Accept (:106) and ToTsv(up = true).
// The reference walks one (query, target) candidate at a time; here the candidate list is cut into
// the same three streams as AlignBags' decision tree and each stream is one GPU batch call:
//   DoMKF_Bags -> host MKF path;  Omega > 0 -> rsk_mu_filter_pairs;  survivors -> rsk_align_pairs.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <memory>
#include <atomic>
#include <cstring>
#include <future>
#include <chrono>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "reseek_host.h"
#include "../rsk_internal.h"

namespace reseek_amd {

static void check(int rc, const char *what)
{
    if (rc != RSK_OK) throw std::runtime_error(std::string(what) + ": " + rsk_last_error());
}

void MuPreFilterLetters(rsk_ctx *ctx, const std::vector<uint32_t> &qlen, const std::vector<uint8_t> &qmu, rsk_db *tdb, uint NT, int idx_mode,
                        uint rsb_size, const std::string &OutputFN);

int ReplaySortedKeys(RankedScoresBag &RSB, const uint64_t *keys, size_t n, uint32_t nqueries, uint32_t rsb_size);     // prefilter.cpp
void MuPreFilterBags(rsk_ctx *ctx, const std::vector<uint32_t> &qlen, const std::vector<uint8_t> &qmu, rsk_db *tdb, uint NT, int idx_mode,
                     uint rsb_size, RankedScoresBag &RSB);

// MuPreFilter (muprefilter.cpp:70) up to the filled bags; OutputFN (may be empty) receives the hand-off file, pq / pt (may
// be NULL) the candidate pairs in the file's order -- the single-process search hands them to PostMuFilterPairs in memory
// and writes the file only for -keeptmp.
static void MuPreFilterImpl(DBSearcher &QDB, DBSearcher &TDB, const std::string &OutputFN, std::vector<uint32_t> *pq, std::vector<uint32_t> *pt)
{
    rsk_ctx *ctx = QDB.m_Ctx;
    if (!ctx) throw std::runtime_error("MuPreFilter: no GPU context");
    TDB.m_Ctx = ctx;
    const uint NQ = QDB.GetDBChainCount();
    // target letters: the DB chains' own Mu letters (MuSeqSource::OpenChains, m_ASCII = false) or, with -dbmu,
    // a Mu FASTA read through g_CharToLetterMu (MuSeqSource::OpenFasta museqsource.cpp:21-30, search.cpp:93-96)
    rsk_db *tdb = nullptr;
    struct tdb_guard { rsk_db *d = nullptr; ~tdb_guard() { if (d) rsk_db_destroy(d); } } tguard;
    uint NT;
    if (!QDB.m_Opts.dbmu.empty()) {
        std::vector<std::string> TLabels;
        std::vector<std::vector<byte> > TSeqs;
        ReadMuFasta(QDB.m_Opts.dbmu, TLabels, TSeqs);
        NT = (uint) TSeqs.size();
        std::vector<uint32_t> tlen(NT);
        std::vector<uint8_t> tmu;
        for (uint i = 0; i < NT; ++i) { tlen[i] = (uint32_t) TSeqs[i].size(); tmu.insert(tmu.end(), TSeqs[i].begin(), TSeqs[i].end()); }
        check(rsk_db_create(ctx, NT, tlen.data(), tmu.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &tdb), "rsk_db_create");
        tguard.d = tdb;
    } else {
        TDB.UploadToGpu();
        tdb = TDB.m_Db;
        NT = TDB.GetDBChainCount();
    }
    // Query letters as cmd_search hands them over (search.cpp:91-98): MuSeqSource writes the query chains
    // as text with 'A' + letter (museqsource.cpp:45-53, pdbchain.cpp:70), SeqDB::ToLetters reads the text
    // back through g_CharToLetterMu, whose table has L = 10 and K = 11 (alpha.cpp:3291) -- so the QUERY side
    // of the prefilter sees letters 10 and 11 exchanged, the target side (m_ASCII = false) does not.
    // Reference behaviour, kept: the hand-off file is compared byte for byte.
    std::vector<uint32_t> qlen(NQ);
    size_t qtot = 0;
    for (uint i = 0; i < NQ; ++i) { qlen[i] = QDB.m_DBChains[i]->GetSeqLength(); qtot += qlen[i]; }
    std::vector<uint8_t> qmu(qtot);
    size_t qo = 0;
    for (uint i = 0; i < NQ; ++i)
        for (byte l : *QDB.m_DBMuLettersVec[i]) qmu[qo++] = l == 10 ? 11 : (l == 11 ? 10 : l);
    RankedScoresBag RSB;
    MuPreFilterBags(ctx, qlen, qmu, tdb, NT, QDB.m_Opts.idx_mode, QDB.m_Opts.rsb_size, RSB);
    PhaseTimer tm("MuPreFilter");
    if (!OutputFN.empty()) {
        FILE *f = fopen(OutputFN.c_str(), "w");
        if (!f) throw std::runtime_error("MuPreFilter: cannot create " + OutputFN);
        RSB.ToTsv(f);
        fclose(f);
        tm.lap("hand-off file");
    }
    if (pq && pt) { RSB.ToPairs(*pq, *pt); tm.lap("candidate pairs (memory)"); }
}

void MuPreFilter(const DSSParams &Params, DBSearcher &QDB, DBSearcher &TDB, const std::string &OutputFN)
{
    (void) Params;
    MuPreFilterImpl(QDB, TDB, OutputFN, nullptr, nullptr);
}

void MuPreFilterToPairs(DBSearcher &QDB, DBSearcher &TDB, std::vector<uint32_t> &pq, std::vector<uint32_t> &pt, const std::string &KeepTmpFN)
{
    MuPreFilterImpl(QDB, TDB, KeepTmpFN, &pq, &pt);
}

// index of the query letters + scan of every target (muprefilter.cpp:90-126): (query, target, score) triples, unordered
void MuPreFilterScan(rsk_ctx *ctx, const std::vector<uint32_t> &qlen, const std::vector<uint8_t> &qmu, rsk_db *tdb, uint NT, int idx_mode,
                     std::vector<uint32_t> &hq, std::vector<uint32_t> &ht, std::vector<uint32_t> &hs)
{
    const uint NQ = (uint) qlen.size();
    PhaseTimer tm("MuPreFilter");
    rsk_db *qdb = nullptr;
    check(rsk_db_create(ctx, NQ, qlen.data(), qmu.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &qdb), "rsk_db_create");
    struct db_guard { rsk_db *d; ~db_guard() { rsk_db_destroy(d); } } guard{ qdb };
    auto hipok = [](hipError_t e, const char *w) { if (e != hipSuccess) throw std::runtime_error(std::string(w) + ": " + hipGetErrorString(e)); };
    // every (query, target) can appear at most once
    size_t cap = (size_t) std::min<uint64_t>((uint64_t) NQ * NT, 1ull << 28);
    hq.clear(); ht.clear(); hs.clear();
    DeviceBuffer Count(ctx, 4, "prefilter result counter");
    for (;;) {
        DeviceBuffer Q(ctx, cap * 4, "prefilter results"), T(ctx, cap * 4, "prefilter results"), S(ctx, cap * 4, "prefilter results");
        check(rsk_mu_prefilter_dev(ctx, qdb, tdb, idx_mode, Q.As<uint32_t>(), T.As<uint32_t>(), S.As<uint32_t>(), cap, Count.As<uint32_t>()), "rsk_mu_prefilter_dev");
        check(rsk_ctx_sync(ctx), "rsk_ctx_sync");
        uint32_t n = 0;
        hipok(hipMemcpy(&n, Count.As<uint32_t>(), 4, hipMemcpyDeviceToHost), "copy n");
        if (n <= cap) {
            hq.resize(n); ht.resize(n); hs.resize(n);
            hipok(hipMemcpy(hq.data(), Q.As<uint32_t>(), (size_t) n * 4, hipMemcpyDeviceToHost), "copy");
            hipok(hipMemcpy(ht.data(), T.As<uint32_t>(), (size_t) n * 4, hipMemcpyDeviceToHost), "copy");
            hipok(hipMemcpy(hs.data(), S.As<uint32_t>(), (size_t) n * 4, hipMemcpyDeviceToHost), "copy");
            break;
        }
        cap = n;                               // the count is exact even when the list was truncated
    }
    tm.lap("index + scan (GPU)");
}

// scan + bags (muprefilter.cpp:90-133).  The DB is scanned in a few contiguous target ranges: the triples of a range are
// sorted on the device (query, then target = the arrival order of the reference's bags with -threads 1) and downloaded as
// packed keys, and while the device scans range k + 1 a host task replays range k into the bags (the ranges arrive in target
// order, a query's bag sees its targets ascending over the whole DB; the final truncation follows the last range).
int ReplayAppendSortedKeys(RankedScoresBag &RSB, const uint64_t *keys, size_t n, uint32_t nqueries, bool Final);      // prefilter.cpp
void MuPreFilterBags(rsk_ctx *ctx, const std::vector<uint32_t> &qlen, const std::vector<uint8_t> &qmu, rsk_db *tdb, uint NT, int idx_mode,
                     uint rsb_size, RankedScoresBag &RSB)
{
    const uint NQ = (uint) qlen.size();
    PhaseTimer tm("MuPreFilter");
    rsk_db *qdb = nullptr;
    check(rsk_db_create(ctx, NQ, qlen.data(), qmu.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &qdb), "rsk_db_create");
    struct db_guard { rsk_db *d; ~db_guard() { rsk_db_destroy(d); } } guard{ qdb };
    auto hipok = [](hipError_t e, const char *w) { if (e != hipSuccess) throw std::runtime_error(std::string(w) + ": " + hipGetErrorString(e)); };
    RSB.m_B = rsb_size;
    RSB.Init(NQ);
    // a few target ranges per scan (RSK_PF_RANGES; 1 = one scan).  Measured on the 11,211 x 11,211 synthetic set: 1 / 2 / 3 / 4 / 6
    // ranges -> scans + bags 1.25 / 1.07 / 1.06 / 1.10 / 1.29 s (every launch ends with a tail of long targets): three.
    uint nranges = (uint) std::min<uint64_t>(3, std::max<uint64_t>(1, (uint64_t) NQ * NT / (16u << 20)));
    if (const char *e = getenv("RSK_PF_RANGES")) nranges = (uint) std::max(1, atoi(e));
    // every (query, target) pair appears at most once: ranges of fewer than 2^31 pairs keep the 32-bit triple counter of a
    // scan (and its result buffers, ~16 bytes per triple) in range whatever the query set and the DB
    nranges = (uint) std::max<uint64_t>(nranges, ((uint64_t) NQ * NT >> 31) + 1);
    nranges = std::max(1u, std::min(nranges, std::max(1u, NT)));
    DeviceBuffer Count(ctx, 4, "prefilter result counter");
    std::future<void> replay;                                            // the host task of the previous range
    struct Join { std::future<void> &f; ~Join() { if (f.valid()) f.wait(); } } join_on_exit{ replay };
    double t_scan = 0, t_keys = 0;
    for (uint r = 0; r < nranges; ++r) {
        const uint t_lo = (uint) ((uint64_t) NT * r / nranges), t_hi = (uint) ((uint64_t) NT * (r + 1) / nranges);
        if (t_hi == t_lo) continue;
        const auto c0 = std::chrono::steady_clock::now();
        size_t cap = (size_t) std::min<uint64_t>((uint64_t) NQ * (t_hi - t_lo), 1ull << 28);      // every (query, target) can appear at most once
        std::unique_ptr<uint64_t[]> keys;                                // ~1 GB per scan on a dense set: not value-initialised
        uint32_t n = 0;
        for (;;) {
            DeviceBuffer Q(ctx, cap * 4, "prefilter results"), T(ctx, cap * 4, "prefilter results"), S(ctx, cap * 4, "prefilter results");
            check(rsk_mu_prefilter_range_dev(ctx, qdb, tdb, idx_mode, t_lo, t_hi, Q.As<uint32_t>(), T.As<uint32_t>(), S.As<uint32_t>(), cap, Count.As<uint32_t>()),
                  "rsk_mu_prefilter_dev");
            check(rsk_ctx_sync(ctx), "rsk_ctx_sync");
            hipok(hipMemcpy(&n, Count.As<uint32_t>(), 4, hipMemcpyDeviceToHost), "copy n");
            if (n > cap) { cap = n; continue; }                          // the count is exact even when the list was truncated
            const auto c1 = std::chrono::steady_clock::now();
            t_scan += std::chrono::duration<double, std::milli>(c1 - c0).count();
            DeviceBuffer K(ctx, (size_t) std::max<uint32_t>(n, 1) * 8, "prefilter keys");
            check(rsk_triples_sort_dev(ctx, Q.As<uint32_t>(), T.As<uint32_t>(), S.As<uint32_t>(), n, K.As<uint64_t>()), "rsk_triples_sort_dev");
            keys.reset(new uint64_t[std::max<uint32_t>(n, 1)]);
            hipok(hipMemcpy(keys.get(), K.As<uint64_t>(), (size_t) n * 8, hipMemcpyDeviceToHost), "copy keys");
            t_keys += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c1).count();
            break;
        }
        if (replay.valid()) replay.get();                                // range r - 1 is in the bags (rethrows its error)
        const bool Final = r + 1 == nranges;
        std::shared_ptr<uint64_t[]> held(keys.release());
        replay = std::async(std::launch::async, [&RSB, held, n, NQ, Final]() {
            if (ReplayAppendSortedKeys(RSB, held.get(), n, NQ, Final) != RSK_OK) throw std::runtime_error(std::string("MuPreFilter: ") + rsk_last_error());
        });
    }
    if (getenv("RSK_TRACE")) fprintf(stderr, "[MuPreFilter] %u target ranges: index + scans %.1f ms, triples sorted on the device + downloaded %.1f ms\n", nranges, t_scan, t_keys);
    tm.lap("index + scans (GPU), bags of the earlier ranges beside them");
    if (replay.valid()) replay.get();
    else RSB.Finish();                                                   // (no target at all)
    tm.lap("top-B bags of the last range");
}

// ... + hand-off file (muprefilter.cpp:127-133): the form compat.cpp's MuPreFilter(SeqDB &, MuSeqSource &) forwards to
void MuPreFilterLetters(rsk_ctx *ctx, const std::vector<uint32_t> &qlen, const std::vector<uint8_t> &qmu, rsk_db *tdb, uint NT, int idx_mode,
                        uint rsb_size, const std::string &OutputFN)
{
    RankedScoresBag RSB;
    MuPreFilterBags(ctx, qlen, qmu, tdb, NT, idx_mode, rsb_size, RSB);
    PhaseTimer tm("MuPreFilter");
    FILE *f = fopen(OutputFN.c_str(), "w");
    if (!f) throw std::runtime_error("MuPreFilter: cannot create " + OutputFN);
    RSB.ToTsv(f);
    fclose(f);
    tm.lap("hand-off file");
}

static bool Accept(const DSSAligner &DA, double MaxEvalue, double MaxPvalue, double MinTS)     // postmufilter.cpp:106-115
{
    if (DA.m_EvalueA <= MaxEvalue) return true;
    if (DA.m_PvalueA <= MaxPvalue) return true;
    if (DA.m_NewTestStatisticA >= MinTS) return true;
    return false;
}

// The hand-off file of the prefilter (rankedscoresbag.cpp:185-231): "prefilter\t<#targets>", then "TIdx\tK\tQIdx..." lines ->
// candidate pairs in file order (A = query, B = target).  NoHits = the header announces no target line (postmufilter.cpp:219-223).
void ReadHandOff(const std::string &MuFilterTsvFN, uint NQ, uint64_t NT, std::vector<uint32_t> &pq, std::vector<uint32_t> &pt, bool &NoHits)
{
    NoHits = false;
    pq.clear(); pt.clear();
    FILE *fin = fopen(MuFilterTsvFN.c_str(), "r");
    if (!fin) throw std::runtime_error("PostMuFilter: cannot open " + MuFilterTsvFN);
    std::vector<char> buf(1 << 16);
    std::string line;
    auto readline = [&]() -> bool {
        line.clear();
        for (;;) {
            if (!fgets(buf.data(), (int) buf.size(), fin)) return !line.empty();
            line += buf.data();
            if (!line.empty() && line.back() == '\n') { line.pop_back(); return true; }
        }
    };
    if (!readline()) { fclose(fin); throw std::runtime_error("PostMuFilter: empty hand-off file"); }
    unsigned LineCount = 0;
    if (sscanf(line.c_str(), "prefilter\t%u", &LineCount) != 1) { fclose(fin); throw std::runtime_error("PostMuFilter: bad header line"); }
    if (LineCount == 0) { fclose(fin); fprintf(stderr, "Warning: No hits found by mufilter pass\n"); NoHits = true; return; }   // :219-223 (no hits file)
    for (unsigned k = 0; k < LineCount; ++k) {
        if (!readline()) { fclose(fin); throw std::runtime_error("PostMuFilter: truncated hand-off file"); }
        const char *s = line.c_str();
        char *e;
        const unsigned long T = strtoul(s, &e, 10);
        const unsigned long K = strtoul(e, &e, 10);
        if (T >= NT) { fclose(fin); throw std::runtime_error("PostMuFilter: target index out of range"); }
        for (unsigned long c = 0; c < K; ++c) {
            if (*e != '\t') { fclose(fin); throw std::runtime_error("PostMuFilter: short target line"); }
            const unsigned long qi = strtoul(e, &e, 10);
            if (qi >= NQ) { fclose(fin); throw std::runtime_error("PostMuFilter: query index out of range"); }
            pq.push_back((uint32_t) qi); pt.push_back((uint32_t) T);
        }
    }
    fclose(fin);
}

void PostMuFilterPairs(const DSSParams &Params, DBSearcher &Q, DBSearcher &DB, const std::vector<uint32_t> &pq, const std::vector<uint32_t> &pt,
                       const std::string &HitsFN);

void PostMuFilter(const DSSParams &Params, const std::string &MuFilterTsvFN, DBSearcher &Q, DBSearcher &DB, const std::string &HitsFN)
{
    PhaseTimer tm("PostMuFilter");
    std::vector<uint32_t> pq, pt;
    bool NoHits = false;
    ReadHandOff(MuFilterTsvFN, Q.GetDBChainCount(), DB.GetDBChainCount(), pq, pt, NoHits);
    tm.lap("read hand-off file");
    if (NoHits) return;
    PostMuFilterPairs(Params, Q, DB, pq, pt, HitsFN);
}

// AlignBags (chainbag.cpp:44) of every candidate, Accept (postmufilter.cpp:106) and ToTsv(up = true)
void PostMuFilterPairs(const DSSParams &Params, DBSearcher &Q, DBSearcher &DB, const std::vector<uint32_t> &pq, const std::vector<uint32_t> &pt,
                       const std::string &HitsFN)
{
    PhaseTimer tm("PostMuFilter");
    const SearchOptions &O = Q.m_Opts;
    double MaxEvalue = 10, MaxPvalue = -1, MinTS = 9e9;                    // postmufilter.cpp:31-33,196-203
    if (O.evalue_set) MaxEvalue = O.evalue;
    else if (O.mode == AM_VerySensitive) MaxEvalue = 9e9;
    if (O.pvalue_set) MaxPvalue = O.pvalue;
    if (O.mints_set) MinTS = O.mints;

    rsk_ctx *ctx = Q.m_Ctx;
    if (!ctx) throw std::runtime_error("PostMuFilter: no GPU context");
    DB.m_Ctx = ctx;
    Q.UploadToGpu();
    DB.UploadToGpu();
    tm.lap("upload both sets");
    FILE *fTsv = fopen(HitsFN.c_str(), "w");
    if (!fTsv) throw std::runtime_error("PostMuFilter: cannot create " + HitsFN);
    DSSAligner &DA = Q.m_DA;
    DA.SetParams(Params);
    DA.SetColumns(O.columns);
    DA.m_Ctx = ctx;

    // DoMKF_Bags chainbag.cpp:6-21: long-chain pairs leave the list (slices of the candidate list on the host threads,
    // concatenated in order)
    const uint NQc = Q.GetDBChainCount(), NTc = DB.GetDBChainCount();
    std::vector<uint32_t> lenQ(NQc), lenT(NTc);
    for (uint i = 0; i < NQc; ++i) lenQ[i] = Q.m_DBChains[i]->GetSeqLength();
    for (uint j = 0; j < NTc; ++j) lenT[j] = DB.m_DBChains[j]->GetSeqLength();
    std::vector<uint32_t> fq, ft;
    std::vector<std::pair<uint32_t, uint32_t> > mkf;
    {
        const size_t np = pq.size(), nsl = std::max<size_t>(1, std::min<size_t>(64, np / 65536 + 1));
        std::vector<std::vector<uint32_t> > sq(nsl), st(nsl);
        std::vector<std::vector<std::pair<uint32_t, uint32_t> > > sm(nsl);
        rsk_parallel_for(nsl, 1, [&](size_t lo, size_t hi) {
            for (size_t sl = lo; sl < hi; ++sl)
                for (size_t p = np * sl / nsl, e = np * (sl + 1) / nsl; p < e; ++p) {
                    if (lenQ[pq[p]] >= Params.m_MKFL || lenT[pt[p]] >= Params.m_MKFL) sm[sl].emplace_back(pq[p], pt[p]);
                    else { sq[sl].push_back(pq[p]); st[sl].push_back(pt[p]); }
                }
        });
        for (size_t sl = 0; sl < nsl; ++sl) {
            fq.insert(fq.end(), sq[sl].begin(), sq[sl].end()); ft.insert(ft.end(), st[sl].begin(), st[sl].end());
            mkf.insert(mkf.end(), sm[sl].begin(), sm[sl].end());
        }
    }
    Q.m_MKFPairCount = mkf.size();
    Q.m_ProcessedPairCount = pq.size();
    // Mu filter (chainbag.cpp:67-73)
    std::vector<uint32_t> ia, ib;
    if (Params.m_Omega > 0 && !fq.empty()) {
        std::vector<uint8_t> pass(fq.size());
        check(rsk_mu_filter_pairs(ctx, Q.m_Db, DB.m_Db, fq.data(), ft.data(), fq.size(), Params.m_ParaMuGapOpen, Params.m_ParaMuGapExt,
                                  Params.m_Omega, Params.m_OmegaFwd, pass.data(), nullptr, nullptr),
              "rsk_mu_filter_pairs");
        ia.reserve(fq.size() / 2); ib.reserve(fq.size() / 2);
        for (size_t p = 0; p < fq.size(); ++p)
            if (pass[p]) { ia.push_back(fq[p]); ib.push_back(ft[p]); }
        Q.m_MuFilterInputCount = fq.size();
        Q.m_MuFilterDiscardCount = fq.size() - ia.size();
    } else { ia.swap(fq); ib.swap(ft); }
    tm.lap("Mu filter (pair list)");
    // SetSMx_NoRev + SWFast + CalcEvalue (chainbag.cpp:74-84) in batches; beside it, on a helper context, the long chains:
    // MKF (chainbag.cpp:58-65)
    std::sort(mkf.begin(), mkf.end());
    auto align_job = [&]() {
        ForEachAlignedBatch(Params, ctx, O, Q, DB, ia, ib,
                            [&](const std::vector<uint32_t> &bia, const std::vector<uint32_t> &bib, const std::vector<rsk_aln> &out, const char *paths) {
            const size_t n = bia.size();
            Q.m_SWCount += n;
            {
                uint64_t scored = 0;                       // rsk_path_counters: CalcEvalue ran (score >= m_MinFwdScore)
                for (size_t p = 0; p < n; ++p) scored += out[p].evalue != FLT_MAX;
                g_rsk_counters.sw_pairs += n;
                g_rsk_counters.sw_pairs_scored += scored;
            }
            for (size_t p = 0; p < n; ++p) {
                // Accept (postmufilter.cpp:106-115) on the batch record: rejected pairs need no string work
                if (!(out[p].evalue <= MaxEvalue || out[p].pvalue <= MaxPvalue || (out[p].evalue != FLT_MAX && out[p].ts >= MinTS))) continue;
                const uint i = bia[p], j = bib[p];
                DA.m_ChainA = Q.m_DBChains[i]; DA.m_ProfileA = Q.m_DBProfiles[i];
                DA.m_ChainB = DB.m_DBChains[j]; DA.m_ProfileB = DB.m_DBProfiles[j];
                DA.m_SelfRevScoreA = Q.m_DBSelfRevScores[i]; DA.m_SelfRevScoreB = DB.m_DBSelfRevScores[j];
                DA.SetFromAln(out[p], paths + out[p].path_off);
                if (Accept(DA, MaxEvalue, MaxPvalue, MinTS)) { DA.ToTsv(fTsv, true); ++Q.m_HitCount; }
            }
        });
    };
    Q.m_HitCount += RunMKFPairsBeside(ctx, Params, O.columns, Q, DB, mkf, align_job,
                                      [&](const DSSAligner &TA) { return Accept(TA, MaxEvalue, MaxPvalue, MinTS); }, true, fTsv);
    tm.lap("align + replay | MKF side by side");
    fclose(fTsv);
}

}   // namespace reseek_amd

// compat.cpp -- the rest of the reference's class surface for the -search path, so that a caller written like
// search.cpp:20-111 compiles against reseek_host.h and runs on the GPU path:
//   process-wide state   g_Opts / g_Arg1 / g_fTsv, OpenOutputFiles / CloseOutputFiles (output.cpp:8-20),
//                        DSSParams::SetDSSParams(DECIDE_MODE) (dssparams.cpp:16-104)
//   ChainReader2         chainreader2.h:10 (Open / GetNext; .bca only)
//   ChainBag forms       DSSAligner::DoMKF_Bags / AlignBags / AlignBagsMKF (chainbag.cpp:6-84),
//                        MuKmerFilter::SetBagQ / AlignBag (mukmerfilter.h:82-87)
//   MuSeqSource, SeqDB   museqsource.cpp:21-75, seqdb.cpp (FromSS / ToLetters)
//   MuPreFilter / PostMuFilter with the argument lists of search.cpp:9-18
// Everything forwards to the batch C-ABI (batches of one pair where the reference's call is per pair).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "reseek_host.h"

namespace reseek_amd {

static void check(int rc, const char *what)
{
    if (rc != RSK_OK) throw std::runtime_error(std::string(what) + ": " + rsk_last_error());
}

SearchOptions g_Opts;
std::string g_Arg1;
FILE *g_fTsv = nullptr;

void OpenOutputFiles()
{
    g_fTsv = nullptr;
    if (g_Opts.output.empty()) return;                                     // CreateStdioFile("") == 0 (myutils.cpp)
    g_fTsv = fopen(g_Opts.output.c_str(), "w");
    if (!g_fTsv) throw std::runtime_error("OpenOutputFiles: cannot create " + g_Opts.output);
}

void CloseOutputFiles()
{
    if (g_fTsv) fclose(g_fTsv);
    g_fTsv = nullptr;
}

rsk_ctx *DefaultCtx()
{
    static struct holder {
        rsk_ctx *c = nullptr;
        std::mutex m;
        ~holder() { if (c) rsk_ctx_destroy(c); }
    } h;
    std::lock_guard<std::mutex> g(h.m);
    if (!h.c) {
        const char *e = getenv("RSK_DEVICE");
        check(rsk_ctx_create(e ? atoi(e) : 0, &h.c), "rsk_ctx_create");
    }
    return h.c;
}

std::mutex &CtxMutex(rsk_ctx *ctx)
{
    static std::mutex reg;
    static std::map<rsk_ctx *, std::unique_ptr<std::mutex>> m;
    std::lock_guard<std::mutex> g(reg);
    std::unique_ptr<std::mutex> &p = m[ctx];
    if (!p) p.reset(new std::mutex);
    return *p;
}

void DSSParams::SetDSSParams(DECIDE_MODE DM)
{
    SearchOptions o = g_Opts;
    switch (DM) {                                                            // GetAlgoMode dssparams.cpp:29-42
    case DM_AlwaysFast: o.mode = AM_Fast; break;
    case DM_AlwaysSensitive: o.mode = AM_Sensitive; break;
    case DM_AlwaysVerysensitive: o.mode = AM_VerySensitive; break;
    case DM_DefaultFast: if (o.mode == AM_Invalid) o.mode = AM_Fast; break;
    case DM_DefaultSensitive: if (o.mode == AM_Invalid) o.mode = AM_Sensitive; break;
    case DM_UseCommandLineOption:
        if (o.mode == AM_Invalid) throw std::runtime_error("Must set -fast, -sensitive or -verysensitive");   // dssparams.cpp:25
        break;
    default: throw std::runtime_error("SetDSSParams: invalid DECIDE_MODE");
    }
    SetDSSParams(o);
}

// ---- ChainReader2 ----------------------------------------------------------------------------------------------
void ChainReader2::Open(const std::string &FileName)
{
    OpenRange(FileName, 0, UINT64_MAX);
}

void ChainReader2::OpenRange(const std::string &FileName, uint64_t Lo, uint64_t Hi)
{
    const bool bca = FileName.size() >= 4 && FileName.compare(FileName.size() - 4, 4, ".bca") == 0;
    if (!bca) throw std::runtime_error("ChainReader2::Open: only .bca files are read on this path (" + FileName + ")");
    m_BCA.Open(FileName);
    m_CurrentFN = FileName;
    m_EndIdx_BCA = std::min<uint64_t>(Hi, m_BCA.GetChainCount());
    m_ChainIdx_BCA = std::min<uint64_t>(Lo, m_EndIdx_BCA);
}

PDBChain *ChainReader2::GetNext()
{
    uint64_t k;
    {
        std::lock_guard<std::mutex> g(m_CRGlobalLock);
        if (m_ChainIdx_BCA >= m_EndIdx_BCA) return nullptr;
        k = m_ChainIdx_BCA++;
    }
    std::unique_ptr<PDBChain> C(new PDBChain);
    m_BCA.ReadChain(k, *C);
    C->m_Idx = (uint) k;
    return C.release();
}

// ---- bag forms -------------------------------------------------------------------------------------------------
void MuKmerFilter::SetBagQ(const ChainBag &BagQ)
{
    SetQ(BagQ.m_ptrChain ? BagQ.m_ptrChain->m_Label : std::string(), BagQ.m_ptrMuLetters, BagQ.m_ptrMuKmers);
}

void MuKmerFilter::AlignBag(const ChainBag &BagT)
{
    Align(*BagT.m_ptrMuLetters, *BagT.m_ptrMuKmers);
}

bool DSSAligner::DoMKF_Bags(const ChainBag &BagA, const ChainBag &BagB) const
{
    if (BagA.m_ptrMuLetters == nullptr || BagB.m_ptrMuLetters == nullptr) return false;
    const uint LA = BagA.m_ptrChain->GetSeqLength(), LB = BagB.m_ptrChain->GetSeqLength();
    return LA >= m_Params->m_MKFL || LB >= m_Params->m_MKFL;
}

static void SetFromBags(DSSAligner &DA, const ChainBag &BagA, const ChainBag &BagB)
{
    DA.m_ChainA = BagA.m_ptrChain; DA.m_ChainB = BagB.m_ptrChain;
    DA.m_ProfileA = BagA.m_ptrProfile; DA.m_ProfileB = BagB.m_ptrProfile;
    DA.m_MuLettersA = BagA.m_ptrMuLetters; DA.m_MuLettersB = BagB.m_ptrMuLetters;
    DA.m_MuKmersA = BagA.m_ptrMuKmers; DA.m_MuKmersB = BagB.m_ptrMuKmers;
    DA.m_SelfRevScoreA = BagA.m_SelfRevScore; DA.m_SelfRevScoreB = BagB.m_SelfRevScore;
}

void DSSAligner::AlignBagsMKF(const ChainBag &BagA, const ChainBag &BagB)
{
    ClearAlign();
    SetFromBags(*this, BagA, BagB);
    m_MKF.SetBagQ(BagA);
    m_MKF.AlignBag(BagB);
    PostAlignMKF();
}

void DSSAligner::AlignBags(const ChainBag &BagA, const ChainBag &BagB)
{
    ClearAlign();
    SetFromBags(*this, BagA, BagB);
    if (DoMKF_Bags(BagA, BagB)) {
        m_MKF.SetBagQ(BagA);
        m_MKF.AlignBag(BagB);
        PostAlignMKF();
        return;
    }
    AlignPairOnGpu();                        // Mu filter (Omega > 0), SetSMx_NoRev + SWFast + CalcEvalue: chainbag.cpp:67-84
}

// ---- Mu sequence source / SeqDB ----------------------------------------------------------------------------------
// g_CharToLetterMu alpha.cpp:3291: the inverse of "ABCDEFGHIJLKMNOPQRSTUVWXYZabcdefghij" (g_LetterToCharMu) EXCEPT that
// 'K' -> 10 and 'L' -> 11 there, i.e. text written with g_LetterToCharMu comes back with letters 10 and 11 exchanged.
static struct CharToLetterMuInit {
    byte t[256];
    CharToLetterMuInit()
    {
        memset(t, 0xFF, sizeof(t));
        static const char Chars[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghij";
        for (int i = 0; i < 36; ++i) t[(unsigned char) Chars[i]] = (byte) i;
    }
} s_c2l;
const byte *const g_CharToLetterMu = s_c2l.t;

void MuSeqSource::OpenFasta(const std::string &FileName)
{
    m_IsFasta = true;
    ReadMuFasta(FileName, m_FaLabels, m_FaSeqs);
    m_FaNext = 0;
}

void MuSeqSource::OpenChains(const std::string &FileName, const DSSParams &Params)
{
    m_IsFasta = false;
    m_Params = &Params;
    m_CR.Open(FileName);
}

static const char s_MuLetterToChar[] = "ABCDEFGHIJLKMNOPQRSTUVWXYZabcdefghij";      // g_LetterToCharMu alpha.cpp:3550 (L before K)

bool MuSeqSource::GetNext(std::string &Label, std::vector<byte> &Seq)
{
    if (m_IsFasta) {
        if (m_FaNext >= m_FaSeqs.size()) return false;
        Label = m_FaLabels[m_FaNext];
        Seq = m_FaSeqs[m_FaNext++];                       // OpenFasta text is converted through g_CharToLetterMu in either mode (museqsource.cpp:23-30)
        return true;
    }
    std::unique_ptr<PDBChain> C(m_CR.GetNext());
    if (!C) return false;
    DSS D;
    D.SetParams(*m_Params);
    D.Init(*C);
    D.GetMuLetters(Seq);
    Label = C->m_Label;
    if (m_ASCII)
        for (byte &l : Seq) l = (byte) s_MuLetterToChar[l];
    return true;
}

void MuSeqSource::GetAll(std::vector<std::string> &Labels, std::vector<std::vector<byte> > &Seqs)
{
    Labels.clear(); Seqs.clear();
    if (m_IsFasta) {
        std::string L; std::vector<byte> S;
        while (GetNext(L, S)) { Labels.push_back(L); Seqs.push_back(S); }
        return;
    }
    // chains: read sequentially, featurise on the host threads
    std::vector<std::unique_ptr<PDBChain> > Chains;
    for (;;) { PDBChain *C = m_CR.GetNext(); if (!C) break; Chains.emplace_back(C); }
    const size_t N = Chains.size();
    Labels.resize(N); Seqs.resize(N);
    std::atomic<size_t> next{0};
    auto body = [&]() {
        DSS D;
        D.SetParams(*m_Params);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= N) return;
            D.Init(*Chains[i]);
            D.GetMuLetters(Seqs[i]);
            Labels[i] = Chains[i]->m_Label;
            if (m_ASCII)
                for (byte &l : Seqs[i]) l = (byte) s_MuLetterToChar[l];
        }
    };
    std::vector<std::thread> ts;
    const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>(HostThreads(128), N));
    for (unsigned t = 0; t < T; ++t) ts.emplace_back(body);
    for (auto &t : ts) t.join();
}

void SeqDB::FromSS(MuSeqSource &SS)
{
    std::vector<std::string> Labels;
    std::vector<std::vector<byte> > Seqs;
    SS.GetAll(Labels, Seqs);
    for (size_t i = 0; i < Seqs.size(); ++i) AddSeq(Labels[i], std::string(Seqs[i].begin(), Seqs[i].end()));
}

void SeqDB::ToLetters(const byte *CharToLetter)
{
    for (std::string &s : m_Seqs)
        for (char &c : s) c = (char) CharToLetter[(unsigned char) c];
}

// ---- search.cpp:9-18 forms ----------------------------------------------------------------------------------------
void MuPreFilterLetters(rsk_ctx *ctx, const std::vector<uint32_t> &qlen, const std::vector<uint8_t> &qmu, rsk_db *tdb, uint NT, int idx_mode,
                        uint rsb_size, const std::string &OutputFN);       // host/postmufilter.cpp

void MuPreFilter(const DSSParams &Params, SeqDB &QueryDB, MuSeqSource &FSS, const std::string &OutputFN)
{
    (void) Params;
    rsk_ctx *ctx = DefaultCtx();
    FSS.m_ASCII = false;                                   // muprefilter.cpp:84
    QueryDB.ToLetters(s_c2l.t);                            // muprefilter.cpp:88 (the L / K exchange happens here, as in the reference)
    const uint NQ = QueryDB.GetSeqCount();
    std::vector<uint32_t> qlen(NQ);
    std::vector<uint8_t> qmu;
    for (uint i = 0; i < NQ; ++i) {
        qlen[i] = QueryDB.GetSeqLength(i);
        for (char c : QueryDB.GetSeq(i)) {
            if ((byte) c >= RSK_MU_ALPHA) throw std::runtime_error("MuPreFilter: query sequence is not Mu text");
            qmu.push_back((uint8_t) c);
        }
    }
    std::vector<std::string> TLabels;
    std::vector<std::vector<byte> > TSeqs;
    FSS.GetAll(TLabels, TSeqs);
    const uint NT = (uint) TSeqs.size();
    std::vector<uint32_t> tlen(NT);
    std::vector<uint8_t> tmu;
    for (uint i = 0; i < NT; ++i) { tlen[i] = (uint32_t) TSeqs[i].size(); tmu.insert(tmu.end(), TSeqs[i].begin(), TSeqs[i].end()); }
    rsk_db *tdb = nullptr;
    check(rsk_db_create(ctx, NT, tlen.data(), tmu.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &tdb), "rsk_db_create");
    struct guard { rsk_db *d; ~guard() { rsk_db_destroy(d); } } g{ tdb };
    MuPreFilterLetters(ctx, qlen, qmu, tdb, NT, g_Opts.idx_mode, g_Opts.rsb_size, OutputFN);
}

void ReadHandOff(const std::string &TsvFN, uint NQ, uint64_t NT, std::vector<uint32_t> &pq, std::vector<uint32_t> &pt, bool &NoHits);   // host/postmufilter.cpp
void PostMuFilterPairs(const DSSParams &Params, DBSearcher &Q, DBSearcher &DB, const std::vector<uint32_t> &pq, const std::vector<uint32_t> &pt,
                       const std::string &HitsFN);                                                                                    // host/postmufilter.cpp

// postmufilter.cpp:190-290: the query chains are loaded whole (with their self-rev scores under these params, :79), the
// DB chains by the target indexes of the hand-off file only (BCAData random access, :157-172).
void PostMuFilter(const DSSParams &Params, const std::string &MuFilterTsvFN, const std::string &QueryCAFN, const std::string &DBBCAFN,
                  const std::string &HitsFN)
{
    rsk_ctx *ctx = DefaultCtx();
    DBSearcher Q;
    Q.m_Params = &Params;
    Q.m_SelfRevQueryFlavour = true;
    Q.m_Opts = g_Opts;
    Q.m_Ctx = ctx;
    Q.LoadDB(QueryCAFN);
    BCAData B;
    B.Open(DBBCAFN);
    std::vector<uint32_t> pq, pt;
    bool NoHits = false;
    ReadHandOff(MuFilterTsvFN, Q.GetDBChainCount(), B.GetChainCount(), pq, pt, NoHits);
    if (NoHits) return;
    std::vector<uint32_t> uniq(pt);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    std::vector<PDBChain *> Chains;
    for (uint32_t t : uniq) {
        PDBChain *C = new PDBChain;
        B.ReadChain(t, *C);
        if (C->GetSeqLength() < 1) { delete C; throw std::runtime_error("PostMuFilter: empty chain listed in the hand-off file"); }
        Chains.push_back(C);
    }
    DBSearcher DB;
    DB.m_Params = &Params;
    DB.m_SelfRevQueryFlavour = true;                       // postmufilter.cpp:171
    DB.m_Opts = g_Opts;
    DB.m_Ctx = ctx;
    DB.LoadChains(Chains);
    for (uint32_t &t : pt) t = (uint32_t) (std::lower_bound(uniq.begin(), uniq.end(), t) - uniq.begin());
    PostMuFilterPairs(Params, Q, DB, pq, pt, HitsFN);
}

}   // namespace reseek_amd

// dssaligner.cpp -- host mirror of DSSParams / DSSAligner / MuKmerFilter (see reseek_host.h).
// The per-pair DP of the common path runs on the GPU (k_mu_sw.hip, k_sw_float.hip); what lives here is
//   * mode presets and option overrides                     dssparams.cpp:44-104, namedparams.cpp:32-53
//   * result bookkeeping and the -columns / TSV formatting  dssaligner.cpp:100-136,1016, userfields.cpp, cigar.cpp:95
//   * the long-chain MKF path, host resident for now: 3-mer seeds -> ungapped
//     integer X-drop HSPs -> chain -> banded float X-drop from the best 8-mer
//     (mukmerfilter.cpp:105-460, chainer.cpp:31, dssaligner.cpp:488,1387-1430, xdrophsp.cpp:42,
//      xdropfwd.cpp:71, xdropbwd.cpp:28, mergefwdback.cpp:6)
// Compiled with -ffp-contract=off: float results must match the reference's strict-IEEE build.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "reseek_host.h"
#include "../rsk_tables_data.h"

void rsk_set_error(const char *fmt, ...);

namespace reseek_amd {

static const float MINUS_INFINITY = -9e9f;   // xdpmem.h:6
static const byte TRACEBITS_DM = 0x01, TRACEBITS_IM = 0x02, TRACEBITS_MD = 0x04, TRACEBITS_MI = 0x08;

std::mutex DSSAligner::m_OutputLock;

// ---------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------
void DSSParams::SetDefaults()
{
    m_GapOpen = rsk_gap_open;      // -0.685533f
    m_GapExt = rsk_gap_ext;        // -0.051881f
    m_MinFwdScore = 7.0f;
    m_Omega = 29;
    m_OmegaFwd = 29;
    m_MKFPatternStr = "111";
}

void DSSParams::SetDSSParams(const SearchOptions &o)
{
    SetDefaults();
    switch (o.mode) {
    case AM_Fast:
        m_Omega = 22; m_OmegaFwd = 50; m_MKFL = 500; m_MKF_X1 = 8; m_MKF_X2 = 8; m_MKF_MinHSPScore = 50; m_MKF_MinMegaHSPScore = -4;
        break;
    case AM_Sensitive:
        m_Omega = 12; m_OmegaFwd = 20; m_MKFL = 600; m_MKF_X1 = 8; m_MKF_X2 = 8; m_MKF_MinHSPScore = 50; m_MKF_MinMegaHSPScore = -4;
        break;
    case AM_VerySensitive:
        m_Omega = 0; m_OmegaFwd = 0; m_MKFL = 99999; m_MKF_X1 = 99999; m_MKF_X2 = 99999; m_MKF_MinHSPScore = 0;
        m_MKF_MinMegaHSPScore = -99999; m_MinFwdScore = 0;
        break;
    default:
        break;
    }
    if (o.omega_set) m_Omega = o.omega;
    if (o.omegafwd_set) m_OmegaFwd = o.omegafwd;
    if (o.minfwdscore_set) m_MinFwdScore = o.minfwdscore;
    if (o.gapopen_set) { m_GapOpen = -o.gapopen; m_GapExt = -o.gapext; }   // -gapext only with -gapopen (dssparams.cpp:96-97)
    if (o.mkfl_set) m_MKFL = o.mkfl;
}

USERFIELD StrToUF(const std::string &s)
{
    static const char *names[] = { "", "query", "target", "pvalue", "evalue", "qlo", "qhi", "tlo", "thi", "ql", "tl", "pctid",
                                   "cigar", "qrow", "trow", "qrowg", "trowg", "ts", "newts", "dpscore", "lddt", "ids", "gaps", "aq",
                                   "muhsp", "muchain", "gscore", "raw", "muscore", "qcovpct", "tcovpct" };
    for (int k = 1; k <= (int) UF_tcovpct; ++k)
        if (s == names[k]) return (USERFIELD) k;
    return UF_Undefined;
}

// ---------------------------------------------------------------------------------------------
// small path helpers
// ---------------------------------------------------------------------------------------------
void InvertPath(const std::string &Path, std::string &Inv)
{
    Inv.clear();
    Inv.reserve(Path.size());
    for (char c : Path) Inv += (c == 'D') ? 'I' : (c == 'I') ? 'D' : c;
}

void GetPathCounts(const std::string &Path, uint &M, uint &D, uint &I)
{
    M = D = I = 0;
    for (char c : Path) {
        if (c == 'M') ++M;
        else if (c == 'D') ++D;
        else if (c == 'I') ++I;
    }
}

// cigar.cpp:95-140.  Run-length encode; D and I are exchanged unless FlipDI.
void PathToCIGAR(const char *Path, std::string &CIGAR, bool FlipDI)
{
    CIGAR.clear();
    auto emit = [&](uint n, char c) {
        if (!FlipDI) c = (c == 'D') ? 'I' : (c == 'I') ? 'D' : c;
        char tmp[32];
        snprintf(tmp, sizeof(tmp), "%u%c", n, c);
        CIGAR += tmp;
    };
    char last = *Path;
    uint n = 1;
    for (uint i = 1;; ++i) {
        const char c = Path[i];
        if (c == 0) break;
        if (c == last) { ++n; continue; }
        emit(n, last);
        last = c;
        n = 1;
    }
    if (n > 0) emit(n, last);
}

// ---------------------------------------------------------------------------------------------
// DSSAligner: state
// ---------------------------------------------------------------------------------------------
DSSAligner::DSSAligner() { SetColumns(""); }

void DSSAligner::SetColumns(const std::string &Columns)
{
    auto default_columns = [&]() {
        static const USERFIELD d[] = { UF_query, UF_target, UF_qlo, UF_qhi, UF_ql, UF_tlo, UF_thi, UF_tl, UF_pctid, UF_pvalue };
        for (USERFIELD u : d) m_UFs.push_back(u);
    };
    m_UFs.clear();
    if (Columns.empty()) { default_columns(); return; }
    size_t pos = 0;
    while (pos <= Columns.size()) {
        size_t e = Columns.find('+', pos);
        if (e == std::string::npos) e = Columns.size();
        const std::string f = Columns.substr(pos, e - pos);
        if (f == "std") default_columns();
        else if (!f.empty()) m_UFs.push_back(StrToUF(f));
        pos = e + 1;
    }
}

void DSSAligner::SetParams(const DSSParams &Params)
{
    m_Params = &Params;
    m_MKF.SetParams(Params);
}

void DSSAligner::UnsetQuery()
{
    m_MKF.ResetQ();
    m_ChainA = nullptr; m_ProfileA = nullptr; m_MuLettersA = nullptr; m_MuKmersA = nullptr;
    m_SelfRevScoreA = 0;
}

void DSSAligner::SetQuery(const PDBChain &Chain, const std::vector<std::vector<byte> > *ptrProfile, const std::vector<byte> *ptrMuLetters,
                          const std::vector<uint> *ptrMuKmers, float SelfRevScore)
{
    if (ptrMuKmers != nullptr && ptrMuLetters != nullptr) m_MKF.SetQ(Chain.m_Label, ptrMuLetters, ptrMuKmers);
    m_ChainA = &Chain; m_ProfileA = ptrProfile; m_MuLettersA = ptrMuLetters; m_MuKmersA = ptrMuKmers;
    m_SelfRevScoreA = SelfRevScore;
    // the striped parasail profiles of SetMuQP_Para are built inside the GPU kernel
}

void DSSAligner::SetTarget(const PDBChain &Chain, const std::vector<std::vector<byte> > *ptrProfile, const std::vector<byte> *ptrMuLetters,
                           const std::vector<uint> *ptrMuKmers, float SelfRevScore)
{
    m_ChainB = &Chain; m_ProfileB = ptrProfile; m_MuKmersB = ptrMuKmers; m_MuLettersB = ptrMuLetters;
    m_SelfRevScoreB = SelfRevScore;
}

bool DSSAligner::DoMKF() const
{
    if (m_MuLettersA == nullptr || m_MuLettersB == nullptr) return false;
    if (m_MuKmersA == nullptr || m_MuKmersB == nullptr) return false;
    if (m_MuLettersA->empty() || m_MuLettersB->empty()) return false;
    if (m_MuKmersA->empty() || m_MuKmersB->empty()) return false;
    const uint LA = m_ChainA->GetSeqLength(), LB = m_ChainB->GetSeqLength();
    return LA >= m_Params->m_MKFL || LB >= m_Params->m_MKFL;
}

void DSSAligner::ClearAlign()
{
    m_Path.clear();
    m_LoA = m_LoB = m_HiA = m_HiB = UINT_MAX;
    m_Ids = m_Gaps = m_IdentCount = UINT_MAX;
    m_PvalueA = m_PvalueB = m_EvalueA = m_EvalueB = FLT_MAX;
    m_TestStatisticA = m_TestStatisticB = -FLT_MAX;
    m_NewTestStatisticA = m_NewTestStatisticB = -FLT_MAX;
    m_AlnFwdScore = 0;
    m_LDDT = FLT_MAX;
}

void DSSAligner::SetFromAln(const rsk_aln &a, const char *Path)
{
    ClearAlign();
    m_AlnFwdScore = a.score;
    m_Path.assign(Path, a.path_len);
    m_LoA = a.lo_a; m_LoB = a.lo_b;
    if (a.hi_a != RSK_NO_POS) { m_HiA = a.hi_a; m_HiB = a.hi_b; }      // the long-chain batch sets Hi also below MinFwdScore (PostAlignMKF)
    if (a.evalue != FLT_MAX) {
        m_HiA = a.hi_a; m_HiB = a.hi_b; m_Ids = a.ids; m_Gaps = a.gaps;
        m_IdentCount = a.nident == RSK_NO_POS ? UINT_MAX : a.nident;
        m_LDDT = a.lddt;
        m_NewTestStatisticA = m_NewTestStatisticB = a.ts;
        m_PvalueA = m_PvalueB = a.pvalue;
        m_EvalueA = m_EvalueB = a.evalue;
        m_QualityA = m_QualityB = a.qual;
    }
}

// ---------------------------------------------------------------------------------------------
// statistics on the host (MKF path): lddt.cpp:63-124, statsig.cpp:27-50, dssaligner.cpp:852-904
// ---------------------------------------------------------------------------------------------
float DSSAligner::GetLDDT() const
{
    std::vector<uint> PosAs, PosBs;
    uint PosA = m_LoA, PosB = m_LoB;
    for (char c : m_Path) {
        if (c == 'M') { PosAs.push_back(PosA++); PosBs.push_back(PosB++); }
        else if (c == 'D') ++PosA;
        else if (c == 'I') ++PosB;
    }
    const uint n = (uint) PosAs.size();
    if (n == 0) return 0;
    const PDBChain &Q = *m_ChainA, &T = *m_ChainB;
    // coordinates of the aligned columns, contiguous: the squared distances of a column to all later columns are a
    // branch-free (vectorisable) loop; the few pairs within R0 are finished in a second pass
    static thread_local std::vector<float> buf;
    buf.resize((size_t) 8 * n);
    float *ax = buf.data(), *ay = ax + n, *az = ay + n, *bx = az + n, *by = bx + n, *bz = by + n, *d1 = bz + n, *d2 = d1 + n;
    for (uint c = 0; c < n; ++c) {
        ax[c] = Q.m_Xs[PosAs[c]]; ay[c] = Q.m_Ys[PosAs[c]]; az[c] = Q.m_Zs[PosAs[c]];
        bx[c] = T.m_Xs[PosBs[c]]; by[c] = T.m_Ys[PosBs[c]]; bz[c] = T.m_Zs[PosBs[c]];
    }
    std::vector<uint> cons(n, 0), pres(n, 0);
    const float R0sq = 15.0f * 15.0f;
    for (uint ci = 0; ci < n; ++ci) {
        const float x1 = ax[ci], y1 = ay[ci], z1 = az[ci], u1 = bx[ci], v1 = by[ci], w1 = bz[ci];
        for (uint cj = ci + 1; cj < n; ++cj) {
            const float dx = x1 - ax[cj], dy = y1 - ay[cj], dz = z1 - az[cj];
            const float ex = u1 - bx[cj], ey = v1 - by[cj], ez = w1 - bz[cj];
            d1[cj] = dx * dx + dy * dy + dz * dz;
            d2[cj] = ex * ex + ey * ey + ez * ez;
        }
        uint consi = 0, presi = 0;
        for (uint cj = ci + 1; cj < n; ++cj) {
            if (d1[cj] > R0sq && d2[cj] > R0sq) continue;
            const float diff = fabsf(sqrtf(d1[cj]) - sqrtf(d2[cj]));
            const uint k = (diff <= 0.5f) + (diff <= 1.0f) + (diff <= 2.0f) + (diff <= 4.0f);
            presi += k; pres[cj] += k;
            consi += 4; cons[cj] += 4;
        }
        cons[ci] += consi; pres[ci] += presi;
    }
    float total = 0;
    for (uint c = 0; c < n; ++c) {
        float s = 0;
        if (cons[c] > 0) s = float(pres[c]) / cons[c];
        total += s;
    }
    return total / n;
}

static double StatSig_GetPvalue(double TS)
{
    const double l = (TS < 0.11) ? (-80.0 * TS + -0.58) : (-52.0 * TS + -3.7);
    double P = pow(10, l);
    if (P > 1) P = 1;
    return P;
}
static double StatSig_GetQual(double TS)
{
    const double logE = 5.0 + -40.0 * TS;
    if (logE < -20) return 1;
    const double x = pow(10, logE / 10);
    return 1 / (1 + x / 2);
}

void DSSAligner::CalcEvalue()
{
    if (m_AlnFwdScore < m_Params->m_MinFwdScore) return;
    uint M, D, I;
    GetPathCounts(m_Path, M, D, I);
    m_HiA = m_LoA + M + D - 1;
    m_HiB = m_LoB + M + I - 1;
    m_Ids = M;
    m_Gaps = D + I;
    const float LDDT = GetLDDT();
    m_LDDT = LDDT;
    float RevDPScore = 0;
    if (m_SelfRevScoreA != FLT_MAX && m_SelfRevScoreB != FLT_MAX) RevDPScore = (m_SelfRevScoreA + m_SelfRevScoreB) / 2;
    const uint LA = m_ChainA->GetSeqLength(), LB = m_ChainB->GetSeqLength();
    const float L = float(LA + LB) / 2;
    const float dpw = 1.7f, lddtw = 0.13f, ladd = 250.0f, revtsw = 2.0f;
    m_NewTestStatisticA = lddtw * LDDT;
    m_NewTestStatisticA += (dpw * m_AlnFwdScore - revtsw * RevDPScore) / (L + ladd);
    m_NewTestStatisticB = m_NewTestStatisticA;
    const float Pval = (float) StatSig_GetPvalue(m_NewTestStatisticA);
    const float Qual = (float) StatSig_GetQual(m_NewTestStatisticA);
    const float E = (float) (StatSig_GetPvalue(m_NewTestStatisticA) * 8340);
    m_QualityA = m_QualityB = Qual;
    m_PvalueA = m_PvalueB = Pval;
    m_EvalueA = m_EvalueB = E;
}

// ---------------------------------------------------------------------------------------------
// MuKmerFilter (mukmerfilter.cpp): the per-pair form of the long-chain seeding.  No seeding or extension loop runs on the
// host (r04): SetQ only remembers the query, Align is a device batch of ONE pair through the entry point the search uses
// (rsk_mkf_seed_pairs: compact 3-mer table, ungapped X-drop, keep rule -- k_mkf.hip), then ChainHSPs.
// ---------------------------------------------------------------------------------------------
void MuKmerFilter::ResetQ()
{
    m_ptrMuKmersQ = nullptr;
    m_ptrMuLettersQ = nullptr;
}

void MuKmerFilter::SetQ(const std::string &, const std::vector<byte> *ptrMuLettersQ, const std::vector<uint> *ptrMuKmersQ)
{
    m_ptrMuLettersQ = ptrMuLettersQ;
    m_ptrMuKmersQ = ptrMuKmersQ;
}

namespace {
void rsk_ok(int rc, const char *what)
{
    if (rc != RSK_OK) throw std::runtime_error(std::string(what) + ": " + rsk_last_error());
}

// one chain's Mu letters as a device set
struct MuSet {
    rsk_db *d = nullptr;
    MuSet(rsk_ctx *ctx, const std::vector<byte> &Mu)
    {
        const uint32_t L = (uint32_t) Mu.size();
        rsk_ok(rsk_db_create(ctx, 1, &L, Mu.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &d), "rsk_db_create");
    }
    ~MuSet() { rsk_db_destroy(d); }
    MuSet(const MuSet &) = delete;
    MuSet &operator=(const MuSet &) = delete;
};
}   // namespace

void MuKmerFilter::Align(const std::vector<byte> &MuLettersT, const std::vector<uint> &)
{
    m_ptrMuLettersT = &MuLettersT;
    SetSeedHSPs(nullptr, 0);
    if (!m_ptrMuLettersQ || m_ptrMuLettersQ->size() < 3 || MuLettersT.size() < 3) return;
    rsk_ctx *ctx = m_Ctx ? m_Ctx : DefaultCtx();       // the owning aligner's context (and device)
    std::lock_guard<std::mutex> lock(CtxMutex(ctx));
    MuSet Q(ctx, *m_ptrMuLettersQ), T(ctx, MuLettersT);
    const uint32_t zero = 0;
    // the search's record size first; a pair that keeps more HSPs than that runs again with the large one
    for (uint32_t cap : { 32u, 1024u }) {
        std::vector<int32_t> kept((size_t) cap * 4);
        uint8_t found = 0;
        uint32_t rec_pair = 0, rec_nkept = 0;
        size_t nrec = 0;
        rsk_ok(rsk_mkf_seed_pairs(ctx, Q.d, T.d, &zero, &zero, 1, m_Params->m_MKF_X1, m_Params->m_MKF_MinHSPScore, cap, &found, 1, &nrec, &rec_pair,
                                  &rec_nkept, kept.data()),
               "rsk_mkf_seed_pairs");
        if (!found || nrec == 0) return;
        if (rec_nkept <= cap) { SetSeedHSPs(kept.data(), rec_nkept); return; }
    }
    throw std::runtime_error("MuKmerFilter::Align: a pair keeps more than 1024 seed HSPs");
}

// Chainer::Chain (chainer.cpp:31-178) for the pairs the device hands back: k_mkf_chain chains every pair whose outcome is
// defined by the sweep alone and flags those where two intervals END at one position with equal chain scores -- there the
// reference's result is whatever its libc qsort made of a comparator that calls such end points equal (chainer.cpp:11-29),
// so exactly that is done here: the same comparator through the same qsort, then the sweep.
namespace {
struct EndPoint { uint32_t pos, hsp; int opens; };
int CompareEndPoints(const void *pa, const void *pb)
{
    const EndPoint &a = *(const EndPoint *) pa, &b = *(const EndPoint *) pb;
    if (a.pos != b.pos) return a.pos < b.pos ? -1 : 1;
    return b.opens - a.opens;                              // at one position an opening end point first; two of a kind: "equal"
}
}   // namespace

void MuKmerFilter::ChainHSPs()
{
    m_ChainHSPLois.clear(); m_ChainHSPLojs.clear(); m_ChainHSPLens.clear();
    m_BestChainScore = 0;
    const uint32_t N = (uint32_t) m_MuKmerHSPLois.size();
    if (N == 0) return;
    std::vector<EndPoint> ev;
    ev.reserve(2 * (size_t) N);
    for (uint32_t h = 0; h < N; ++h) {
        ev.push_back(EndPoint{ (uint32_t) m_MuKmerHSPLois[h], h, 1 });
        ev.push_back(EndPoint{ (uint32_t) (m_MuKmerHSPLois[h] + m_MuKmerHSPLens[h] - 1), h, 0 });
    }
    qsort(ev.data(), ev.size(), sizeof(EndPoint), CompareEndPoints);
    // sweep: an HSP that opens continues the best chain closed so far; an HSP that closes becomes that chain if strictly better
    const uint32_t NONE = UINT_MAX;
    std::vector<uint32_t> before(N, NONE);
    std::vector<float> chain(N, MINUS_INFINITY);
    uint32_t closed = NONE;
    for (const EndPoint &e : ev) {
        if (e.opens) {
            before[e.hsp] = closed;
            chain[e.hsp] = (float) m_MuKmerHSPScores[e.hsp] + (closed == NONE ? 0.0f : chain[closed]);
        } else if (closed == NONE || chain[e.hsp] > chain[closed])
            closed = e.hsp;
    }
    float total = 0;
    for (uint32_t h = closed; h != NONE; h = before[h]) {                 // chain end -> start: the order PostAlignMKF sums in
        total += (float) m_MuKmerHSPScores[h];
        m_ChainHSPLois.push_back(m_MuKmerHSPLois[h]);
        m_ChainHSPLojs.push_back(m_MuKmerHSPLojs[h]);
        m_ChainHSPLens.push_back(m_MuKmerHSPLens[h]);
    }
    m_BestChainScore = (int) total;
}

// ---------------------------------------------------------------------------------------------
// Long-chain pair, per-pair form (AlignMKF dssaligner.cpp:1387, XDropHSP xdrophsp.cpp:42).  The gapped X-drop
// extensions exist on the device only (k_xdrop.hip): a single pair is a batch of one through the same entry points the
// search uses (rsk_mkf_align_pairs / rsk_xdrop_pairs).
// ---------------------------------------------------------------------------------------------
namespace {
// the two chains of the aligner's current pair as one-chain device sets
struct PairSets {
    rsk_db *a = nullptr, *b = nullptr;
    PairSets(rsk_ctx *ctx, const DSSAligner &DA)
    {
        a = make(ctx, *DA.m_ChainA, *DA.m_ProfileA, DA.m_MuLettersA, DA.m_SelfRevScoreA);
        try { b = make(ctx, *DA.m_ChainB, *DA.m_ProfileB, DA.m_MuLettersB, DA.m_SelfRevScoreB); }
        catch (...) { rsk_db_destroy(a); throw; }
    }
    ~PairSets() { rsk_db_destroy(a); rsk_db_destroy(b); }
    PairSets(const PairSets &) = delete;
    PairSets &operator=(const PairSets &) = delete;
    static rsk_db *make(rsk_ctx *ctx, const PDBChain &C, const std::vector<std::vector<byte> > &Prof, const std::vector<byte> *Mu, float SelfRev)
    {
        const uint32_t L = C.GetSeqLength();
        std::vector<uint8_t> flat((size_t) L * RSK_NFEATURES);
        for (uint f = 0; f < RSK_NFEATURES; ++f) memcpy(&flat[(size_t) f * L], Prof[f].data(), L);
        rsk_db *db = nullptr;
        rsk_ok(rsk_db_create(ctx, 1, &L, Mu ? Mu->data() : nullptr, flat.data(), C.m_Xs.data(), C.m_Ys.data(), C.m_Zs.data(), &SelfRev, &db), "rsk_db_create");
        return db;
    }
};

// mergefwdback.cpp:6: the backward path ends at (BwdHiA, BwdHiB) = start - 1, the forward path begins at the start
void JoinExtensions(uint FwdLoA, uint FwdLoB, const std::string &FwdPath, uint BwdHiA, uint BwdHiB, const std::string &BwdPath, uint &LoA, uint &LoB,
                    uint &HiA, uint &HiB, std::string &Path)
{
    uint m, d, i;
    GetPathCounts(BwdPath, m, d, i);
    LoA = BwdPath.empty() ? FwdLoA : BwdHiA + 1 - (m + d);
    LoB = BwdPath.empty() ? FwdLoB : BwdHiB + 1 - (m + i);
    GetPathCounts(FwdPath, m, d, i);
    HiA = FwdPath.empty() ? BwdHiA : FwdLoA + (m + d) - 1;
    HiB = FwdPath.empty() ? BwdHiB : FwdLoB + (m + i) - 1;
    Path = BwdPath;
    Path += FwdPath;
}
}   // namespace

float DSSAligner::SubstScore(uint PosA, uint PosB)
{
    float Total = 0;
    for (uint f = 0; f < RSK_NFEATURES; ++f)
        Total += rsk_feature_mx[f][RSK_FEATURE_DIM * (*m_ProfileA)[f][PosA] + (*m_ProfileB)[f][PosB]];
    return Total;
}

float DSSAligner::GetMegaHSPScore(uint Lo_i, uint Lo_j, uint Len)
{
    float Total = 0;
    for (uint f = 0; f < RSK_NFEATURES; ++f) {
        const std::vector<byte> &RowA = (*m_ProfileA)[f], &RowB = (*m_ProfileB)[f];
        for (uint k = 0; k < Len; ++k) Total += rsk_feature_mx[f][RSK_FEATURE_DIM * RowA[Lo_i + k] + RowB[Lo_j + k]];
    }
    return Total;
}

// xdrophsp.cpp:42: start = the best 8-mer of the HSP (its middle if it has none), both gapped extensions from there
// (one-pair batch on the device), total < 10 => no alignment, else the joined path
float DSSAligner::XDropHSP(uint Loi_in, uint Loj_in, uint Len, uint &Loi_out, uint &Loj_out, uint &Hii_out, uint &Hij_out)
{
    Loi_out = Loj_out = Hii_out = Hij_out = UINT_MAX;
    m_XDropPath.clear();
    const uint LA = m_ChainA->GetSeqLength(), LB = m_ChainB->GetSeqLength();
    const uint K = 8;
    uint StartA = Loi_in + Len / 2, StartB = Loj_in + Len / 2;
    std::vector<float> Col(Len);
    for (uint c = 0; c < Len; ++c) Col[c] = SubstScore(Loi_in + c, Loj_in + c);
    float BestWindow = 0;
    for (uint w = 0; w + K <= Len; ++w) {
        float Window = 0;
        for (uint k = 0; k < K; ++k) Window += Col[w + k];       // every window summed afresh, in column order
        if (Window > BestWindow) { BestWindow = Window; StartA = Loi_in + w; StartB = Loj_in + w; }
    }
    if (std::min(StartA, StartB) < K / 2) { StartA += K / 2; StartB += K / 2; }
    if (StartA >= LA || StartB >= LB) return 0;                  // (the reference's extents would wrap around)
    if (!m_Ctx) m_Ctx = DefaultCtx();
    std::lock_guard<std::mutex> lock(CtxMutex(m_Ctx));
    PairSets Sets(m_Ctx, *this);
    const uint32_t zero = 0, lo_a = StartA, lo_b = StartB;
    float ScoreFwd = 0, ScoreBwd = 0;
    std::vector<char> buf((size_t) LA + LB + 8);
    uint64_t fo = 0, bo = 0;
    uint32_t fl = 0, bl = 0;
    rsk_ok(rsk_xdrop_pairs(m_Ctx, Sets.a, Sets.b, &zero, &zero, &lo_a, &lo_b, 1, float(m_Params->m_MKF_X2), m_Params->m_GapOpen, m_Params->m_GapExt,
                           &ScoreFwd, &ScoreBwd, buf.data(), buf.size(), &fo, &fl, &bo, &bl),
           "rsk_xdrop_pairs");
    const float TotalScore = ScoreFwd + ScoreBwd;
    if (TotalScore < 10) return 0;
    JoinExtensions(StartA, StartB, std::string(buf.data() + fo, fl), StartA - 1, StartB - 1, std::string(buf.data() + bo, bl), Loi_out, Loj_out, Hii_out,
                   Hij_out, m_XDropPath);
    return TotalScore;
}

void DSSAligner::AlignMKF()
{
    ClearAlign();
    m_MKF.m_Ctx = m_Ctx;
    m_MKF.Align(*m_MuLettersB, *m_MuKmersB);
    PostAlignMKF();
}

void MuKmerFilter::SetSeedHSPs(const int32_t *Kept4, uint Count)
{
    m_MuKmerHSPLois.clear(); m_MuKmerHSPLojs.clear(); m_MuKmerHSPLens.clear(); m_MuKmerHSPScores.clear();
    m_ChainHSPLois.clear(); m_ChainHSPLojs.clear(); m_ChainHSPLens.clear();
    m_BestChainScore = 0;
    m_BestHSPScore = 0;
    for (uint k = 0; k < Count; ++k) {
        m_MuKmerHSPLois.push_back(Kept4[4 * k]); m_MuKmerHSPLojs.push_back(Kept4[4 * k + 1]);
        m_MuKmerHSPLens.push_back(Kept4[4 * k + 2]); m_MuKmerHSPScores.push_back(Kept4[4 * k + 3]);
        m_BestHSPScore = std::max(m_BestHSPScore, (int) Kept4[4 * k + 3]);
    }
    if (Count) ChainHSPs();                               // FoundHSP (mukmerfilter.cpp:387-388)
}

void DSSAligner::AlignMKF_FromSeeds(const int32_t *Kept4, uint Count)
{
    ClearAlign();
    m_MKF.SetSeedHSPs(Kept4, Count);
    PostAlignMKF();
}

// dssaligner.cpp:1395: the chained HSPs of the current pair -> mega-HSP gate, XDropHSP, statistics: a batch of one
// through rsk_mkf_align_pairs (k_mkf_start / k_xdrop_wave / k_mkf_merge / k_lddt)
void DSSAligner::PostAlignMKF()
{
    if (m_MKF.m_BestChainScore <= 0) return;
    const size_t M = m_MKF.m_ChainHSPLois.size();
    if (M == 0) return;
    if (!m_Ctx) m_Ctx = DefaultCtx();
    std::lock_guard<std::mutex> lock(CtxMutex(m_Ctx));
    PairSets Sets(m_Ctx, *this);
    std::vector<int32_t> lo_a(m_MKF.m_ChainHSPLois.begin(), m_MKF.m_ChainHSPLois.end()), lo_b(m_MKF.m_ChainHSPLojs.begin(), m_MKF.m_ChainHSPLojs.end()),
        len(m_MKF.m_ChainHSPLens.begin(), m_MKF.m_ChainHSPLens.end());
    const uint32_t zero = 0, first[2] = { 0, (uint32_t) M };
    rsk_aln out;
    uint8_t status = 0;
    std::vector<char> paths((size_t) m_ChainA->GetSeqLength() + m_ChainB->GetSeqLength() + 24);
    rsk_ok(rsk_mkf_align_pairs(m_Ctx, Sets.a, Sets.b, &zero, &zero, 1, first, lo_a.data(), lo_b.data(), len.data(), float(m_Params->m_MKF_X2),
                               m_Params->m_GapOpen, m_Params->m_GapExt, m_Params->m_MKF_MinMegaHSPScore, m_Params->m_MinFwdScore, &out, &status,
                               paths.data(), paths.size()),
           "rsk_mkf_align_pairs");
    if (out.path_len == 0) return;
    SetFromAln(out, paths.data() + out.path_off);
    m_XDropScore = m_AlnFwdScore;
    m_XDropPath = m_Path;
}

// ---------------------------------------------------------------------------------------------
// output (dssaligner.cpp:1016-1034, userfields.cpp:19-152, dssaligner.cpp:1119-1281,1325-1372)
// ---------------------------------------------------------------------------------------------
double DSSAligner::GetQCovPct(bool Top) const
{
    const uint QL = GetQL(Top);
    if (QL == 0) return 0;
    double Pct = (100.0 * (GetHi(Top) - GetLo(Top) + 1)) / QL;
    if (Pct > 100) Pct = 100;
    return Pct;
}

double DSSAligner::GetTCovPct(bool Top) const
{
    const uint TL = GetQL(Top);          // sic (dssaligner.cpp:1134)
    double Pct = (100.0 * (GetHi(!Top) - GetLo(!Top) + 1)) / TL;
    if (Pct > 100) Pct = 100;
    return Pct;
}

float DSSAligner::GetPctId() const                    // dssaligner.cpp:1325: identical residues / aligned columns
{
    // the batch kernels counted the identical columns while walking the path (rsk_aln.nident): same integers, same division
    if (m_IdentCount != UINT_MAX && m_Ids != UINT_MAX) return m_Ids == 0 ? 0 : (m_IdentCount * 100.0f) / m_Ids;
    uint PosA = m_LoA, PosB = m_LoB, N = 0, n = 0;
    const char *SeqA = m_ChainA->m_Seq.data(), *SeqB = m_ChainB->m_Seq.data();
    const char *P = m_Path.data();
    const size_t L = m_Path.size();
    // run by run (paths are mostly long M runs): the end of a run is found eight path characters at a time, the
    // comparison loop of a run has no data-dependent branch (it vectorises)
    for (size_t k = 0; k < L;) {
        const char c = P[k];
        size_t e = k + 1;
        const uint64_t cc = 0x0101010101010101ull * (uint8_t) c;
        while (e + 8 <= L) {
            uint64_t w;
            memcpy(&w, P + e, 8);
            w ^= cc;
            if (w) { e += (size_t) (__builtin_ctzll(w) >> 3); goto run_end; }
            e += 8;
        }
        while (e < L && P[e] == c) ++e;
    run_end:
        const uint r = (uint) (e - k);
        if (c == 'M') {
            const char *a = SeqA + PosA, *b = SeqB + PosB;
            uint eq = 0;
            for (uint t = 0; t < r; ++t) eq += a[t] == b[t];
            n += eq; PosA += r; PosB += r; N += r;
        } else if (c == 'D') PosA += r;
        else if (c == 'I') PosB += r;
        k = e;
    }
    return N == 0 ? 0 : (n * 100.0f) / N;
}

void DSSAligner::GetRow(bool Up, bool Top, bool Global, std::string &Row) const
{
    if (Up == Top) GetRow_A(Row, Global);
    else GetRow_B(Row, Global);
}

void DSSAligner::GetRow_A(std::string &Row, bool Global) const
{
    Row.clear();
    const std::string &SeqA = m_ChainA->m_Seq, &SeqB = m_ChainB->m_Seq;
    const uint LA = (uint) SeqA.size(), LB = (uint) SeqB.size();
    if (Global) {
        for (uint i = m_LoA; i < m_LoB; ++i) Row += '.';
        for (uint i = 0; i < m_LoA; ++i) Row += (char) tolower(SeqA[i]);
    }
    uint PosA = m_LoA, PosB = m_LoB;
    for (char c : m_Path) {
        if (c == 'M') { Row += SeqA[PosA++]; ++PosB; }
        else if (c == 'D') Row += SeqA[PosA++];
        else if (c == 'I') { Row += '-'; ++PosB; }
    }
    if (Global) {
        while (PosA < LA) { Row += (char) tolower(SeqA[PosA++]); ++PosB; }
        while (PosB++ < LB) Row += '.';
    }
}

void DSSAligner::GetRow_B(std::string &Row, bool Global) const
{
    Row.clear();
    const std::string &SeqA = m_ChainA->m_Seq, &SeqB = m_ChainB->m_Seq;
    const uint LA = (uint) SeqA.size(), LB = (uint) SeqB.size();
    if (Global) {
        for (uint i = m_LoB; i < m_LoA; ++i) Row += '.';
        for (uint i = 0; i < m_LoB; ++i) Row += (char) tolower(SeqB[i]);
    }
    uint PosA = m_LoA, PosB = m_LoB;
    for (char c : m_Path) {
        if (c == 'M') { ++PosA; Row += SeqB[PosB++]; }
        else if (c == 'D') { ++PosA; Row += '-'; }
        else if (c == 'I') Row += SeqB[PosB++];
    }
    if (Global) {
        while (PosB < LB) { Row += (char) tolower(SeqB[PosB++]); ++PosA; }
        while (PosA++ < LA) Row += '.';
    }
}

// printf("%.1f") and printf("%.3g") of the values hit lines carry (percent identity, coverage, P-value, scores): two
// snprintf calls were a third of the per-row cost of a 90-million-row -verysensitive output.  The fast paths scale the
// value by an exactly representable power of ten (one rounding, relative error 2^-53), so the scaled value is off by
// < 1e-12; whenever it lies within 1e-6 of a rounding boundary -- or the value is outside the range handled --
// snprintf decides.  The text is therefore printf's in every case (rsk_selftest_format compares them at random).
static const double kP10[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20,
                                 1e21, 1e22 };

static void AppendFixed1(std::string &out, double x)                  // "%.1f"
{
    if (x >= 0 && x < 1e5 && !std::signbit(x)) {
        const double s = x * 10.0, fl = floor(s), fr = s - fl;
        if (fabs(fr - 0.5) >= 1e-6) {
            uint32_t n = (uint32_t) fl + (fr > 0.5 ? 1u : 0u);
            char b[16];
            int k = 16;
            b[--k] = (char) ('0' + n % 10); n /= 10;
            b[--k] = '.';
            do { b[--k] = (char) ('0' + n % 10); n /= 10; } while (n);
            out.append(b + k, (size_t) (16 - k));
            return;
        }
    }
    char tmp[64];
    const int k = snprintf(tmp, sizeof(tmp), "%.1f", x);
    out.append(tmp, (size_t) k);
}

static void AppendG3(std::string &out, double x)                      // "%.3g"
{
    if (x >= 1e-20 && x < 1000) {
        // decimal exponent from the binary one: floor(log10 x) is this estimate or one more, which the scaled value
        // below tells apart (no log10 call: it was a quarter of the cost of a default hit line)
        int E = (int) floor(ilogb(x) * 0.30102999566398120);
        if (E >= -20 && E <= 2) {
            double s = x * kP10[2 - E];
            if (s < 100 && E > -20) { --E; s = x * kP10[2 - E]; }
            else if (s >= 1000 && E < 2) { ++E; s = x * kP10[2 - E]; }
            if (s >= 100 && s < 999.4) {
                const double fl = floor(s), fr = s - fl;
                if (fabs(fr - 0.5) >= 1e-6) {
                    const uint32_t d = (uint32_t) fl + (fr > 0.5 ? 1u : 0u);          // 100..999
                    char dg[3] = { (char) ('0' + d / 100), (char) ('0' + d / 10 % 10), (char) ('0' + d % 10) };
                    int nd = 3;
                    while (nd > 1 && dg[nd - 1] == '0') --nd;                         // %g drops trailing zeros
                    char b[32];
                    int k = 0;
                    if (E < -4) {                                                     // d.dde-XX
                        b[k++] = dg[0];
                        if (nd > 1) { b[k++] = '.'; for (int q = 1; q < nd; ++q) b[k++] = dg[q]; }
                        b[k++] = 'e'; b[k++] = '-';
                        const int a = -E;
                        b[k++] = (char) ('0' + a / 10); b[k++] = (char) ('0' + a % 10);
                    } else if (E < 0) {                                               // 0.000ddd
                        b[k++] = '0'; b[k++] = '.';
                        for (int q = 0; q < -E - 1; ++q) b[k++] = '0';
                        for (int q = 0; q < nd; ++q) b[k++] = dg[q];
                    } else {                                                          // E + 1 digits before the point
                        for (int q = 0; q <= E; ++q) b[k++] = dg[q];
                        if (nd > E + 1) { b[k++] = '.'; for (int q = E + 1; q < nd; ++q) b[k++] = dg[q]; }
                    }
                    out.append(b, (size_t) k);
                    return;
                }
            }
        }
    }
    char tmp[64];
    const int k = snprintf(tmp, sizeof(tmp), "%.3g", x);
    out.append(tmp, (size_t) k);
}

// Self-test of the two formatters against snprintf: n values from the distributions hit lines see (ratios of small
// integers times 100, floats of every decade from 1e-25 to 1e4, decimal ties and their neighbours).  Returns the number
// of differing strings (0 expected); the first difference goes to rsk_last_error.
extern "C" uint64_t rsk_selftest_format(uint64_t seed, uint64_t n)
{
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1, bad = 0;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    char t1[64], t2[64];
    std::string a;
    for (uint64_t it = 0; it < n; ++it) {
        double x;
        const uint64_t r = rnd();
        switch (r & 7) {
        case 0: { const uint64_t den = 1 + (rnd() % 2000); x = 100.0 * (double) (rnd() % (den + 1)) / (double) den; break; }        // percent identity
        case 1: x = (double) (float) pow(10.0, -25.0 + 29.0 * (double) (rnd() >> 11) * (1.0 / 9007199254740992.0)); break;        // floats, every decade
        case 2: x = pow(10.0, -25.0 + 29.0 * (double) (rnd() >> 11) * (1.0 / 9007199254740992.0)); break;
        case 3: x = ((double) (rnd() % 20000) + 0.5) / 10.0; break;                                                                // "%.1f" ties
        case 4: x = (double) (float) (((double) (rnd() % 20000) + 0.5) / 10.0); break;
        case 5: { const int e = (int) (rnd() % 24) - 21; x = ((double) (100 + rnd() % 900) + 0.5) * pow(10.0, e - 2); break; }       // "%.3g" ties
        case 6: { const int e = (int) (rnd() % 24) - 21; x = (double) (float) ((double) (100 + rnd() % 900) * pow(10.0, e - 2)); break; }   // powers of ten, 9995...
        default: x = (double) (rnd() % 100001) / 1000.0; if (rnd() & 1) x = nextafter(x, (rnd() & 2) ? 1e9 : -1e9); break;
        }
        a.clear(); AppendFixed1(a, x); snprintf(t1, sizeof(t1), "%.1f", x);
        if (a != t1) { if (!bad) rsk_set_error("rsk_selftest_format: %%.1f of %.17g: \"%s\" vs printf \"%s\"", x, a.c_str(), t1); ++bad; }
        a.clear(); AppendG3(a, x); snprintf(t2, sizeof(t2), "%.3g", x);
        if (a != t2) { if (!bad) rsk_set_error("rsk_selftest_format: %%.3g of %.17g: \"%s\" vs printf \"%s\"", x, a.c_str(), t2); ++bad; }
    }
    return bad;
}

static const char *EvalueToStr(double E, char *buf, size_t n)
{
    if (E > 10) E = 99;
    if (E > 1) snprintf(buf, n, "%.1f", E);
    else if (E > 0.001) snprintf(buf, n, "%.4f", E);
    else snprintf(buf, n, "%.3g", E);
    return buf;
}

// WriteUserField userfields.cpp:45: one field of a hit line, appended to `out` (printf formats as the reference)
void DSSAligner::AppendUserField(std::string &out, USERFIELD UF, bool Up)
{
    char tmp[64];
    std::string s;
    auto u = [&](uint v) {                       // "%u"
        char b[12];
        int k = 12;
        do { b[--k] = (char) ('0' + v % 10); v /= 10; } while (v);
        out.append(b + k, (size_t) (12 - k));
    };
    auto f = [&](const char *fmt, double v) {
        if (fmt[2] == '1') return AppendFixed1(out, v);                   // "%.1f"
        if (fmt[2] == '3') return AppendG3(out, v);                       // "%.3g"
        const int k = snprintf(tmp, sizeof(tmp), fmt, v);
        out.append(tmp, (size_t) k);
    };
    switch (UF) {
    case UF_query: out += GetLabel(Up); break;
    case UF_target: out += GetLabel(!Up); break;
    case UF_evalue: out += EvalueToStr(GetEvalue(Up), tmp, sizeof(tmp)); break;
    case UF_pvalue: f("%.3g", GetPvalue(Up)); break;
    case UF_ql: u(GetQL(Up)); break;
    case UF_tl: u(GetTL(Up)); break;
    case UF_qlo: u(GetLo(Up) + 1); break;
    case UF_qhi: u(GetHi(Up) + 1); break;
    case UF_tlo: u(GetLo(!Up) + 1); break;
    case UF_thi: u(GetHi(!Up) + 1); break;
    case UF_qcovpct: f("%.1f", GetQCovPct(Up)); break;
    case UF_tcovpct: f("%.1f", GetTCovPct(Up)); break;
    case UF_pctid: f("%.1f", GetPctId()); break;
    case UF_ts: f("%.3g", GetTestStatistic(Up)); break;
    case UF_newts: f("%.3g", GetNewTestStatistic(Up)); break;
    case UF_raw: f("%.3g", m_AlnFwdScore); break;
    case UF_ids: u(m_Ids); break;
    case UF_gaps: u(m_Gaps); break;
    case UF_cigar: PathToCIGAR(m_Path.c_str(), s, Up); out += s; break;
    case UF_qrow: GetRow(Up, true, false, s); out += s; break;
    case UF_trow: GetRow(Up, false, false, s); out += s; break;
    case UF_qrowg: GetRow(Up, true, true, s); out += s; break;
    case UF_trowg: GetRow(Up, false, true, s); out += s; break;
    case UF_dpscore: f("%.4g", m_AlnFwdScore); break;
    case UF_lddt: f("%.4g", m_LDDT != FLT_MAX ? m_LDDT : GetLDDT()); break;
    case UF_aq: f("%.4f", GetAQ(Up)); break;
    case UF_muhsp: { const int k = snprintf(tmp, sizeof(tmp), "%d", m_MKF.m_BestHSPScore); out.append(tmp, (size_t) k); break; }
    case UF_muchain: { const int k = snprintf(tmp, sizeof(tmp), "%d", m_MKF.m_BestChainScore); out.append(tmp, (size_t) k); break; }
    default: out += '?'; break;        // gscore / muscore belong to commands outside -search
    }
}

void DSSAligner::WriteUserField(FILE *f, USERFIELD UF, bool Up)
{
    if (f == nullptr) return;
    std::string s;
    AppendUserField(s, UF, Up);
    fputs(s.c_str(), f);
}

// ToTsv dssaligner.cpp:1016: one hit line appended to `out`
void DSSAligner::AppendTsv(std::string &out, bool Up)
{
    for (size_t i = 0; i < m_UFs.size(); ++i) {
        if (i > 0) out += '\t';
        AppendUserField(out, m_UFs[i], Up);
    }
    out += '\n';
}

void DSSAligner::ToTsv(FILE *f, bool Up, bool NoSelf)
{
    if (f == nullptr) return;
    if (NoSelf && m_ChainA->m_Label == m_ChainB->m_Label) return;
    std::string line;
    AppendTsv(line, Up);
    std::lock_guard<std::mutex> g(m_OutputLock);
    fwrite(line.data(), 1, line.size(), f);
}

}   // namespace reseek_amd

// ---------------------------------------------------------------------------------------------
// C-ABI: MergeFwdBwd (mergefwdback.cpp:6) on explicit paths, the companion of rsk_xdrop_fwd / rsk_xdrop_bwd (k_xdrop.hip)
// ---------------------------------------------------------------------------------------------
void rsk_set_error(const char *fmt, ...);

extern "C" int rsk_merge_fwd_bwd(uint32_t LA, uint32_t LB, uint32_t fwd_lo_a, uint32_t fwd_lo_b, const char *fwd_path, uint32_t bwd_hi_a,
                                 uint32_t bwd_hi_b, const char *bwd_path, uint32_t *lo_a, uint32_t *lo_b, uint32_t *hi_a, uint32_t *hi_b,
                                 char *path, size_t path_cap, uint32_t *path_len)
{
    (void) LA; (void) LB;
    if (!fwd_path || !bwd_path || !lo_a || !lo_b || !hi_a || !hi_b) { rsk_set_error("rsk_merge_fwd_bwd: NULL argument"); return RSK_E_INVALID; }
    if (!*fwd_path && !*bwd_path) { rsk_set_error("rsk_merge_fwd_bwd: both paths are empty (mergefwdback.cpp:11 asserts)"); return RSK_E_INVALID; }
    std::string Joined;
    unsigned la, lb, ha, hb;
    reseek_amd::JoinExtensions(fwd_lo_a, fwd_lo_b, fwd_path, bwd_hi_a, bwd_hi_b, bwd_path, la, lb, ha, hb, Joined);
    *lo_a = la; *lo_b = lb; *hi_a = ha; *hi_b = hb;
    if (path_len) *path_len = (uint32_t) Joined.size();
    if (Joined.size() + 1 > path_cap || !path) { rsk_set_error("rsk_merge_fwd_bwd: path buffer too small (%zu needed)", Joined.size() + 1); return RSK_E_INVALID; }
    memcpy(path, Joined.c_str(), Joined.size() + 1);
    return RSK_OK;
}

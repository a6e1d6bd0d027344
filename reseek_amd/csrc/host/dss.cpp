// dss.cpp -- host mirror of the per-chain featurisation of the reference:
//   DSS           dss.h:15, dss.cpp (NEN/REN neighbours, densities, SSE distances, Mu letters, profile),
//                 myss.cpp:125-205 (Conf letters), valuetoint.cpp (bins), getss.cpp:33 (secondary structure)
//   PDBChain      pdbchain.cpp:310 GetDist, :478 GetReverse, pdbchain.h:89-90 coordinate quantisation
//   BCAData       bcadata.cpp:60-117,191-234 (.bca container reader)
// The eight profile features are AA, NENDist, Conf, NENConf, RENDist, DstNxtHlx, StrandDens, NormDens
// (rsk_feature_name, namedparams.cpp:36-43); the Mu letter is SS3 + 3*NENSS3 + 9*RENDist4
// (dssparams.cpp:7-14).  All arithmetic follows the reference's types: CA-CA distances are
// (float) sqrt(double), everything downstream is double, exp() is libm's.
// Per-chain work is O(L * window) and independent per chain; it stays on the host cores as in the
// reference (ProfileLoader threads), the byte outputs are pinned by the dumped profiles of q10/q100/palms.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <thread>

#include "../rsk_tables_data.h"
#include "dss_data.h"
#include "reseek_host.h"

namespace reseek_amd {

// pdbchain.cpp:310-318 -> the FLOAT overload of GetDist3D (abcxyz.h:116-126): differences, squares, their sum and the
// square root are all single precision.  (Computing in double and rounding once differs in the last bit for about one
// residue pair in a million, enough to flip a nearest-neighbour tie: found by comparing 3000 synthetic chains with the
// reference, the q10 / q100 / palms fixtures do not contain such a pair.)
float PDBChain::GetDist(uint Pos1, uint Pos2) const
{
    const float dx = m_Xs[Pos1] - m_Xs[Pos2];
    const float dy = m_Ys[Pos1] - m_Ys[Pos2];
    const float dz = m_Zs[Pos1] - m_Zs[Pos2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    return sqrtf(d2);
}

void PDBChain::GetReverse(PDBChain &Rev) const                       // pdbchain.cpp:470-483
{
    Rev = *this;
    std::reverse(Rev.m_Seq.begin(), Rev.m_Seq.end());
    std::reverse(Rev.m_Xs.begin(), Rev.m_Xs.end());
    std::reverse(Rev.m_Ys.begin(), Rev.m_Ys.end());
    std::reverse(Rev.m_Zs.begin(), Rev.m_Zs.end());
    Rev.m_Label += ".rev";
}

// getss.cpp:6-31 (after sec_str() of TM-align): helix / strand / turn / loop from five CA-CA distances
static char SSChar(double d13, double d14, double d15, double d24, double d25, double d35)
{
    const double DH = 2.1;
    if (fabs(d15 - 6.37) < DH && fabs(d14 - 5.18) < DH && fabs(d25 - 5.18) < DH && fabs(d13 - 5.45) < DH && fabs(d24 - 5.45) < DH &&
        fabs(d35 - 5.45) < DH)
        return 'h';
    const double DS = 1.42;
    if (fabs(d15 - 13) < DS && fabs(d14 - 10.4) < DS && fabs(d25 - 10.4) < DS && fabs(d13 - 6.1) < DS && fabs(d24 - 6.1) < DS &&
        fabs(d35 - 6.1) < DS)
        return 's';
    if (d15 < 8.2) return 't';
    return '~';
}

void PDBChain::GetSS(std::string &SS) const                          // getss.cpp:33-60
{
    SS.clear();
    const uint L = GetSeqLength();
    for (uint Pos = 0; Pos < L; ++Pos) {
        if (Pos < 2 || Pos + 2 >= L) { SS += '~'; continue; }
        const double d13 = GetDist(Pos - 2, Pos), d14 = GetDist(Pos - 2, Pos + 1), d15 = GetDist(Pos - 2, Pos + 2);
        const double d24 = GetDist(Pos - 1, Pos + 1), d25 = GetDist(Pos - 1, Pos + 2), d35 = GetDist(Pos, Pos + 2);
        SS += SSChar(d13, d14, d15, d24, d25, d35);
    }
}

void DSS::Init(const PDBChain &Chain)
{
    m_Chain = &Chain;
    m_SS.clear();
    m_NENs.clear();
    m_RENs.clear();
    m_Density_ScaledValues.clear();
    m_DistFactors.clear();
    m_ConfLetters.clear();
    m_DensityValues.clear();
    m_StrandDensValues.clear();
    m_SSE_Mids.clear();
    m_SSE_cs.clear();
    m_SSEsDone = false;
}

// Init for the reversed copy of the chain `Fwd` is initialised on (GetSelfRevScore alignpair.cpp:7-24 featurises
// it).  dist(i, j) of the reversed chain is dist(L-1-i, L-1-j) of the chain bit for bit (the coordinate differences
// only change sign), so its exp() table is the mirrored table of the chain: no second round of exp() calls, which
// are two thirds of the featurisation time.  Everything else (nearest neighbours and their tie-breaks, SS, the
// sums in ascending position) is computed on the reversed chain as usual.
void DSS::InitReversed(const PDBChain &Rev, DSS &Fwd)
{
    Fwd.SetDistFactors();
    Init(Rev);
    const uint L = GetSeqLength();
    if (Fwd.GetSeqLength() != L) throw std::runtime_error("DSS::InitReversed: chain lengths differ");
    const int W = Fwd.m_DistFactorW;
    m_DistFactorW = W;
    m_DistFactors.assign((size_t) L * W + 1, 0.0);
    for (uint Pos = 0; Pos < L; ++Pos)
        for (int k = 1; k <= W && Pos + k < L; ++k)
            m_DistFactors[(size_t) Pos * W + (k - 1)] = Fwd.m_DistFactors[(size_t) (L - 1 - Pos - k) * W + (k - 1)];
}

static uint Bin(const double *T, double Value)                       // valuetoint.cpp: "if (Value < t_k) return k"
{
    for (uint k = 0; k < 15; ++k)
        if (Value < T[k]) return k;
    return 15;
}

void DSS::SetSS()
{
    if (m_SS.empty()) m_Chain->GetSS(m_SS);
}

// First index k in [0, n) whose sqrtf(d2[k]) is the smallest (what the running "Dist < MinDist" loops of the reference's CalcNEN /
// CalcREN, dss.cpp:374-440, keep), or -1 when the range is empty or nothing is below the initial MinDist of 999.  sqrtf is monotone, so
// the smallest distance is sqrtf(min d2); several different d2 can round to that same float, they all lie within a
// few ulp of the minimum: candidates are d2 <= min * (1 + 1e-6), each confirmed with its own sqrtf.
static int FirstNearest(const float *d2, int n)
{
    if (n <= 0) return -1;
    float m8[8];
    for (int l = 0; l < 8; ++l) m8[l] = FLT_MAX;
    int k = 0;
    for (; k + 8 <= n; k += 8)
        for (int l = 0; l < 8; ++l) m8[l] = d2[k + l] < m8[l] ? d2[k + l] : m8[l];
    float m2 = FLT_MAX;
    for (; k < n; ++k) m2 = d2[k] < m2 ? d2[k] : m2;
    for (int l = 0; l < 8; ++l) m2 = m8[l] < m2 ? m8[l] : m2;
    if (m2 == FLT_MAX) return -1;                                    // every position excluded
    const float s = sqrtf(m2);
    if (!((double) s < 999)) return -1;
    const float thr = m2 + m2 * 1e-6f;
    for (k = 0; k + 8 <= n; k += 8) {
        int any = 0;
        for (int l = 0; l < 8; ++l) any |= d2[k + l] <= thr;
        if (any) break;
    }
    for (; k < n; ++k)
        if (d2[k] <= thr && sqrtf(d2[k]) == s) return k;
    return -1;                                                        // not reached: the minimum itself qualifies
}

// dss.cpp:374-440 (CalcNEN: nearest residue within +-100 with |offset| > 12; CalcREN: the nearest on the other side) for every position: the
// squared distances of the +-100 window are computed once per position as a vectorisable loop (same float
// operations per element as GetDist, minus the square root), NEN is the nearest of the whole window, REN the
// nearest of the part of it on the other side of Pos.
void DSS::SetNENs()
{
    if (!m_NENs.empty()) return;
    const uint L = GetSeqLength();
    const float *X = m_Chain->m_Xs.data(), *Y = m_Chain->m_Ys.data(), *Z = m_Chain->m_Zs.data();
    std::vector<float> buf((size_t) 2 * m_NEN_W + 16);
    float *d2 = buf.data();
    m_NENs.reserve(L);
    m_RENs.reserve(L);
    for (uint Pos = 0; Pos < L; ++Pos) {
        const int lo = std::max(0, (int) Pos - m_NEN_W), hi = std::min((int) L - 1, (int) Pos + m_NEN_W), n = hi - lo + 1;
        const float x = X[Pos], y = Y[Pos], z = Z[Pos];
        for (int k = 0; k < n; ++k) {
            const float dx = x - X[lo + k], dy = y - Y[lo + k], dz = z - Z[lo + k];
            d2[k] = dx * dx + dy * dy + dz * dz;
        }
        // |offset| <= m_NEN_w is skipped by both searches
        const int xlo = std::max(lo, (int) Pos - m_NEN_w), xhi = std::min(hi, (int) Pos + m_NEN_w);
        for (int q = xlo; q <= xhi; ++q) d2[q - lo] = FLT_MAX;
        const int kn = FirstNearest(d2, n);
        if (kn < 0) { m_NENs.push_back(UINT_MAX); m_RENs.push_back(UINT_MAX); continue; }
        const uint NEN = (uint) (lo + kn);
        m_NENs.push_back(NEN);
        int kr;
        if (NEN > Pos) kr = FirstNearest(d2, (int) Pos - lo);                                   // [lo, Pos - 1]
        else { kr = FirstNearest(d2 + (Pos + 1 - lo), hi - (int) Pos); if (kr >= 0) kr += Pos + 1 - lo; }   // [Pos + 1, hi]
        m_RENs.push_back(kr < 0 ? UINT_MAX : (uint) (lo + kr));
    }
}

// exp(-dist(i, i + k) / radius) for k = 1..W, computed once per chain: the two density features (dss.cpp:217-244,
// 339-372) use the same radius and window and dist is symmetric, so each unordered residue pair needs one exp()
// instead of four.  The features still add the factors in the reference's order (ascending Pos2).
void DSS::SetDistFactors()
{
    if (!m_DistFactors.empty()) return;
    const uint L = GetSeqLength();
    const int W = std::max(m_Density_W, m_SSDensity_W), w = std::min(m_Density_w, m_SSDensity_w);
    m_DistFactorW = W;
    m_DistFactors.assign((size_t) L * W + 1, 0.0);
    for (uint Pos = 0; Pos < L; ++Pos)
        for (int k = w + 1; k <= W && Pos + k < L; ++k)
            m_DistFactors[(size_t) Pos * W + (k - 1)] = exp(-(double) m_Chain->GetDist(Pos, Pos + k) / m_Density_Radius);
}

double DSS::GetDensity(uint Pos)                                      // dss.cpp:217-244
{
    SetDensities();
    return m_DensityValues[Pos];
}

// GetDensity (dss.cpp:217-244) and GetSSDensity(Pos, 's') (dss.cpp:339-372) of every position in one pass over the
// factor table.  Each of the three sums (density; SS density total and its strand part) adds its terms in ascending
// Pos2 exactly as the reference's loops do -- the sums are separate dependency chains, so running them side by side
// hides the latency of the double adds that a single chain is bound by.
void DSS::SetDensities()
{
    if (!m_DensityValues.empty()) return;
    SetSS();
    SetDistFactors();
    const uint L = GetSeqLength();
    m_DensityValues.assign(L, DBL_MAX);
    m_StrandDensValues.assign(L, DBL_MAX);
    if (m_Density_W != m_SSDensity_W || m_Density_w > m_SSDensity_w) throw std::runtime_error("DSS::SetDensities: unexpected windows");
    const int W = m_Density_W, w1 = m_Density_w, w2 = m_SSDensity_w, TW = m_DistFactorW;
    const double *T = m_DistFactors.data();
    const char *SS = m_SS.data();
    for (int Pos = 1; Pos + 1 < (int) L; ++Pos) {
        const int lo = std::max(0, Pos - W), hi = std::min((int) L - 1, Pos + W);
        double D1 = 0, D2 = 0, Dc = 0;
        int q = lo;
        for (; q < Pos - w2; ++q) {                                   // left of both exclusion zones
            const double F = T[(size_t) q * TW + (Pos - q - 1)];
            D1 += F; D2 += F;
            if (SS[q] == 's') Dc += F;
        }
        for (; q < Pos - w1; ++q) D1 += T[(size_t) q * TW + (Pos - q - 1)];
        const double *R = T + (size_t) Pos * TW - Pos - 1;            // R[q] = factor(Pos, q) for q > Pos
        for (q = Pos + w1 + 1; q <= std::min(hi, Pos + w2); ++q) D1 += R[q];
        for (; q <= hi; ++q) {
            const double F = R[q];
            D1 += F; D2 += F;
            if (SS[q] == 's') Dc += F;
        }
        m_DensityValues[Pos] = D1;
        m_StrandDensValues[Pos] = Dc / (D2 + m_SSDensity_epsilon);
    }
}

bool DSS::UseDeviceDensities(const double *Dens, const double *StrandDens)
{
    const uint L = GetSeqLength();
    const double Margin = 1e-9;                                       // device-vs-libm exp: the values differ by ~1e-14
    auto near_a_boundary = [&](const double *T, double v) {
        for (uint k = 0; k < 15; ++k)
            if (fabs(v - T[k]) <= Margin) return true;
        return false;
    };
    // feature 6 (dss.cpp:339-372) bins the strand density itself
    for (uint Pos = 0; Pos < L; ++Pos)
        if (StrandDens[Pos] != DBL_MAX && near_a_boundary(rsk_bins_StrandDens, StrandDens[Pos])) return false;
    // feature 7 bins (D - min) / max(range, 1) (SetDensity_ScaledValues, dss.cpp:179-215)
    double MinValue = 999, MaxValue = 0;
    for (uint Pos = 0; Pos < L; ++Pos)
        if (Dens[Pos] != DBL_MAX) { MinValue = std::min(MinValue, Dens[Pos]); MaxValue = std::max(MaxValue, Dens[Pos]); }
    double Range = MaxValue - MinValue;
    if (fabs(Range - 1) <= Margin) return false;                      // which side of "Range < 1" is not certain
    if (Range < 1) Range = 1;
    for (uint Pos = 0; Pos < L; ++Pos)
        if (Dens[Pos] != DBL_MAX && near_a_boundary(rsk_bins_NormDens, (Dens[Pos] - MinValue) / Range)) return false;
    m_DensityValues.assign(Dens, Dens + L);
    m_StrandDensValues.assign(StrandDens, StrandDens + L);
    return true;
}

void DSS::SetDensity_ScaledValues()                                   // dss.cpp:179-215
{
    if (!m_Density_ScaledValues.empty()) return;
    const uint L = GetSeqLength();
    std::vector<double> Values;
    double MinValue = 999, MaxValue = 0;
    for (uint Pos = 0; Pos < L; ++Pos) {
        const double D = GetDensity(Pos);
        Values.push_back(D);
        if (D != DBL_MAX) { MinValue = std::min(MinValue, D); MaxValue = std::max(MaxValue, D); }
    }
    double Range = MaxValue - MinValue;
    if (Range < 1) Range = 1;
    for (uint Pos = 0; Pos < L; ++Pos) {
        const double Value = Values[Pos];
        m_Density_ScaledValues.push_back(Value == DBL_MAX ? DBL_MAX : (Value - MinValue) / Range);
    }
}

double DSS::GetSSDensity(uint Pos, char c)                            // dss.cpp:339-372
{
    if (c == 's') { SetDensities(); return m_StrandDensValues[Pos]; }
    SetSS();
    SetDistFactors();
    const uint L = GetSeqLength();
    if (Pos == 0 || Pos + 1 >= L) return DBL_MAX;
    int iLo = (int) Pos - m_SSDensity_W;
    if (iLo < 0) iLo = 0;
    int iHi = (int) Pos + m_SSDensity_W;
    if (iHi >= (int) L) iHi = (int) L - 1;
    double D = 0, Dc = 0;
    for (uint Pos2 = (uint) iLo; Pos2 <= (uint) iHi; ++Pos2) {
        if (Pos2 + m_SSDensity_w >= Pos && Pos2 <= Pos + m_SSDensity_w) continue;
        const double Factor = DistFactor(Pos, Pos2);
        D += Factor;
        if (m_SS[Pos2] == c) Dc += Factor;
    }
    return Dc / (D + m_SSDensity_epsilon);
}

void DSS::SetSSEs()                                                   // dss.cpp:78-155: helix/strand runs of >= 8, their midpoints
{
    if (!m_SSE_Mids.empty() || m_SSEsDone) return;
    m_SSEsDone = true;
    SetSS();
    const uint L = GetSeqLength();
    if (L == 0) return;
    char currc = m_SS[0];
    uint StartPos = 0, RunLength = 1;
    for (uint Pos = 1; Pos <= L; ++Pos) {
        const char here = Pos == L ? 0 : m_SS[Pos];                   // std::string[size()] is '\0' in the reference
        if (here == currc) ++RunLength;
        else {
            if (RunLength >= m_SSE_MinLength && (currc == 'h' || currc == 's')) {
                m_SSE_Mids.push_back(StartPos + RunLength / 2);
                m_SSE_cs.push_back(currc);
            }
            currc = here;
            StartPos = Pos;
            RunLength = 1;
        }
    }
}

double DSS::GetFloat_DstNxtHlx(uint Pos)                              // dss.cpp:866-881
{
    SetSSEs();
    for (size_t i = 0; i < m_SSE_Mids.size(); ++i) {
        if (m_SSE_cs[i] != 'h') continue;
        const uint Mid = m_SSE_Mids[i];
        if (Mid <= Pos + m_SSE_Margin) continue;
        return m_Chain->GetDist(Pos, Mid);
    }
    return 0;
}

// myss.cpp:125-160: nine CA-CA distances around Pos, nearest of the 16 cluster centres (first wins ties)
uint DSS::ConfLetter(uint Pos) const
{
    const uint L = GetSeqLength();
    if (Pos < 3 || Pos + 3 >= L) return UINT_MAX;
    static const int iv[9] = { -2, -2, -2, -1, -1, 0, -3, 0, -3 }, jv[9] = { 0, 1, 2, 1, 2, 2, 3, 3, 0 };
    double v[9];
    for (int m = 0; m < 9; ++m) v[m] = m_Chain->GetDist(Pos + iv[m], Pos + jv[m]);
    double MinDist = DBL_MAX;
    uint Best = 0;
    for (uint k = 0; k < 16; ++k) {
        double Sum2 = 0;
        for (int m = 0; m < 9; ++m) { const double diff = v[m] - rsk_conf_means[k][m]; Sum2 += diff * diff; }
        const double d = sqrt(Sum2);
        if (k == 0 || d < MinDist) { Best = k; MinDist = d; }
    }
    return Best;
}

void DSS::SetConfLetters()                                            // Conf of Pos and NENConf of its neighbours read the same letter
{
    if (!m_ConfLetters.empty()) return;
    const uint L = GetSeqLength();
    m_ConfLetters.resize(L);
    for (uint Pos = 0; Pos < L; ++Pos) m_ConfLetters[Pos] = ConfLetter(Pos);
}

static uint SS3(char c)                                              // dss.cpp:64-76: h 0, s 1, t 2, ~ 2, else WILDCARD (0)
{
    switch (c) {
    case 'h': return 0;
    case 's': return 1;
    case 't': return 2;
    case '~': return 2;
    }
    return 0;
}

uint DSS::GetFeature(uint FeatureIndex, uint Pos)                     // dss.cpp:808-838 for the eight profile features
{
    switch (FeatureIndex) {
    case 0: {                                                         // AA
        const uint Letter = rsk_aa_letter[(unsigned char) m_Chain->m_Seq[Pos]];
        return Letter >= 20 ? 0 : Letter;
    }
    case 1: {                                                         // NENDist dss.cpp:496-503
        SetNENs();
        const uint NEN = m_NENs[Pos];
        const double d = NEN == UINT_MAX ? m_DefaultNENDist : (double) m_Chain->GetDist(Pos, NEN);
        return Bin(rsk_bins_NENDist, d);
    }
    case 2: {                                                         // Conf myss.cpp:162-170
        SetConfLetters();
        const uint c = m_ConfLetters[Pos];
        return c == UINT_MAX ? 0 : c;
    }
    case 3: {                                                         // NENConf myss.cpp:172-188
        SetNENs();
        const uint NEN = m_NENs[Pos];
        if (NEN == UINT_MAX) return 0;
        SetConfLetters();
        const uint c = m_ConfLetters[NEN];
        return c == UINT_MAX ? 0 : c;
    }
    case 4: {                                                         // RENDist dss.cpp:521-528
        SetNENs();
        const uint REN = m_RENs[Pos];
        const double d = REN == UINT_MAX ? m_DefaultNENDist : (double) m_Chain->GetDist(Pos, REN);
        return Bin(rsk_bins_RENDist, d);
    }
    case 5: return Bin(rsk_bins_DstNxtHlx, GetFloat_DstNxtHlx(Pos));
    case 6: return Bin(rsk_bins_StrandDens, GetSSDensity(Pos, 's'));
    case 7: SetDensity_ScaledValues(); return Bin(rsk_bins_NormDens, m_Density_ScaledValues[Pos]);
    }
    throw std::runtime_error("DSS::GetFeature: unknown feature");
}

void DSS::GetProfile(std::vector<std::vector<byte> > &Profile)        // dss.cpp:716-741
{
    const uint L = GetSeqLength();
    Profile.assign(RSK_NFEATURES, std::vector<byte>());
    for (uint f = 0; f < RSK_NFEATURES; ++f) {
        Profile[f].reserve(L);
        for (uint Pos = 0; Pos < L; ++Pos) Profile[f].push_back((byte) GetFeature(f, Pos));
    }
}

void DSS::GetMuLetters(std::vector<byte> &Letters)                    // dss.cpp:629-644,700-714
{
    SetSS();
    SetNENs();
    const uint L = GetSeqLength();
    Letters.clear();
    Letters.reserve(L);
    for (uint Pos = 0; Pos < L; ++Pos) {
        const uint ss3 = SS3(m_SS[Pos]);
        const uint NEN = m_NENs[Pos];
        const uint nenss3 = NEN == UINT_MAX ? 0 : SS3(m_SS[NEN]);
        const uint rendist4 = GetFeature(4, Pos) / 4;                 // dss.cpp:548-556
        Letters.push_back((byte) (ss3 + 3 * nenss3 + 9 * rendist4));
    }
}

void DSS::GetMuKmers(const std::vector<byte> &Letters, std::vector<uint> &Kmers, const std::string &PatternStr)   // dss.cpp:659-682
{
    Kmers.clear();
    const size_t PL = PatternStr.size(), L = Letters.size();
    for (size_t Pos = 0; Pos + PL <= L; ++Pos) {
        uint Kmer = 0;
        for (size_t j = 0; j < PL; ++j)
            if (PatternStr[j] == '1') Kmer = Kmer * 36 + Letters[Pos + j];
        Kmers.push_back(Kmer);
    }
}

// ---------------------------------------------------------------------------------------------
// .bca container (bcadata.cpp): u32 magic 0x00BCABCA? (BCA_MAGIC bcadata.h), u64 chains, u64 offset of the
// u32 length array, u64 label bytes; per chain L amino characters + 3L u16 coordinates (x,y,z interleaved,
// coord = IC/10.0f - 1000, pdbchain.h:90); then u32 L[n]; then the NUL-terminated labels.
// ---------------------------------------------------------------------------------------------
void BCAData::Open(const std::string &FN)
{
    Close();
    m_f = fopen(FN.c_str(), "rb");
    if (!m_f) throw std::runtime_error("BCAData::Open: cannot open " + FN);
    auto rd = [&](void *p, size_t n) { if (n && fread(p, 1, n, m_f) != n) throw std::runtime_error("BCAData::Open: truncated " + FN); };
    uint32_t Magic;
    rd(&Magic, 4);
    if (Magic != BCA_MAGIC) throw std::runtime_error("BCAData::Open: bad magic, not a .bca file: " + FN);
    uint64_t ChainCount, SeqLengthsPos, LabelDataSize;
    rd(&ChainCount, 8); rd(&SeqLengthsPos, 8); rd(&LabelDataSize, 8);
    uint64_t Offset = 4 + 3 * 8;
    if (ChainCount > 0xFFFFFFFFull) throw std::runtime_error("BCAData::Open: too many chains");
    m_SeqLengths.resize((size_t) ChainCount);
    if (fseeko(m_f, (off_t) SeqLengthsPos, SEEK_SET) != 0) throw std::runtime_error("BCAData::Open: seek failed");
    rd(m_SeqLengths.data(), 4 * (size_t) ChainCount);
    m_Offsets.clear();
    for (uint64_t i = 0; i < ChainCount; ++i) { m_Offsets.push_back(Offset); Offset += 7ull * m_SeqLengths[i]; }
    std::vector<char> LabelData((size_t) LabelDataSize + 1, 0);
    rd(LabelData.data(), (size_t) LabelDataSize);
    m_Labels.clear();
    if (ChainCount) {
        m_Labels.push_back(LabelData.data());
        for (uint64_t i = 0; i + 1 < LabelDataSize; ++i)
            if (LabelData[i] == 0) m_Labels.push_back(LabelData.data() + i + 1);
    }
    if (m_Labels.size() != ChainCount) throw std::runtime_error("BCAData::Open: label count does not match the chain count");
}

void BCAData::Close()
{
    if (m_f && m_Writing) {
        // bcadata.cpp:140-168: length array, labels, then the three header fields
        const uint64_t ChainCount = m_SeqLengths.size();
        const uint64_t SeqLengthsPos = (uint64_t) ftello(m_f);
        fwrite(m_SeqLengths.data(), 4, (size_t) ChainCount, m_f);
        uint64_t LabelDataSize = 0;
        for (const std::string &Label : m_Labels) { fwrite(Label.c_str(), 1, Label.size() + 1, m_f); LabelDataSize += Label.size() + 1; }
        fseeko(m_f, 4, SEEK_SET);
        fwrite(&ChainCount, 8, 1, m_f);
        fwrite(&SeqLengthsPos, 8, 1, m_f);
        fwrite(&LabelDataSize, 8, 1, m_f);
    }
    if (m_f) fclose(m_f);
    m_f = nullptr;
    m_Writing = false;
    m_Labels.clear(); m_Offsets.clear(); m_SeqLengths.clear();
}

void BCAData::Create(const std::string &FN)
{
    Close();
    m_f = fopen(FN.c_str(), "wb");
    if (!m_f) throw std::runtime_error("BCAData::Create: cannot create " + FN);
    const uint32_t Magic = BCA_MAGIC;
    const uint64_t Placeholder = 0;
    fwrite(&Magic, 4, 1, m_f);
    for (int k = 0; k < 3; ++k) fwrite(&Placeholder, 8, 1, m_f);
    m_Writing = true;
}

void BCAData::WriteChain(const PDBChain &Chain)
{
    if (!m_f || !m_Writing) throw std::runtime_error("BCAData::WriteChain: not open for writing");
    const uint L = Chain.GetSeqLength();
    m_Labels.push_back(Chain.m_Label);
    m_SeqLengths.push_back(L);
    std::vector<uint16_t> ICs(3 * (size_t) L);
    for (uint i = 0; i < L; ++i) {                                   // PDBChain::CoordToIC pdbchain.h:89: uint16((X + 1000)*10 + 0.5)
        ICs[3 * i] = uint16_t((Chain.m_Xs[i] + 1000) * 10 + 0.5);
        ICs[3 * i + 1] = uint16_t((Chain.m_Ys[i] + 1000) * 10 + 0.5);
        ICs[3 * i + 2] = uint16_t((Chain.m_Zs[i] + 1000) * 10 + 0.5);
    }
    fwrite(Chain.m_Seq.data(), 1, L, m_f);
    fwrite(ICs.data(), 2, 3 * (size_t) L, m_f);
}

void ReadMuFasta(const std::string &FN, std::vector<std::string> &Labels, std::vector<std::vector<byte> > &Seqs)
{
    FILE *f = fopen(FN.c_str(), "r");
    if (!f) throw std::runtime_error("ReadMuFasta: cannot open " + FN);
    byte lut[256];
    memset(lut, 0xFF, sizeof(lut));
    static const char MuChars[] = "ABCDEFGHIJLKMNOPQRSTUVWXYZabcdefghij";   // g_LetterToCharMu alpha.cpp:3550 (L before K)
    for (int i = 0; i < 36; ++i) lut[(unsigned char) MuChars[i]] = (byte) i;
    Labels.clear();
    Seqs.clear();
    std::vector<char> buf(1 << 16);
    std::string line;
    for (;;) {
        if (!fgets(buf.data(), (int) buf.size(), f)) break;
        line = buf.data();
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') { Labels.push_back(line.substr(1)); Seqs.emplace_back(); continue; }
        if (Seqs.empty()) { fclose(f); throw std::runtime_error("ReadMuFasta: sequence data before the first label"); }
        for (char c : line) {
            const byte l = lut[(unsigned char) c];
            if (l == 0xFF) { fclose(f); throw std::runtime_error("ReadMuFasta: invalid Mu character"); }
            Seqs.back().push_back(l);
        }
    }
    fclose(f);
}

void BCAData::ReadChain(uint64_t ChainIdx, PDBChain &Chain)
{
    if (!m_f || ChainIdx >= m_SeqLengths.size()) throw std::runtime_error("BCAData::ReadChain: bad chain index");
    const uint L = m_SeqLengths[ChainIdx];
    std::vector<uint16_t> ICs(3 * (size_t) L);
    Chain.m_Seq.assign(L, ' ');
    {
        std::lock_guard<std::mutex> g(m_ReadLock);
        if (fseeko(m_f, (off_t) m_Offsets[ChainIdx], SEEK_SET) != 0) throw std::runtime_error("BCAData::ReadChain: seek failed");
        if (L && (fread(&Chain.m_Seq[0], 1, L, m_f) != L || fread(ICs.data(), 2, 3 * (size_t) L, m_f) != 3 * (size_t) L))
            throw std::runtime_error("BCAData::ReadChain: truncated file");
    }
    // the reference builds the sequence with string(char*): it stops at an embedded NUL (bcadata.cpp:213)
    const size_t z = Chain.m_Seq.find('\0');
    if (z != std::string::npos) Chain.m_Seq.resize(z);
    Chain.m_Xs.resize(L); Chain.m_Ys.resize(L); Chain.m_Zs.resize(L);
    for (uint i = 0; i < L; ++i) {
        Chain.m_Xs[i] = float(ICs[3 * i] / 10.0f) - 1000;
        Chain.m_Ys[i] = float(ICs[3 * i + 1] / 10.0f) - 1000;
        Chain.m_Zs[i] = float(ICs[3 * i + 2] / 10.0f) - 1000;
    }
    Chain.m_Label = m_Labels[ChainIdx];
}

}   // namespace reseek_amd

// ---------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------
using namespace reseek_amd;
void rsk_set_error(const char *fmt, ...);

extern "C" int rsk_dss_featurize(const char *seq, const float *x, const float *y, const float *z, uint32_t L, uint8_t *prof, uint8_t *mu)
{
    if (!seq || !x || !y || !z || (!prof && !mu)) { rsk_set_error("rsk_dss_featurize: NULL argument"); return RSK_E_INVALID; }
    try {
        PDBChain C;
        C.m_Seq.assign(seq, seq + L);
        C.m_Xs.assign(x, x + L); C.m_Ys.assign(y, y + L); C.m_Zs.assign(z, z + L);
        DSS D;
        D.Init(C);
        if (prof) {
            std::vector<std::vector<byte> > P;
            D.GetProfile(P);
            for (int f = 0; f < RSK_NFEAT; ++f) memcpy(prof + (size_t) f * L, P[f].data(), L);
        }
        if (mu) {
            std::vector<byte> M;
            D.GetMuLetters(M);
            memcpy(mu, M.data(), L);
        }
    } catch (const std::exception &e) {
        rsk_set_error("rsk_dss_featurize: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

// The host's (glibc libm exp) values of the two density features of one chain -- what DSS::UseDeviceDensities' margin is
// measured against (tests/test_gpu_dss_density.py).  dens / sdens: L doubles each, DBL_MAX = no value.
extern "C" int rsk_dss_densities_host(const float *x, const float *y, const float *z, uint32_t L, double *dens, double *sdens)
{
    if (!x || !y || !z || !dens || !sdens) { rsk_set_error("rsk_dss_densities_host: NULL argument"); return RSK_E_INVALID; }
    try {
        PDBChain C;
        C.m_Seq.assign(L, 'A');
        C.m_Xs.assign(x, x + L); C.m_Ys.assign(y, y + L); C.m_Zs.assign(z, z + L);
        DSS D;
        D.Init(C);
        for (uint32_t Pos = 0; Pos < L; ++Pos) { dens[Pos] = D.GetDensity(Pos); sdens[Pos] = D.GetSSDensity(Pos, 's'); }
    } catch (const std::exception &e) {
        rsk_set_error("rsk_dss_densities_host: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_dss_featurize_reversed(const char *seq, const float *x, const float *y, const float *z, uint32_t L, uint8_t *prof)
{
    if (!seq || !x || !y || !z || !prof) { rsk_set_error("rsk_dss_featurize_reversed: NULL argument"); return RSK_E_INVALID; }
    try {
        PDBChain C, R;
        C.m_Seq.assign(seq, seq + L);
        C.m_Xs.assign(x, x + L); C.m_Ys.assign(y, y + L); C.m_Zs.assign(z, z + L);
        C.GetReverse(R);
        DSS D, DR;
        D.Init(C);
        DR.InitReversed(R, D);
        std::vector<std::vector<byte> > P;
        DR.GetProfile(P);
        for (int f = 0; f < RSK_NFEAT; ++f) memcpy(prof + (size_t) f * L, P[f].data(), L);
    } catch (const std::exception &e) {
        rsk_set_error("rsk_dss_featurize_reversed: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_bca_info(const char *path, uint64_t *nchains, uint64_t *nresidues, uint32_t *max_len, uint32_t *max_label)
{
    if (!path) { rsk_set_error("rsk_bca_info: NULL path"); return RSK_E_INVALID; }
    try {
        BCAData B;
        B.Open(path);
        uint64_t tot = 0;
        uint32_t ml = 0, mlab = 0;
        for (uint32_t L : B.m_SeqLengths) { tot += L; ml = std::max(ml, L); }
        for (const std::string &s : B.m_Labels) mlab = std::max<uint32_t>(mlab, (uint32_t) s.size());
        if (nchains) *nchains = B.GetChainCount();
        if (nresidues) *nresidues = tot;
        if (max_len) *max_len = ml;
        if (max_label) *max_label = mlab;
    } catch (const std::exception &e) {
        rsk_set_error("rsk_bca_info: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_bca_read_chain(const char *path, uint64_t idx, char *label, size_t label_cap, char *seq, float *x, float *y, float *z,
                                  uint32_t cap, uint32_t *L)
{
    if (!path || !L) { rsk_set_error("rsk_bca_read_chain: NULL argument"); return RSK_E_INVALID; }
    try {
        BCAData B;
        B.Open(path);
        PDBChain C;
        B.ReadChain(idx, C);
        const uint32_t n = (uint32_t) C.m_Xs.size();
        *L = n;
        if (n > cap) { rsk_set_error("rsk_bca_read_chain: chain of %u residues exceeds the buffers (%u)", n, cap); return RSK_E_RANGE; }
        if (label && label_cap) { strncpy(label, C.m_Label.c_str(), label_cap - 1); label[label_cap - 1] = 0; }
        if (seq) { memset(seq, 0, (size_t) cap); memcpy(seq, C.m_Seq.data(), C.m_Seq.size()); }
        if (x) memcpy(x, C.m_Xs.data(), 4 * (size_t) n);
        if (y) memcpy(y, C.m_Ys.data(), 4 * (size_t) n);
        if (z) memcpy(z, C.m_Zs.data(), 4 * (size_t) n);
    } catch (const std::exception &e) {
        rsk_set_error("rsk_bca_read_chain: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_bca_copy(const char *in_bca, const char *out_bca)
{
    if (!in_bca || !out_bca) { rsk_set_error("rsk_bca_copy: NULL argument"); return RSK_E_INVALID; }
    try {
        BCAData In, Out;
        In.Open(in_bca);
        Out.Create(out_bca);
        PDBChain C;
        for (uint64_t k = 0; k < In.GetChainCount(); ++k) { In.ReadChain(k, C); Out.WriteChain(C); }
        Out.Close();
    } catch (const std::exception &e) {
        rsk_set_error("rsk_bca_copy: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_bca_to_mu_fasta(const char *in_bca, const char *out_fasta)
{
    if (!in_bca || !out_fasta) { rsk_set_error("rsk_bca_to_mu_fasta: NULL argument"); return RSK_E_INVALID; }
    try {
        BCAData In;
        In.Open(in_bca);
        FILE *f = fopen(out_fasta, "w");
        if (!f) throw std::runtime_error(std::string("cannot create ") + out_fasta);
        PDBChain C;
        DSS D;
        std::vector<byte> Mu;
        for (uint64_t k = 0; k < In.GetChainCount(); ++k) {
            In.ReadChain(k, C);
            const uint L = C.GetSeqLength();
            if (L == 0) continue;                                    // SeqToFasta sfasta.cpp:10-11
            D.Init(C);
            D.GetMuLetters(Mu);
            fprintf(f, ">%s\n", C.m_Label.c_str());
            for (uint From = 0; From < L; From += 80) {              // ROWLEN 80 (myutils.h:380)
                for (uint Pos = From; Pos < std::min(L, From + 80); ++Pos) fputc(Mu[Pos] < 26 ? 'A' + Mu[Pos] : 'a' + (Mu[Pos] - 26), f);
                fputc('\n', f);
            }
        }
        fclose(f);
    } catch (const std::exception &e) {
        rsk_set_error("rsk_bca_to_mu_fasta: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

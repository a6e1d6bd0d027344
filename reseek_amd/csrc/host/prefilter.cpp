// prefilter.cpp -- host side of the Mu k-mer prefilter (SURVEY 8a row P12 + the hand-off format, 8f4):
//   RankedScoresBag  rankedscoresbag.cpp:5-51 (AddScore/TruncateVecs), ToTsv :185-231
//   sort tie behaviour of the reference's QuickSortOrderDesc sort.h:70-104,144-154
// The (query, target, diagonal score) triples come from the GPU (k_prefilter.hip); the bounded
// per-query top-B selection depends on arrival order under score ties, so it is replayed here in the
// order the reference produces with -threads 1 (targets ascending).  It is O(#triples) host work.
#include <algorithm>
#include <cstdio>
#include <map>
#include <numeric>
#include <vector>

#include "reseek_host.h"

void rsk_set_error(const char *fmt, ...);

namespace reseek_amd {

static void QuickSortOrderDescRecurse(const uint16_t *Values, int left, int right, uint *Order)
{
    int i = left, j = right;
    const uint16_t pivot = Values[Order[(left + right) / 2]];
    while (i <= j) {
        while (Values[Order[i]] > pivot) i++;
        while (Values[Order[j]] < pivot) j--;
        if (i <= j) { std::swap(Order[i], Order[j]); i++; j--; }
    }
    if (left < j) QuickSortOrderDescRecurse(Values, left, j, Order);
    if (i < right) QuickSortOrderDescRecurse(Values, i, right, Order);
}

void RankedScoresBag::Init(uint QueryCount)
{
    m_QueryCount = QueryCount;
    m_QueryIdxToScoreVec.assign(QueryCount, std::vector<uint16_t>());
    m_QueryIdxToTargetIdxVec.assign(QueryCount, std::vector<uint>());
    m_QueryIdxToLoScore.assign(QueryCount, 0);
}

void RankedScoresBag::TruncateVecs(uint QueryIdx)
{
    std::vector<uint16_t> &ScoreVec = m_QueryIdxToScoreVec[QueryIdx];
    const uint CurrentSize = (uint) ScoreVec.size();
    if (CurrentSize < m_B) return;
    std::vector<uint> Order(CurrentSize);
    std::iota(Order.begin(), Order.end(), 0u);
    QuickSortOrderDescRecurse(ScoreVec.data(), 0, (int) CurrentSize - 1, Order.data());
    std::vector<uint> &TargetIdxVec = m_QueryIdxToTargetIdxVec[QueryIdx];
    std::vector<uint16_t> NewScoreVec(m_B);
    std::vector<uint> NewTargetIdxVec(m_B);
    for (uint k = 0; k < m_B; ++k) { NewScoreVec[k] = ScoreVec[Order[k]]; NewTargetIdxVec[k] = TargetIdxVec[Order[k]]; }
    m_QueryIdxToLoScore[QueryIdx] = NewScoreVec[m_B - 1];
    TargetIdxVec.swap(NewTargetIdxVec);
    ScoreVec.swap(NewScoreVec);
}

void RankedScoresBag::AddScore(uint QueryIdx, uint TargetIdx, uint16_t Score)
{
    std::vector<uint16_t> &ScoreVec = m_QueryIdxToScoreVec[QueryIdx];
    if (Score >= m_QueryIdxToLoScore[QueryIdx]) {
        ScoreVec.push_back(Score);
        m_QueryIdxToTargetIdxVec[QueryIdx].push_back(TargetIdx);
        if (ScoreVec.size() >= 2 * (size_t) m_B) TruncateVecs(QueryIdx);
    }
}

void RankedScoresBag::Finish()
{
    for (uint q = 0; q < m_QueryCount; ++q) TruncateVecs(q);
}

void RankedScoresBag::ToTsv(FILE *f)
{
    if (f == nullptr) return;
    Finish();
    std::map<uint, std::vector<uint> > TargetIdxToQueryIdxs;
    for (uint q = 0; q < m_QueryCount; ++q)
        for (uint t : m_QueryIdxToTargetIdxVec[q]) TargetIdxToQueryIdxs[t].push_back(q);
    fprintf(f, "prefilter\t%u\n", (uint) TargetIdxToQueryIdxs.size());
    for (auto &kv : TargetIdxToQueryIdxs) {
        fprintf(f, "%u\t%u", kv.first, (uint) kv.second.size());
        for (uint q : kv.second) fprintf(f, "\t%u", q);
        fputc('\n', f);
    }
}

}   // namespace reseek_amd

using namespace reseek_amd;

// C-ABI: replay device triples through the RankedScoresBag.  Triples may be in any order (they are
// sorted by target, then query, which is the -threads 1 arrival order up to within-target order).
extern "C" int rsk_rsb_select(const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, uint32_t nqueries, uint32_t rsb_size,
                              uint32_t *out_q, uint32_t *out_t, uint32_t *out_score, size_t *nout, const char *tmp_tsv_path)
{
    if ((n && (!q || !t || !score)) || !nout || rsb_size == 0) { rsk_set_error("rsk_rsb_select: bad argument"); return RSK_E_INVALID; }
    std::vector<uint32_t> ord(n);
    std::iota(ord.begin(), ord.end(), 0u);
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return t[a] != t[b] ? t[a] < t[b] : q[a] < q[b]; });
    RankedScoresBag RSB;
    RSB.m_B = rsb_size;
    RSB.Init(nqueries);
    for (uint32_t k : ord) {
        if (q[k] >= nqueries) { rsk_set_error("rsk_rsb_select: query index out of range"); return RSK_E_INVALID; }
        RSB.AddScore(q[k], t[k], (uint16_t) score[k]);
    }
    RSB.Finish();
    size_t m = 0;
    for (uint32_t qi = 0; qi < nqueries; ++qi) {
        const auto &S = RSB.m_QueryIdxToScoreVec[qi];
        const auto &T = RSB.m_QueryIdxToTargetIdxVec[qi];
        for (size_t k = 0; k < S.size(); ++k) {
            if (out_q && m < n) { out_q[m] = qi; out_t[m] = T[k]; out_score[m] = S[k]; }
            ++m;
        }
    }
    *nout = m;
    if (tmp_tsv_path && *tmp_tsv_path) {
        FILE *f = fopen(tmp_tsv_path, "w");
        if (!f) { rsk_set_error("rsk_rsb_select: cannot create %s", tmp_tsv_path); return RSK_E_INVALID; }
        RSB.ToTsv(f);
        fclose(f);
    }
    return RSK_OK;
}

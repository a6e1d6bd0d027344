// prefilter.cpp -- host side of the Mu k-mer prefilter (SURVEY 8a row P12 + the hand-off format, 8f4):
//   RankedScoresBag  rankedscoresbag.cpp:5-51 (AddScore/TruncateVecs), ToTsv :185-231
//   sort tie behaviour of the reference's QuickSortOrderDesc sort.h:70-104,144-154
// The (query, target, diagonal score) triples come from the GPU (k_prefilter.hip); the bounded
// per-query top-B selection depends on arrival order under score ties, so it is replayed here in the
// order the reference produces with -threads 1 (targets ascending).  It is O(#triples) host work.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <map>
#include <numeric>
#include <thread>
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

// rankedscoresbag.cpp:185-231: "prefilter\t<#targets>", then one line per target (ascending) with its queries (ascending).
// Same bytes as the reference's std::map + fprintf version, built with a counting sort and a hand-rolled integer
// formatter (the map insertions and 15 M fprintf calls were 2 s of the SCOP40 x SCOP40 prefilter stage).
void RankedScoresBag::ToTsv(FILE *f)
{
    if (f == nullptr) return;
    Finish();
    uint MaxT = 0;
    size_t Total = 0;
    for (uint q = 0; q < m_QueryCount; ++q)
        for (uint t : m_QueryIdxToTargetIdxVec[q]) { MaxT = std::max(MaxT, t); ++Total; }
    std::vector<size_t> First((size_t) MaxT + 2, 0);
    for (uint q = 0; q < m_QueryCount; ++q)
        for (uint t : m_QueryIdxToTargetIdxVec[q]) ++First[(size_t) t + 1];
    uint TargetCount = 0;
    for (size_t t = 0; t <= MaxT; ++t) { if (Total && First[t + 1]) ++TargetCount; First[t + 1] += First[t]; }
    std::vector<uint> Queries(Total);
    {
        std::vector<size_t> Cur(First.begin(), First.end() - 1);
        for (uint q = 0; q < m_QueryCount; ++q)                          // ascending q: every target's list comes out sorted
            for (uint t : m_QueryIdxToTargetIdxVec[q]) Queries[Cur[t]++] = q;
    }
    std::string Out;
    Out.reserve(Total * 7 + (size_t) TargetCount * 16 + 64);
    auto put = [&](uint v) {
        char b[12];
        int k = 12;
        do { b[--k] = (char) ('0' + v % 10); v /= 10; } while (v);
        Out.append(b + k, (size_t) (12 - k));
    };
    Out += "prefilter\t"; put(TargetCount); Out += '\n';
    for (size_t t = 0; Total && t <= MaxT; ++t) {
        const size_t lo = First[t], hi = First[t + 1];
        if (lo == hi) continue;
        put((uint) t); Out += '\t'; put((uint) (hi - lo));
        for (size_t k = lo; k < hi; ++k) { Out += '\t'; put(Queries[k]); }
        Out += '\n';
        if (Out.size() > (64u << 20)) { fwrite(Out.data(), 1, Out.size(), f); Out.clear(); }
    }
    fwrite(Out.data(), 1, Out.size(), f);
}

}   // namespace reseek_amd

using namespace reseek_amd;

// C-ABI: replay device triples through the RankedScoresBag.  Triples may be in any order (they are
// sorted by target, then query, which is the -threads 1 arrival order up to within-target order).
extern "C" int rsk_rsb_select(const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, uint32_t nqueries, uint32_t rsb_size,
                              uint32_t *out_q, uint32_t *out_t, uint32_t *out_score, size_t *nout, const char *tmp_tsv_path)
{
    if ((n && (!q || !t || !score)) || !nout || rsb_size == 0) { rsk_set_error("rsk_rsb_select: bad argument"); return RSK_E_INVALID; }
    // A query's bag only depends on ITS triples in target order, so the queries are replayed independently on the
    // host threads: counting sort by query, then per query a sort by target and the AddScore sequence.
    for (size_t k = 0; k < n; ++k)
        if (q[k] >= nqueries) { rsk_set_error("rsk_rsb_select: query index out of range"); return RSK_E_INVALID; }
    PhaseTimer tm("rsk_rsb_select");
    std::vector<size_t> qstart((size_t) nqueries + 1, 0);
    for (size_t k = 0; k < n; ++k) ++qstart[q[k] + 1];
    for (uint32_t i = 0; i < nqueries; ++i) qstart[i + 1] += qstart[i];
    std::vector<uint64_t> byq(n);                       // target << 16 | score, grouped by query
    {
        std::vector<size_t> cur(qstart.begin(), qstart.end() - 1);
        for (size_t k = 0; k < n; ++k) byq[cur[q[k]]++] = ((uint64_t) t[k] << 16) | (uint16_t) score[k];
    }
    tm.lap("group by query");
    RankedScoresBag RSB;
    RSB.m_B = rsb_size;
    RSB.Init(nqueries);
    {
        const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>((size_t) HostThreads(64), n / 65536 + 1));
        std::atomic<uint32_t> next{0};
        auto body = [&]() {
            for (;;) {
                const uint32_t qi = next.fetch_add(1);
                if (qi >= nqueries) return;
                uint64_t *b = byq.data() + qstart[qi], *e = byq.data() + qstart[qi + 1];
                std::sort(b, e);                                   // a (query, target) pair occurs once: order = target order
                for (uint64_t *x = b; x != e; ++x) RSB.AddScore(qi, (uint) (*x >> 16), (uint16_t) (*x & 0xFFFF));
                RSB.TruncateVecs(qi);
            }
        };
        if (T == 1) body();
        else {
            std::vector<std::thread> ts;
            for (unsigned k = 0; k < T; ++k) ts.emplace_back(body);
            for (auto &th : ts) th.join();
        }
    }
    tm.lap("replay (threads)");
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
    tm.lap("hand-off file");
    return RSK_OK;
}

// prefilter.cpp -- host side of the Mu k-mer prefilter (SURVEY 8a row P12 + the hand-off format, 8f4):
//   RankedScoresBag  rankedscoresbag.cpp:5-51 (AddScore/TruncateVecs), ToTsv :185-231
//   sort tie behaviour of the reference's QuickSortOrderDesc sort.h:70-104,144-154
// The (query, target, diagonal score) triples come from the GPU (k_prefilter.hip); the bounded
// per-query top-B selection depends on arrival order under score ties, so it is replayed here in the
// order the reference produces with -threads 1 (targets ascending).  It is O(#triples) host work.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <functional>
#include <map>
#include <numeric>
#include <thread>
#include <vector>

#include "reseek_host.h"
#include "../rsk_internal.h"

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
    // scratch of the calling thread, reused between calls (a replay truncates ~6 times per query: three fresh vectors per
    // call were a quarter of its time)
    static thread_local std::vector<uint> Order, NewTargetIdxVec;
    static thread_local std::vector<uint16_t> NewScoreVec;
    Order.resize(CurrentSize);
    std::iota(Order.begin(), Order.end(), 0u);
    QuickSortOrderDescRecurse(ScoreVec.data(), 0, (int) CurrentSize - 1, Order.data());
    std::vector<uint> &TargetIdxVec = m_QueryIdxToTargetIdxVec[QueryIdx];
    NewScoreVec.resize(m_B);
    NewTargetIdxVec.resize(m_B);
    for (uint k = 0; k < m_B; ++k) { NewScoreVec[k] = ScoreVec[Order[k]]; NewTargetIdxVec[k] = TargetIdxVec[Order[k]]; }
    m_QueryIdxToLoScore[QueryIdx] = NewScoreVec[m_B - 1];
    TargetIdxVec.assign(NewTargetIdxVec.begin(), NewTargetIdxVec.end());      // capacity stays (2 B entries)
    ScoreVec.assign(NewScoreVec.begin(), NewScoreVec.end());
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
    // (bags are independent: on the host threads)
    rsk_parallel_for(m_QueryCount, 64, [&](size_t lo, size_t hi) { for (size_t q = lo; q < hi; ++q) TruncateVecs((uint) q); });
}

// The selected candidates target-major: First[t] .. First[t + 1] = the queries (ascending) that kept target t.  This is the
// order of the hand-off file (rankedscoresbag.cpp:185-231), hence of PostMuFilter's candidate pairs.
void RankedScoresBag::GroupByTarget(std::vector<size_t> &First, std::vector<uint> &Queries, uint &TargetCount)
{
    Finish();
    // counting sort by target on the host threads: slices of consecutive queries count their targets, the slices' counts are
    // stacked per target in slice order (so a target's queries come out ascending), then every slice scatters its own
    const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>(HostThreads(32), (size_t) m_QueryCount / 256 + 1));
    std::vector<uint> maxT(T, 0);
    std::vector<size_t> tot(T, 0);
    auto slice = [&](unsigned k, uint &lo, uint &hi) { lo = (uint) ((uint64_t) m_QueryCount * k / T); hi = (uint) ((uint64_t) m_QueryCount * (k + 1) / T); };
    auto on_slices = [&](const std::function<void(unsigned)> &fn) {
        if (T == 1) { fn(0); return; }
        std::vector<std::thread> ts;
        for (unsigned k = 0; k < T; ++k) ts.emplace_back(fn, k);
        for (auto &t : ts) t.join();
    };
    on_slices([&](unsigned k) {
        uint lo, hi;
        slice(k, lo, hi);
        for (uint q = lo; q < hi; ++q)
            for (uint t : m_QueryIdxToTargetIdxVec[q]) { maxT[k] = std::max(maxT[k], t); ++tot[k]; }
    });
    size_t Total = 0;
    uint MaxT = 0;
    for (unsigned k = 0; k < T; ++k) { Total += tot[k]; MaxT = std::max(MaxT, maxT[k]); }
    const size_t NT = Total ? (size_t) MaxT + 1 : 0;
    First.assign(NT + 1, 0);
    Queries.resize(Total);
    TargetCount = 0;
    if (!Total) return;
    std::vector<std::vector<uint32_t> > cnt(T);
    on_slices([&](unsigned k) {
        uint lo, hi;
        slice(k, lo, hi);
        cnt[k].assign(NT, 0);
        for (uint q = lo; q < hi; ++q)
            for (uint t : m_QueryIdxToTargetIdxVec[q]) ++cnt[k][t];
    });
    // First[t] and, per slice, its first position inside target t's run (cnt[k][t] becomes that offset)
    size_t run = 0;
    for (size_t t = 0; t < NT; ++t) {
        First[t] = run;
        size_t here = 0;
        for (unsigned k = 0; k < T; ++k) { const uint32_t c = cnt[k][t]; cnt[k][t] = (uint32_t) here; here += c; }
        if (here) ++TargetCount;
        run += here;
    }
    First[NT] = run;
    on_slices([&](unsigned k) {
        uint lo, hi;
        slice(k, lo, hi);
        for (uint q = lo; q < hi; ++q)                                   // ascending q inside the slice
            for (uint t : m_QueryIdxToTargetIdxVec[q]) Queries[First[t] + cnt[k][t]++] = q;
    });
}

// candidate pairs in hand-off order without the file (the single-process search hands them over in memory)
void RankedScoresBag::ToPairs(std::vector<uint32_t> &pq, std::vector<uint32_t> &pt)
{
    std::vector<size_t> First;
    std::vector<uint> Queries;
    uint TargetCount = 0;
    GroupByTarget(First, Queries, TargetCount);
    pq.assign(Queries.begin(), Queries.end());
    pt.resize(Queries.size());
    rsk_parallel_for(First.size() - 1, 256, [&](size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; ++t)
            for (size_t k = First[t]; k < First[t + 1]; ++k) pt[k] = (uint32_t) t;
    });
}

// rankedscoresbag.cpp:185-231: "prefilter\t<#targets>", then one line per target (ascending) with its queries (ascending).
// Same bytes as the reference's std::map + fprintf version, built with a counting sort and a hand-rolled integer
// formatter (the map insertions and 15 M fprintf calls were 2 s of the SCOP40 x SCOP40 prefilter stage).
void RankedScoresBag::ToTsv(FILE *f)
{
    if (f == nullptr) return;
    std::vector<size_t> First;
    std::vector<uint> Queries;
    uint TargetCount = 0;
    GroupByTarget(First, Queries, TargetCount);
    std::string Out;
    Out.reserve(Queries.size() * 7 + (size_t) TargetCount * 16 + 64);
    auto put = [&](uint v) {
        char b[12];
        int k = 12;
        do { b[--k] = (char) ('0' + v % 10); v /= 10; } while (v);
        Out.append(b + k, (size_t) (12 - k));
    };
    Out += "prefilter\t"; put(TargetCount); Out += '\n';
    for (size_t t = 0; t + 1 < First.size(); ++t) {
        const size_t lo = First[t], hi = First[t + 1];
        if (lo == hi) continue;
        put((uint) t); Out += '\t'; put((uint) (hi - lo));
        for (size_t k = lo; k < hi; ++k) { Out += '\t'; put(Queries[k]); }
        Out += '\n';
        if (Out.size() > (64u << 20)) { fwrite(Out.data(), 1, Out.size(), f); Out.clear(); }
    }
    fwrite(Out.data(), 1, Out.size(), f);
}

// AddScore sequence of every query on the host threads: `items` holds per query (qstart) its triples in target order,
// target in bits 16..47 and score in bits 0..15 of an element (a query's bag depends on its own triples only).
static void ReplayGrouped(RankedScoresBag &RSB, const uint64_t *items, const size_t *qstart, uint32_t nqueries, size_t n, bool Final = true)
{
    const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>((size_t) HostThreads(64), n / 65536 + 1));
    std::atomic<uint32_t> next{0};
    auto body = [&]() {
        for (;;) {
            const uint32_t qi = next.fetch_add(1);
            if (qi >= nqueries) return;
            for (const uint64_t *x = items + qstart[qi], *e = items + qstart[qi + 1]; x != e; ++x)
                RSB.AddScore(qi, (uint) ((*x >> 16) & 0xFFFFFFFFu), (uint16_t) (*x & 0xFFFF));
            if (Final) RSB.TruncateVecs(qi);               // (more targets follow otherwise: the reference only truncates at 2 B entries and at the end)
        }
    };
    if (T == 1) body();
    else {
        std::vector<std::thread> ts;
        for (unsigned k = 0; k < T; ++k) ts.emplace_back(body);
        for (auto &th : ts) th.join();
    }
}

// survivors in query order + the optional hand-off file
static int EmitSelection(RankedScoresBag &RSB, uint32_t nqueries, size_t cap, uint32_t *out_q, uint32_t *out_t, uint32_t *out_score, size_t *nout,
                         const char *tmp_tsv_path)
{
    size_t m = 0;
    for (uint32_t qi = 0; qi < nqueries; ++qi) {
        const auto &S = RSB.m_QueryIdxToScoreVec[qi];
        const auto &T = RSB.m_QueryIdxToTargetIdxVec[qi];
        for (size_t k = 0; k < S.size(); ++k) {
            if (out_q && m < cap) { out_q[m] = qi; out_t[m] = T[k]; out_score[m] = S[k]; }
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

// keys = query << 48 | target << 16 | score, ascending (rsk_triples_sort_dev), of ONE contiguous target range: the bags take
// them in (the ranges of a scan arrive in target order); Final = this is the last range
int ReplayAppendSortedKeys(RankedScoresBag &RSB, const uint64_t *keys, size_t n, uint32_t nqueries, bool Final)
{
    if (n && (keys[n - 1] >> 48) >= nqueries) { rsk_set_error("rsk_rsb_select_keys: query index out of range"); return RSK_E_INVALID; }
    std::vector<size_t> qstart((size_t) nqueries + 1, 0);
    rsk_parallel_for((size_t) nqueries + 1, 64, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) qstart[i] = i > 0xFFFF ? n : (size_t) (std::lower_bound(keys, keys + n, (uint64_t) i << 48) - keys);
    });
    ReplayGrouped(RSB, keys, qstart.data(), nqueries, n, Final);
    return RSK_OK;
}

// ... of the whole scan at once
int ReplaySortedKeys(RankedScoresBag &RSB, const uint64_t *keys, size_t n, uint32_t nqueries, uint32_t rsb_size)
{
    RSB.m_B = rsb_size;
    RSB.Init(nqueries);
    return ReplayAppendSortedKeys(RSB, keys, n, nqueries, true);
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
    // (a (query, target) pair occurs once: sorting a query's elements = target order)
    rsk_parallel_for(nqueries, 16, [&](size_t lo, size_t hi) {
        for (size_t qi = lo; qi < hi; ++qi) std::sort(byq.begin() + qstart[qi], byq.begin() + qstart[qi + 1]);
    });
    RankedScoresBag RSB;
    RSB.m_B = rsb_size;
    RSB.Init(nqueries);
    ReplayGrouped(RSB, byq.data(), qstart.data(), nqueries, n);
    tm.lap("replay (threads)");
    const int rc = EmitSelection(RSB, nqueries, n, out_q, out_t, out_score, nout, tmp_tsv_path);
    if (rc != RSK_OK) return rc;
    tm.lap("hand-off file");
    return RSK_OK;
}

extern "C" int rsk_rsb_select_keys(const uint64_t *keys, size_t n, uint32_t nqueries, uint32_t rsb_size, uint32_t *out_q, uint32_t *out_t,
                                   uint32_t *out_score, size_t *nout, const char *tmp_tsv_path)
{
    if ((n && !keys) || !nout || rsb_size == 0 || nqueries > 65536) { rsk_set_error("rsk_rsb_select_keys: bad argument"); return RSK_E_INVALID; }
    PhaseTimer tm("rsk_rsb_select_keys");
    RankedScoresBag RSB;
    int rc = ReplaySortedKeys(RSB, keys, n, nqueries, rsb_size);
    if (rc != RSK_OK) return rc;
    tm.lap("replay (threads)");
    rc = EmitSelection(RSB, nqueries, n, out_q, out_t, out_score, nout, tmp_tsv_path);
    tm.lap("hand-off file");
    return rc;
}

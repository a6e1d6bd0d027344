// dbsearcher.cpp -- host mirror of DBSearcher (dbsearcher.cpp, runself.cpp, runquery.cpp,
// profileloader.cpp) and the C-ABI entry points that stand for `reseek -search` (search.cpp:20-111).
//
// The reference hands one pair at a time to one DSSAligner per thread (runself.cpp:13-70).  Here
// the pair space is enumerated in the same order but scored in GPU batches through the C-ABI:
//   Mu filter (rsk_mu_filter_dev)  ->  survivors  ->  rsk_align_pairs  ->  hit records,
// then every hit is replayed through DSSAligner + BaseOnAln so Reject/-evalue/-mints, OnAln
// subclasses and -columns behave as in the reference.  Pairs that take the long-chain MKF path
// (DoMKF, dssaligner.cpp:715) go through RunMKFPairs: seeding and the gapped X-drop extensions in GPU batches
// (rsk_mkf_seed_pairs, rsk_xdrop_pairs), chaining / start selection / merge / statistics on host threads.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <chrono>
#include <future>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "reseek_host.h"
#include "../rsk_internal.h"

namespace reseek_amd {
static void check(int rc, const char *what)
{
    if (rc != RSK_OK) throw std::runtime_error(std::string(what) + ": " + rsk_last_error());
}

void DeviceBuffer::Make(rsk_ctx *Ctx, size_t Bytes, const char *What)
{
    Free();
    if (!Ctx) throw std::runtime_error(std::string(What) + ": no GPU context");
    if (hipSetDevice(Ctx->device) != hipSuccess) throw std::runtime_error(std::string(What) + ": hipSetDevice failed");
    check(rsk_dev_malloc(Ctx, &m_Ptr, std::max<size_t>(Bytes, 16)), What);
}

void DeviceBuffer::Free()
{
    if (m_Ptr) (void) hipFree(m_Ptr);
    m_Ptr = nullptr;
}

DBSearcher::~DBSearcher()
{
    if (m_OwnsChains) {
        for (auto p : m_DBChains) delete p;
        for (auto p : m_DBProfiles) delete p;
        for (auto p : m_DBMuLettersVec) delete p;
        for (auto p : m_DBMuKmersVec) delete p;
    }
    if (m_Db) rsk_db_destroy(m_Db);
}

void DBSearcher::AddChain(PDBChain *ptrChain, std::vector<std::vector<byte> > *ptrProfile, std::vector<byte> *ptrMuLetters)
{
    ptrChain->m_Idx = (uint) m_DBChains.size();
    m_DBChains.push_back(ptrChain);
    m_DBProfiles.push_back(ptrProfile);
    m_DBMuLettersVec.push_back(ptrMuLetters);
}

// Mu 3-mers with pattern "111" (DSS::GetMuKmers dss.cpp:659-682): base-36 code of 3 consecutive letters.
static void GetMuKmers(const std::vector<byte> &Mu, std::vector<uint> &Kmers)
{
    Kmers.clear();
    const size_t L = Mu.size();
    for (size_t i = 0; i + 3 <= L; ++i) Kmers.push_back(((uint) Mu[i] * 36 + Mu[i + 1]) * 36 + Mu[i + 2]);
}

static bool EndsWith(const std::string &s, const std::string &suf)
{
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

unsigned HostThreads(unsigned cap)
{
    if (const char *e = getenv("RSK_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned) v; }
    static const unsigned avail = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                         // cgroup v2: "<quota|max> <period>"
            char q[64];
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {      // cgroup v1
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 0; fclose(h); }
        }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned) std::max<long long>(1, (quota + period - 1) / period));
        return n;
    }();
    return std::max(1u, std::min(avail, cap));
}

namespace {
}   // namespace

// ProfileLoader::Load profileloader.cpp:72 for a .bca file: read, featurise (host threads), self-rev (GPU batch)
void DBSearcher::LoadBCA(const std::string &FN)
{
    PhaseTimer tm("LoadBCA");
    BCAData B;
    B.Open(FN);
    const uint64_t n = B.GetChainCount();
    std::vector<PDBChain *> Chains;
    Chains.reserve(n);
    for (uint64_t k = 0; k < n; ++k) {
        PDBChain *C = new PDBChain;
        B.ReadChain(k, *C);
        Chains.push_back(C);
    }
    tm.lap("read chains");
    LoadChains(Chains);
}

// The chains become this searcher's set (ownership taken; Chains is left empty): DSS profile, Mu letters and Mu 3-mers
// of every chain on the host threads, then the self-rev scores in one GPU batch.
void DBSearcher::LoadChains(std::vector<PDBChain *> &Chains)
{
    PhaseTimer tm("LoadChains");
    if (!m_Ctx) m_Ctx = DefaultCtx();
    if (m_Opts.mode == AM_Invalid) m_Opts = g_Opts;
    for (PDBChain *C : Chains) {
        if (C->GetSeqLength() < 1) { delete C; continue; }              // m_MinChainLength = 1 (profileloader.cpp:82)
        AddChain(C, new std::vector<std::vector<byte> >, new std::vector<byte>);
        m_DBMuKmersVec.push_back(new std::vector<uint>);
    }
    Chains.clear();
    const uint N = GetDBChainCount();
    const unsigned T = HostThreads(128);
    std::atomic<uint> next{0};
    const bool WantRev = !m_Opts.selfrev0 && m_Ctx;
    m_RevProfiles.clear();
    if (WantRev) m_RevProfiles.resize(N);
    // The per-residue quantities of the featurisation come from the device for the whole batch, chains and reversed
    // chains (rsk_dss_densities, k_dss.hip): SS characters, Conf letters and nearest neighbours (float comparison chains:
    // identical to the host's) and the two density features (two thirds of the host cost: libm exp), which
    // DSS::UseDeviceDensities accepts chain by chain only where no binned value is near a bin boundary, so the letters
    // stay the host's.  RSK_GPU_DENSITY=0: host only.
    std::vector<uint64_t> roff;
    std::unique_ptr<char[]> ssb;                             // [2][total]: SS of the chains, of the reversed chains
    std::unique_ptr<uint8_t[]> confb;                        // [2][total]: Conf letters
    std::unique_ptr<double[]> dens;                          // [4][total]: density / strand density of the chains, of the reversed chains
    std::unique_ptr<uint32_t[]> nens;                        // [4][total]: NEN / REN of the chains, of the reversed chains
    uint64_t rtotal = 0;
    std::atomic<uint64_t> dens_fallbacks{0};
    if (m_Ctx && N && !(getenv("RSK_GPU_DENSITY") && atoi(getenv("RSK_GPU_DENSITY")) == 0)) {
        roff.assign((size_t) N + 1, 0);
        for (uint i = 0; i < N; ++i) roff[i + 1] = roff[i] + m_DBChains[i]->GetSeqLength();
        rtotal = roff[N];
        std::unique_ptr<float[]> px(new float[rtotal + 1]), py(new float[rtotal + 1]), pz(new float[rtotal + 1]);
        std::vector<uint32_t> len(N);
        rsk_parallel_for(N, 256, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const PDBChain &C = *m_DBChains[i];
                const uint L = C.GetSeqLength();
                len[i] = L;
                memcpy(&px[roff[i]], C.m_Xs.data(), 4 * (size_t) L);
                memcpy(&py[roff[i]], C.m_Ys.data(), 4 * (size_t) L);
                memcpy(&pz[roff[i]], C.m_Zs.data(), 4 * (size_t) L);
            }
        });
        ssb.reset(new char[2 * rtotal + 2]);
        confb.reset(new uint8_t[2 * rtotal + 2]);
        dens.reset(new double[4 * rtotal + 4]);
        nens.reset(new uint32_t[4 * rtotal + 4]);
        DSS D0;
        // device calls of at most 16 M residues (a self search loads its whole set here); RSK_DSS_CHUNK_RESIDUES: tests
        const uint64_t chunk = getenv("RSK_DSS_CHUNK_RESIDUES") ? (uint64_t) std::max(1ll, atoll(getenv("RSK_DSS_CHUNK_RESIDUES"))) : (uint64_t) 16 << 20;
        for (uint c0 = 0; c0 < N;) {
            uint c1 = c0 + 1;
            while (c1 < N && roff[c1 + 1] - roff[c0] <= chunk) ++c1;
            const uint64_t o = roff[c0];
            check(rsk_dss_densities(m_Ctx, c1 - c0, len.data() + c0, px.get() + o, py.get() + o, pz.get() + o, ssb.get() + o, ssb.get() + rtotal + o,
                                    confb.get() + o, confb.get() + rtotal + o, D0.m_Density_W, D0.m_Density_w, D0.m_SSDensity_w, D0.m_Density_Radius,
                                    D0.m_SSDensity_epsilon, dens.get() + o, dens.get() + rtotal + o, dens.get() + 2 * rtotal + o,
                                    dens.get() + 3 * rtotal + o, D0.m_NEN_W, D0.m_NEN_w, nens.get() + o, nens.get() + rtotal + o,
                                    nens.get() + 2 * rtotal + o, nens.get() + 3 * rtotal + o),
                  "rsk_dss_densities");
            c0 = c1;
        }
        tm.lap("densities (device)");
    }
    auto body = [&]() {
        DSS D, DR;
        D.SetParams(*m_Params);
        DR.SetParams(*m_Params);
        for (;;) {
            const uint i = next.fetch_add(1);
            if (i >= N) return;
            // featurise into this thread's own vectors, then hand them over: the destination vector headers of
            // neighbouring chains share cache lines, per-residue push_back on them would ping-pong between cores
            std::vector<std::vector<byte> > Prof;
            std::vector<byte> Mu;
            std::vector<uint> Kmers;
            D.Init(*m_DBChains[i]);
            if (dens) {
                D.UseDeviceLocal(ssb.get() + roff[i], confb.get() + roff[i]);
                D.UseDeviceNENs(nens.get() + roff[i], nens.get() + rtotal + roff[i]);
                if (!D.UseDeviceDensities(dens.get() + roff[i], dens.get() + rtotal + roff[i])) ++dens_fallbacks;
            }
            D.GetProfile(Prof);
            D.GetMuLetters(Mu);
            DSS::GetMuKmers(Mu, Kmers, m_Params->m_MKFPatternStr);
            m_DBProfiles[i]->swap(Prof);
            m_DBMuLettersVec[i]->swap(Mu);
            m_DBMuKmersVec[i]->swap(Kmers);
            if (WantRev) {
                // profile of the reversed chain for ComputeSelfRevScores, while D still holds this chain's exp() table
                PDBChain R;
                std::vector<std::vector<byte> > RevProf;
                m_DBChains[i]->GetReverse(R);
                DR.Init(R);
                if (!(dens && DR.UseDeviceDensities(dens.get() + 2 * rtotal + roff[i], dens.get() + 3 * rtotal + roff[i]))) {
                    if (dens) ++dens_fallbacks;
                    DR.InitReversed(R, D);
                }
                if (dens) {
                    DR.UseDeviceLocal(ssb.get() + rtotal + roff[i], confb.get() + rtotal + roff[i]);
                    DR.UseDeviceNENs(nens.get() + 2 * rtotal + roff[i], nens.get() + 3 * rtotal + roff[i]);
                }
                DR.GetProfile(RevProf);
                m_RevProfiles[i].swap(RevProf);
            }
        }
    };
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < T; ++t) ts.emplace_back(body);
    for (auto &t : ts) t.join();
    tm.lap("featurise (host)");
    if (dens && getenv("RSK_TRACE")) fprintf(stderr, "[LoadChains] %u chains: %llu chain featurisations redone on the host (density near a bin boundary)\n", N,
                                             (unsigned long long) dens_fallbacks.load());
    ComputeSelfRevScores();
    tm.lap("self-rev scores");
}

// GetSelfRevScore alignpair.cpp:7-24 for every chain: AlignQueryTarget of the chain against its reversed copy
// (profile of the reversed chain; the Mu letters / k-mers passed for BOTH sides are the un-reversed ones -- the
// reference's behaviour), m_AlnFwdScore is the result.  Chains that take the MKF path (DoMKF: length >= m_MKFL) go through
// the same device batch as the search's long-chain pairs (RunMKFPairs), against a view of the reversed chains.
void DBSearcher::ComputeSelfRevScores()
{
    const uint N = GetDBChainCount();
    m_DBSelfRevScores.assign(N, 0.0f);
    if (m_Opts.selfrev0 || N == 0) return;
    if (!m_Ctx) throw std::runtime_error("DBSearcher: self-rev scores need a GPU context");
    DSSParams DAP = *m_Params;
    bool HaveMu = true;
    if (!m_SelfRevQueryFlavour) {
        DAP.m_UsePara = false;
        DAP.m_Omega = 0;
        HaveMu = m_Params->m_Omega > 0;                                 // LoadDB dbsearcher.cpp:249-251
    }
    PhaseTimer tm("SelfRev");
    // reversed chains and their profiles
    std::vector<PDBChain> Rev(N);
    std::vector<std::vector<std::vector<byte> > > RevProf(N);
    if (m_RevProfiles.size() == N) {
        // LoadBCA featurised the reversed chains together with the chains (shared exp() tables)
        RevProf.swap(m_RevProfiles);                                    // (the long chains below reverse themselves)
    } else {
        const unsigned T = HostThreads(128);
        std::atomic<uint> next{0};
        auto body = [&]() {
            DSS D;
            D.SetParams(*m_Params);
            for (;;) {
                const uint i = next.fetch_add(1);
                if (i >= N) return;
                PDBChain R;
                std::vector<std::vector<byte> > Prof;
                m_DBChains[i]->GetReverse(R);
                D.Init(R);
                D.GetProfile(Prof);
                std::swap(Rev[i], R);
                RevProf[i].swap(Prof);
            }
        };
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < T; ++t) ts.emplace_back(body);
        for (auto &t : ts) t.join();
    }
    m_RevProfiles.clear();
    tm.lap("reverse + featurise");
    std::vector<uint32_t> gpu, mkf;
    for (uint i = 0; i < N; ++i) {
        const uint L = m_DBChains[i]->GetSeqLength();
        const bool DoMKF = HaveMu && !m_DBMuKmersVec[i]->empty() && L >= DAP.m_MKFL;      // DoMKF dssaligner.cpp:715
        (DoMKF ? mkf : gpu).push_back(i);
    }
    // The chains themselves go up once, as the set the search will use (UploadToGpu, its self-rev scores completed at the
    // end of this function): it is the query side here.  Only the reversed profiles need a set of their own -- with the
    // un-reversed Mu letters (what the reference passes for both sides) and, when long chains are present, the reversed
    // coordinates (the long-chain batch computes the alignment statistics of every pair; only the score is used here).
    std::vector<uint32_t> len(N);
    std::vector<size_t> start((size_t) N + 1, 0);
    for (uint i = 0; i < N; ++i) { len[i] = m_DBChains[i]->GetSeqLength(); start[i + 1] = start[i] + len[i]; }
    const size_t tot = start[N];
    std::vector<uint8_t> mu(tot), pr(tot * RSK_NFEAT);
    std::vector<float> rx, ry, rz;
    if (!mkf.empty()) { rx.resize(tot); ry.resize(tot); rz.resize(tot); }
    rsk_parallel_for(N, 512, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint L = len[i];
            const size_t o = start[i];
            memcpy(&mu[o], m_DBMuLettersVec[i]->data(), L);
            for (int f = 0; f < RSK_NFEAT; ++f) memcpy(&pr[o * RSK_NFEAT + (size_t) f * L], RevProf[i][f].data(), L);
            if (!rx.empty()) {
                const PDBChain &C = *m_DBChains[i];
                for (uint k = 0; k < L; ++k) { rx[o + k] = C.m_Xs[L - 1 - k]; ry[o + k] = C.m_Ys[L - 1 - k]; rz[o + k] = C.m_Zs[L - 1 - k]; }
            }
        }
    });
    tm.lap("pack");
    UploadToGpu();
    rsk_db *fdb = m_Db, *rdb = nullptr;
    struct guard { rsk_db *d; ~guard() { if (d) rsk_db_destroy(d); } } g2{ nullptr };
    check(rsk_db_create(m_Ctx, N, len.data(), mu.data(), pr.data(), rx.empty() ? nullptr : rx.data(), rx.empty() ? nullptr : ry.data(),
                        rx.empty() ? nullptr : rz.data(), nullptr, &rdb),
          "rsk_db_create");
    g2.d = rdb;
    tm.lap("upload");
    if (!gpu.empty()) {
        std::vector<uint32_t> idx = gpu;
        if (DAP.m_Omega > 0) {                                           // MuFilter dssaligner.cpp:817-826 (self vs self letters)
            std::vector<uint8_t> pass(idx.size());
            check(rsk_mu_filter_pairs(m_Ctx, fdb, fdb, idx.data(), idx.data(), idx.size(), DAP.m_ParaMuGapOpen, DAP.m_ParaMuGapExt, DAP.m_Omega,
                                      DAP.m_OmegaFwd, pass.data(), nullptr, nullptr),
                  "rsk_mu_filter_pairs");
            std::vector<uint32_t> keep;
            for (size_t k = 0; k < idx.size(); ++k)
                if (pass[k]) keep.push_back(idx[k]);
            idx.swap(keep);
        }
        for (auto &be : AlignBatches(m_Opts, *this, *this, idx, idx)) {
            const size_t b = be.first, m = be.second - be.first;
            std::vector<rsk_aln> out(m);
            check(rsk_align_pairs(m_Ctx, fdb, rdb, idx.data() + b, idx.data() + b, m, DAP.m_GapOpen, DAP.m_GapExt, DAP.m_MinFwdScore, out.data(),
                                  nullptr, 0),
                  "rsk_align_pairs");
            for (size_t k = 0; k < m; ++k) m_DBSelfRevScores[idx[b + k]] = out[k].score;
        }
        tm.lap("GPU filter + SW");
    }
    if (!mkf.empty()) {
        // B side of the long-chain batch: the reversed chains as a borrowed view (chain objects only for the long ones)
        DBSearcher RevView;
        RevView.m_OwnsChains = false;
        RevView.m_Params = &DAP; RevView.m_Opts = m_Opts; RevView.m_Ctx = m_Ctx;
        RevView.m_DBChains.assign(N, nullptr);
        RevView.m_DBProfiles.resize(N);
        for (uint i = 0; i < N; ++i) RevView.m_DBProfiles[i] = &RevProf[i];
        RevView.m_DBMuLettersVec = m_DBMuLettersVec;
        RevView.m_DBMuKmersVec = m_DBMuKmersVec;
        RevView.m_DBSelfRevScores.assign(N, FLT_MAX);
        std::vector<float> SelfRevA(N, FLT_MAX);
        SelfRevA.swap(m_DBSelfRevScores);                                 // SetQuery(..., FLT_MAX) alignpair.cpp:14-17
        std::vector<std::pair<uint32_t, uint32_t> > Pairs;
        for (uint32_t i : mkf) {
            if (Rev[i].GetSeqLength() == 0) m_DBChains[i]->GetReverse(Rev[i]);
            RevView.m_DBChains[i] = &Rev[i];
            Pairs.emplace_back(i, i);
        }
        RevView.m_Db = rdb;
        std::vector<float> Score(N, 0.0f);
        try {
            RunMKFPairs(m_Ctx, DAP, "", *this, RevView, Pairs, [&](DSSAligner &DA, uint i, uint) { Score[i] = DA.m_AlnFwdScore; });
        } catch (...) {
            RevView.m_Db = nullptr;
            SelfRevA.swap(m_DBSelfRevScores);
            throw;
        }
        RevView.m_Db = nullptr;                                           // rdb belongs to the guard above
        SelfRevA.swap(m_DBSelfRevScores);
        for (uint32_t i : mkf) m_DBSelfRevScores[i] = Score[i];
        tm.lap("long chains (device batch)");
    }
    if (m_Db) check(rsk_db_update_selfrev(m_Db, m_DBSelfRevScores.data()), "rsk_db_update_selfrev");
}

void DBSearcher::LoadDB(const std::string &DBFN)
{
    if (!m_Ctx) m_Ctx = DefaultCtx();
    if (m_Opts.mode == AM_Invalid) m_Opts = g_Opts;
    if (EndsWith(DBFN, ".bca")) { LoadBCA(DBFN); return; }
    FILE *f = fopen(DBFN.c_str(), "rb");
    if (!f) throw std::runtime_error("LoadDB: cannot open " + DBFN);
    auto rd = [&](void *p, size_t n) { if (n && fread(p, 1, n, f) != n) { fclose(f); throw std::runtime_error("LoadDB: truncated " + DBFN); } };
    char magic[8];
    rd(magic, 8);
    if (memcmp(magic, "RSKDB1\0\0", 8) != 0) { fclose(f); throw std::runtime_error("LoadDB: " + DBFN + " is not an RSKDB1 container"); }
    uint32_t n, nfeat;
    rd(&n, 4); rd(&nfeat, 4);
    if (nfeat != RSK_NFEAT) { fclose(f); throw std::runtime_error("LoadDB: feature count mismatch"); }
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t L, ll;
        rd(&L, 4); rd(&ll, 4);
        PDBChain *C = new PDBChain;
        C->m_Label.resize(ll); rd(&C->m_Label[0], ll);
        C->m_Seq.resize(L); rd(&C->m_Seq[0], L);
        auto *Mu = new std::vector<byte>(L);
        rd(Mu->data(), L);
        auto *Prof = new std::vector<std::vector<byte> >(nfeat, std::vector<byte>(L));
        for (uint32_t fi = 0; fi < nfeat; ++fi) rd((*Prof)[fi].data(), L);
        C->m_Xs.resize(L); C->m_Ys.resize(L); C->m_Zs.resize(L);
        rd(C->m_Xs.data(), 4 * (size_t) L); rd(C->m_Ys.data(), 4 * (size_t) L); rd(C->m_Zs.data(), 4 * (size_t) L);
        float selfrev;
        rd(&selfrev, 4);
        uint32_t nk;
        rd(&nk, 4);
        std::vector<uint> stored(nk);
        rd(stored.data(), 4 * (size_t) nk);
        auto *Kmers = new std::vector<uint>;
        GetMuKmers(*Mu, *Kmers);
        if (*Kmers != stored) { fclose(f); throw std::runtime_error("LoadDB: stored Mu k-mers disagree with the letters"); }
        AddChain(C, Prof, Mu);
        m_DBMuKmersVec.push_back(Kmers);
        m_DBSelfRevScores.push_back(m_Opts.selfrev0 ? 0.0f : selfrev);
    }
    fclose(f);
}

void DBSearcher::Setup()
{
    if (!m_Ctx) m_Ctx = DefaultCtx();
    if (m_Opts.mode == AM_Invalid) m_Opts = g_Opts;
    if (m_Opts.evalue_set) m_MaxEvalue = m_Opts.evalue;
    else m_MaxEvalue = (m_Opts.mode == AM_VerySensitive) ? DBL_MAX : 10;
    m_HitCount = 0;
    m_ProcessedPairCount = 0;
    m_DA.SetParams(*m_Params);
    m_DA.SetColumns(m_Opts.columns);
    m_DA.m_Ctx = m_Ctx;
    if (m_Devices.empty() && m_OwnsChains) m_Devices = ParseDeviceList(getenv("RSK_DEVICES"));      // views / replicas stay on their context
    OnSetup();
}

bool DBSearcher::Reject(DSSAligner &DA, bool Up) const
{
    if (!m_Opts.scores_are_not_evalues && DA.GetEvalue(Up) > m_MaxEvalue) return true;
    if (m_Opts.mints_set && DA.GetNewTestStatistic(Up) < m_Opts.mints) return true;
    return false;
}

void DBSearcher::BaseOnAln(DSSAligner &DA, bool Up)
{
    if (Reject(DA, Up)) return;
    std::lock_guard<std::mutex> g(m_Lock);
    ++m_HitCount;
    DA.ToTsv(m_fTsv, Up, m_Opts.noself);
    OnAln(DA, Up);
}

void DBSearcher::UploadToGpu()
{
    if (m_Db) return;
    if (!m_Ctx) throw std::runtime_error("DBSearcher: no GPU context");
    const uint n = GetDBChainCount();
    std::vector<uint32_t> len(n);
    std::vector<size_t> start((size_t) n + 1, 0);
    for (uint i = 0; i < n; ++i) { len[i] = m_DBChains[i]->GetSeqLength(); start[i + 1] = start[i] + len[i]; }
    const size_t tot = start[n];
    std::unique_ptr<uint8_t[]> mu(new uint8_t[tot + 1]), prof(new uint8_t[tot * RSK_NFEAT + 1]);      // filled below, not value-initialised
    std::unique_ptr<float[]> x(new float[tot + 1]), y(new float[tot + 1]), z(new float[tot + 1]);
    rsk_parallel_for(n, 512, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint L = len[i];
            const size_t o = start[i];
            memcpy(&mu[o], m_DBMuLettersVec[i]->data(), L);
            for (int f = 0; f < RSK_NFEAT; ++f) memcpy(&prof[o * RSK_NFEAT + (size_t) f * L], (*m_DBProfiles[i])[f].data(), L);
            memcpy(&x[o], m_DBChains[i]->m_Xs.data(), 4 * (size_t) L);
            memcpy(&y[o], m_DBChains[i]->m_Ys.data(), 4 * (size_t) L);
            memcpy(&z[o], m_DBChains[i]->m_Zs.data(), 4 * (size_t) L);
        }
    });
    check(rsk_db_create(m_Ctx, n, len.data(), mu.get(), prof.get(), x.get(), y.get(), z.get(), m_DBSelfRevScores.data(), &m_Db),
          "rsk_db_create");
    // residue characters: the statistics kernel counts the identical columns of an alignment (GetPctId) while it walks the path
    {
        std::unique_ptr<char[]> seq(new char[tot + 1]);
        rsk_parallel_for(n, 512, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) memcpy(&seq[start[i]], m_DBChains[i]->m_Seq.data(), len[i]);
        });
        check(rsk_db_set_seq(m_Db, seq.get()), "rsk_db_set_seq");
    }
}

// Align a batch of (ia, ib) pairs of one chain set on the GPU and replay the hits.
// One batch of (ia, ib) pairs: the GPU stage (AlignBatch: rsk_align_pairs) and the host stage (ReplayBatch: hit
// records -> Reject -> TSV lines).  RunPairs runs the GPU stage of batch k + 1 while batch k is replayed.
// Page-locked host buffers for the packed paths of a batch (hundreds of MB; the device-to-host copy into pageable
// memory was ~25 % of the GPU stage): two or three buffers are recycled between the batches of a run.
// A secondary context of this device, with a non-blocking stream of its own: its kernels run next to the primary
// context's (the long-chain job's X-drop tail under the alignment job's kernels, batch k + 1's uploads under batch k).
// Audited for this (r02): every entry point queues its copies / memsets / kernels on the context's stream and the host
// callers call rsk_ctx_sync before their own synchronous copies; chain sets are uploaded with synchronous copies before
// any context uses them.  RSK_OWN_STREAMS=0 puts every context back on the default stream.
struct SecondaryCtx {
    rsk_ctx *c = nullptr;
    hipStream_t st = nullptr;
    int device = -1;
    // Idle secondary contexts are kept per device and handed out again: a context's allocator pool holds the scratch of
    // its last job (tens of GB of X-drop trace, the SW trace blocks), and hipMalloc / hipFree of blocks that size cost
    // hundreds of ms -- per search and, with a streamed -db file, per batch.  rsk_ctx_trim() releases them.
    // A context is handed back to the ROLE it served (second alignment stage / long-chain job / -db loader): the roles'
    // scratch differs by orders of magnitude, and a 26 GB trace block that has to be allocated again costs 0.7 s on
    // some hosts.
    struct Idle { int device; rsk_ctx *c; hipStream_t st; const char *role; };
    const char *role = "";
    static std::mutex &Lock() { static std::mutex m; return m; }
    // The parked contexts keep their pools (that is the point), so two things bound what they can hold on to: the library's
    // out-of-memory ladder (rsk_dev_malloc) destroys the idle contexts of the device before any allocation fails -- the
    // hook is registered with the list -- and the list is destroyed with the process.
    struct IdleHolder {
        std::vector<Idle> v;
        IdleHolder() { rsk_set_oom_hook(&SecondaryCtx::Trim); }
        // Static destruction runs in an unspecified order relative to the HIP runtime's own teardown: no HIP call here.
        // The driver reclaims the parked contexts with the process; rsk_ctx_trim(ctx) / rsk_shutdown() release them earlier.
        ~IdleHolder() { rsk_set_oom_hook(nullptr); }
    };
    static std::vector<Idle> &IdleList() { static IdleHolder h; return h.v; }
    void Create(int dev, const char *Role)
    {
        device = dev;
        role = Role;
        {
            std::lock_guard<std::mutex> g(Lock());
            auto &v = IdleList();
            for (size_t k = 0; k < v.size(); ++k)
                if (v[k].device == dev && strcmp(v[k].role, Role) == 0) { c = v[k].c; st = v[k].st; v.erase(v.begin() + k); return; }
        }
        rsk_device_guard on(dev);                                       // the calling thread keeps ITS current device (a stream belongs to the device current at creation)
        check(rsk_ctx_create(dev, &c), "rsk_ctx_create");
        if (!(getenv("RSK_OWN_STREAMS") && atoi(getenv("RSK_OWN_STREAMS")) == 0)) {
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return; }
            rsk_ctx_set_stream(c, (void *) st);
        }
    }
    ~SecondaryCtx()
    {
        if (!c) return;
        (void) rsk_ctx_sync(c);
        std::lock_guard<std::mutex> g(Lock());
        auto &v = IdleList();
        if (v.size() < 6) { v.push_back(Idle{ device, c, st, role }); return; }
        rsk_ctx_destroy(c);
        if (st) (void) hipStreamDestroy(st);
    }
    static void Trim(int dev)
    {
        std::lock_guard<std::mutex> g(Lock());
        auto &v = IdleList();
        for (size_t k = 0; k < v.size();)
            if (dev < 0 || v[k].device == dev) {
                rsk_ctx_destroy(v[k].c);
                if (v[k].st) (void) hipStreamDestroy(v[k].st);
                v.erase(v.begin() + k);
            } else ++k;
    }
};

struct PinnedPool {
    std::mutex lock;
    std::vector<std::pair<char *, size_t> > idle;
    char *Get(size_t bytes, size_t &cap)
    {
        {
            std::lock_guard<std::mutex> g(lock);
            for (size_t k = 0; k < idle.size(); ++k)
                if (idle[k].second >= bytes) {
                    char *p = idle[k].first;
                    cap = idle[k].second;
                    idle.erase(idle.begin() + k);
                    return p;
                }
            if (!idle.empty()) { (void) hipHostFree(idle.back().first); idle.pop_back(); }      // too small: replace it
        }
        void *p = nullptr;
        cap = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) throw std::runtime_error("hipHostMalloc failed for the path buffer");
        return (char *) p;
    }
    void Put(char *p, size_t cap)
    {
        std::lock_guard<std::mutex> g(lock);
        idle.emplace_back(p, cap);
    }
    ~PinnedPool() { for (auto &b : idle) (void) hipHostFree(b.first); }
};

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
    for (size_t k = 0; k < ia.size(); ++k) {
        const uint64_t c = (uint64_t) la[ia[k]] * lb[ib[k]];
        if (k > b && (k - b >= maxp || cells + c > maxc)) { out.emplace_back(b, k); b = k; cells = 0; }
        cells += c;
    }
    if (b < ia.size()) out.emplace_back(b, ia.size());
    return out;
}

// rsk_align_pairs over the batches of a pair list with the GPU stage of batch k + 1 running while `OnBatch` consumes
// batch k on the calling thread (RunPairs: hit replay; PostMuFilter: Accept + hit lines).
void ForEachAlignedBatch(const DSSParams &P, rsk_ctx *ctx, const SearchOptions &O, DBSearcher &SrcA, DBSearcher &SrcB,
                         const std::vector<uint32_t> &ia, const std::vector<uint32_t> &ib,
                         const std::function<void(const std::vector<uint32_t> &, const std::vector<uint32_t> &, const std::vector<rsk_aln> &,
                                                  const char *)> &OnBatch)
{
    const auto batches = AlignBatches(O, SrcA, SrcB, ia, ib);
    PinnedPool Pool;                                                     // outlives every batch of the loop below
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
            return AlignBatch(P, c, Pool, SrcA, SrcB, std::vector<uint32_t>(ia.begin() + be.first, ia.begin() + be.second),
                              std::vector<uint32_t>(ib.begin() + be.first, ib.begin() + be.second));
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
static void RunPairs(DBSearcher &S, DBSearcher &SrcA, bool Self, int64_t SelfOffset = -1)
{
    PhaseTimer tm;
    const DSSParams &P = *S.m_Params;
    rsk_ctx *ctx = S.m_Ctx;
    const uint NA = SrcA.GetDBChainCount(), NB = S.GetDBChainCount();
    const bool UseMu = P.m_Omega > 0;            // LoadDB keeps Mu letters only when Omega > 0 (dbsearcher.cpp:249-251)
    const bool Tri = Self && SelfOffset < 0;     // the whole triangle in one call
    const uint joff = SelfOffset > 0 ? (uint) SelfOffset : 0;
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
    if (Self) for (uint j = 0; j < NB; ++j) SelfTotal += std::min<uint64_t>(NA, (uint64_t) joff + j + 1);
    std::vector<uint32_t> ia, ib;                // pairs for the full alignment
    std::vector<std::pair<uint32_t, uint32_t> > mkf;
    uint64_t npairs = 0;
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
        std::vector<uint32_t> longB;
        for (uint j = 0; j < NB; ++j)
            if (!S.m_DBMuKmersVec[j]->empty() && S.m_DBChains[j]->GetSeqLength() >= P.m_MKFL) longB.push_back(j);
        for (uint i = 0; i < NA; ++i) {
            if (SrcA.m_DBMuKmersVec[i]->empty()) continue;
            const uint j0 = Self ? (i > joff ? i - joff : 0) : 0;
            if (SrcA.m_DBChains[i]->GetSeqLength() >= P.m_MKFL) {
                for (uint j = j0; j < NB; ++j) {
                    if (S.m_DBMuKmersVec[j]->empty() || Skip(i, j)) continue;
                    mkf.emplace_back(i, j); ++nmkf;
                }
            } else {
                for (auto it = std::lower_bound(longB.begin(), longB.end(), j0); it != longB.end(); ++it) {
                    if (Skip(i, *it)) continue;
                    mkf.emplace_back(i, *it); ++nmkf;
                }
            }
        }
        if (S.m_Opts.noself) {
            if (Self) nskip = NA > joff ? std::min<uint64_t>(NB, NA - joff) : 0;      // the diagonal pairs this pass holds
            else {
                std::unordered_map<std::string, uint32_t> cntB;
                for (uint j = 0; j < NB; ++j) ++cntB[S.m_DBChains[j]->m_Label];
                for (uint i = 0; i < NA; ++i) {
                    auto it = cntB.find(SrcA.m_DBChains[i]->m_Label);
                    if (it != cntB.end()) nskip += it->second;
                }
            }
        }
        // The long-chain job does not depend on the filter (its pairs are known from the chain lengths): it starts NOW on a
        // context of its own and runs under the filter kernels of this one (r04; r01-r03 started it after the filter, beside
        // the alignment job only).
        const bool early = !(getenv("RSK_MKF_EARLY") && atoi(getenv("RSK_MKF_EARLY")) == 0);      // 0: start it after the filter (r03 behaviour, for A/B timing)
        if (early) start_mkf_job();
        tm.lap("  long-chain pair list (job started)");
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
            check(rsk_mu_filter_dev(ctx, FilterQ, FilterT, Tri ? 1 : 0, P.m_ParaMuGapOpen, P.m_ParaMuGapExt, P.m_Omega, P.m_OmegaFwd, Fwd.As<uint8_t>(), ldo,
                                    ListQ.As<uint32_t>(), ListT.As<uint32_t>(), nullptr, nullptr, cap, Count.As<uint32_t>()),
                  "rsk_mu_filter_dev");
            check(rsk_ctx_sync(ctx), "rsk_ctx_sync");                  // the filter is queued on the context's stream; the copies below are not
            hipok(hipMemcpy(&ns, Count.As<uint32_t>(), 4, hipMemcpyDeviceToHost), "copy n");
            if (ns <= cap) break;
            cap = ns;
        }
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
        npairs = total - nskip;
        S.m_MKFPairCount = nmkf;
        S.m_MuFilterInputCount = npairs - nmkf;
        S.m_MuFilterDiscardCount = S.m_MuFilterInputCount - ia.size();
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
        ForEachAlignedBatch(P, ctx, S.m_Opts, SrcA, S, ia, ib,
                            [&](const std::vector<uint32_t> &bia, const std::vector<uint32_t> &bib, const std::vector<rsk_aln> &out, const char *paths) {
                                ReplayBatch(S, SrcA, S, bia, bib, out, paths, Self, joff);
                            });
    };
    if (!job.valid()) start_mkf_job();             // (RSK_MKF_EARLY=0, or no filter in this mode)
    if (job.valid()) {
        align();
        job.get();
        tm.lap("align + replay | long-chain job side by side (started before the filter)");
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
// One context per entry of the device list, on streams of their own; parked between calls like every helper context.
struct DeviceTeam {
    std::vector<std::unique_ptr<SecondaryCtx> > member;
    explicit DeviceTeam(const std::vector<int> &Devices)
    {
        static const char *const roles[] = { "team0", "team1", "team2", "team3", "team4", "team5", "team6", "team7" };
        for (size_t k = 0; k < Devices.size(); ++k) {
            member.emplace_back(new SecondaryCtx);
            member.back()->Create(Devices[k], roles[k % 8]);
        }
    }
    rsk_ctx *ctx(size_t k) const { return member[k]->c; }
};

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

// shard `Index` of `Count` of the self search: targets [Lo, Hi) chosen so that the cells of the triangle are balanced
void DBSearcher::RunSelfShard(uint Index, uint Count)
{
    const uint N = GetDBChainCount();
    std::vector<uint32_t> Lens(N);
    for (uint j = 0; j < N; ++j) Lens[j] = m_DBChains[j]->GetSeqLength();
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
        try {
            Src->LoadChains(Chains);
        } catch (...) {
            for (PDBChain *C : Chains) delete C;
            throw;
        }
        Src->UploadToGpu();                              // here, beside the search of the previous batch (synchronous copies)
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

extern "C" int rsk_shard_range(int kind, const uint32_t *lengths, uint64_t n, uint32_t index, uint32_t count, uint64_t *lo, uint64_t *hi)
{
    if ((n && !lengths) || !lo || !hi || count == 0 || index >= count || kind < 0 || kind > 1) { rsk_set_error("rsk_shard_range: invalid argument"); return RSK_E_INVALID; }
    if (kind == 0) reseek_amd::DBSearcher::SelfShardRange(lengths, n, index, count, *lo, *hi);
    else reseek_amd::DBSearcher::ResidueShardRange(lengths, n, index, count, *lo, *hi);
    return RSK_OK;
}

extern "C" void rsk_ctx_trim(rsk_ctx *ctx)
{
    if (!ctx) return;
    reseek_amd::SecondaryCtx::Trim(ctx->device);
    rsk_pool_release(ctx);
}

// ---------------------------------------------------------------------------------------------
// C-ABI: `reseek -search Q [-db DB] -fast|-sensitive|-verysensitive -output F [-columns C] [-evalue E]`
// ---------------------------------------------------------------------------------------------
using namespace reseek_amd;

static bool parse_mode(const char *mode, SearchOptions &o)
{
    const std::string m = mode ? mode : "";
    if (m == "fast") o.mode = AM_Fast;
    else if (m == "sensitive") o.mode = AM_Sensitive;
    else if (m == "verysensitive") o.mode = AM_VerySensitive;
    else return false;
    return true;
}

void rsk_set_error(const char *fmt, ...);
namespace reseek_amd {
// rsk_search_opts -> SearchOptions.  Reads no member beyond opts->struct_size (members appended to the struct by later
// headers read as "not given" for a caller built with an older one).
int ParseSearchOpts(const rsk_search_opts *opts, SearchOptions &o, const char *who)
{
    const size_t have = opts->struct_size;
#define RSK_OPT_HAS(f) (have >= offsetof(rsk_search_opts, f) + sizeof(opts->f))
    if (!RSK_OPT_HAS(mode) || have > 4096) {
        rsk_set_error("%s: opts.struct_size = %zu; set it to sizeof(rsk_search_opts) (first member since ABI 4)", who, have);
        return RSK_E_INVALID;
    }
    if (!parse_mode(opts->mode, o)) { rsk_set_error("%s: mode must be fast, sensitive or verysensitive", who); return RSK_E_INVALID; }
    if (RSK_OPT_HAS(columns) && opts->columns) o.columns = opts->columns;
    if (RSK_OPT_HAS(evalue_set) && opts->evalue_set) { o.evalue_set = true; o.evalue = opts->evalue; }
    if (RSK_OPT_HAS(mints_set) && opts->mints_set) { o.mints_set = true; o.mints = opts->mints; }
    if (RSK_OPT_HAS(pvalue_set) && opts->pvalue_set) { o.pvalue_set = true; o.pvalue = opts->pvalue; }
    if (RSK_OPT_HAS(noself)) o.noself = opts->noself != 0;
    if (RSK_OPT_HAS(selfrev0)) o.selfrev0 = opts->selfrev0 != 0;
    if (RSK_OPT_HAS(idx_mode)) {
        if (opts->idx_mode < 0 || opts->idx_mode > 2) { rsk_set_error("%s: idx_mode must be 0, 1 or 2", who); return RSK_E_INVALID; }
        o.idx_mode = opts->idx_mode == 0 ? -1 : opts->idx_mode;
    }
    if (RSK_OPT_HAS(rsb_size) && opts->rsb_size) o.rsb_size = opts->rsb_size;
    if (RSK_OPT_HAS(dbmu) && opts->dbmu) o.dbmu = opts->dbmu;
    if (RSK_OPT_HAS(keeptmp)) o.keeptmp = opts->keeptmp != 0;
    if (RSK_OPT_HAS(shard_index)) o.shard_index = opts->shard_index;
    if (RSK_OPT_HAS(shard_count)) o.shard_count = opts->shard_count;
    if (RSK_OPT_HAS(devices) && opts->devices) o.devices = opts->devices;
    if (RSK_OPT_HAS(hits_digest)) o.hits_digest = opts->hits_digest != 0;
#undef RSK_OPT_HAS
    return RSK_OK;
}
void FastDbOnContexts(const std::vector<rsk_ctx *> &Ctx, const char *query_path, const char *db_path, const SearchOptions &o, const char *out_tsv,
                      const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8);
}

// rsk_search_opts.hits_digest: the hit lines go to a digest instead of a file.  A `-verysensitive` search of 1k queries
// against a PDB-sized DB writes 7e8 lines (30 GB); to compare the union of 8 shards with the unsharded table only an
// order-independent summary is needed: number of lines, their bytes, and the sum and xor of a 64-bit hash of every line.
// The FILE the searchers write to is a glibc cookie stream that cuts the byte stream at newlines (stdio's buffer
// boundaries are arbitrary) and hashes each line; out_tsv then receives ONE line "digest\t<lines>\t<bytes>\t<sum>\t<xor>".
namespace {
struct HitsDigest {
    uint64_t lines = 0, bytes = 0, sum = 0, x = 0;
    std::string carry;
    static uint64_t hash_line(const char *p, size_t n)
    {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
        auto mix = [&](uint64_t v) { h = (h ^ v) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; };
        for (; n >= 8; p += 8, n -= 8) { uint64_t v; memcpy(&v, p, 8); mix(v); }
        if (n) { uint64_t v = 0; memcpy(&v, p, n); mix(v); }
        h *= 0xC4CEB9FE1A85EC53ull;
        return h ^ (h >> 29);
    }
    void line(const char *p, size_t n) { const uint64_t h = hash_line(p, n); ++lines; bytes += n + 1; sum += h; x ^= h; }
    void feed(const char *p, size_t n)
    {
        const char *end = p + n;
        if (!carry.empty()) {
            const char *nl = (const char *) memchr(p, '\n', n);
            if (!nl) { carry.append(p, n); return; }
            carry.append(p, (size_t) (nl - p));
            line(carry.data(), carry.size());
            carry.clear();
            p = nl + 1;
        }
        while (p < end) {
            const char *nl = (const char *) memchr(p, '\n', (size_t) (end - p));
            if (!nl) { carry.assign(p, (size_t) (end - p)); return; }
            line(p, (size_t) (nl - p));
            p = nl + 1;
        }
    }
    static ssize_t cookie_write(void *c, const char *buf, size_t n) { ((HitsDigest *) c)->feed(buf, n); return (ssize_t) n; }
    FILE *open()
    {
        cookie_io_functions_t io = {};
        io.write = &HitsDigest::cookie_write;
        FILE *f = fopencookie(this, "w", io);
        if (f) setvbuf(f, nullptr, _IOFBF, 8u << 20);
        return f;
    }
};
struct FileCloser { FILE *f; ~FileCloser() { if (f) fclose(f); } };
}   // namespace

static bool keep_tmp_env() { const char *e = getenv("RSK_KEEPTMP"); return e && *e && *e != '0'; }    // -keeptmp

static int search_impl(rsk_ctx *ctx, const char *query_rskdb, const char *db_rskdb, const SearchOptions &o, const char *out_tsv,
                       uint64_t *nhits, uint64_t *stats8)
{
    try {
        DSSParams Params;
        Params.SetDSSParams(o);
        const bool have_db = db_rskdb != nullptr && *db_rskdb;
        const bool prefilter_path = have_db && o.mode == AM_Fast;        // search.cpp:76-111
        SearchOptions o2 = o;
        o2.mode = AM_Sensitive;                  // DM_AlwaysSensitive dssparams.cpp:27-42
        DSSParams Params2;
        Params2.SetDSSParams(o2);
        const std::vector<int> devs = DBSearcher::ParseDeviceList(o.devices.empty() ? getenv("RSK_DEVICES") : o.devices.c_str());
        // (the several-device form streams its target shards from a .bca file; any other -db container keeps the one-device
        // two-stage path below, which takes both -- a device list must not make a call fail that works without it)
        const bool db_is_bca = have_db && std::string(db_rskdb).size() >= 4 && std::string(db_rskdb).compare(std::string(db_rskdb).size() - 4, 4, ".bca") == 0;
        if (prefilter_path && devs.size() > 1 && o.shard_count <= 1 && db_is_bca) {
            // the two-stage path on several devices: one target shard per context, the top-B exchange in host memory
            DeviceTeam Team(devs);
            std::vector<rsk_ctx *> cs;
            for (size_t k = 0; k < devs.size(); ++k) cs.push_back(Team.ctx(k));
            const std::string tmp = std::string(out_tsv) + ".prefilter.tmp";
            const bool keep = o.keeptmp || keep_tmp_env();
            FastDbOnContexts(cs, query_rskdb, db_rskdb, o, out_tsv, keep ? tmp.c_str() : nullptr, nhits, stats8);
            return RSK_OK;
        }
        if (devs.size() == 1 && devs[0] != ctx->device) {
            // a one-entry list names THE device of the call: the search runs on a helper context there
            DeviceTeam Team(devs);
            SearchOptions o1 = o;
            o1.devices = std::to_string(devs[0]);
            return search_impl(Team.ctx(0), query_rskdb, db_rskdb, o1, out_tsv, nhits, stats8);      // (o1.devices set: the environment is not consulted again)
        }
        DBSearcher DBS;                       // SelfSearch search.cpp:20-37 / Search_NoMuFilter :39-60
        DBS.m_Params = prefilter_path ? &Params2 : &Params;
        DBS.m_SelfRevQueryFlavour = prefilter_path;      // PostMuFilter computes query self-rev scores itself (postmufilter.cpp:79)
        DBS.m_Opts = o;
        DBS.m_Ctx = ctx;
        if (!o.devices.empty()) DBS.m_Devices = DBSearcher::ParseDeviceList(o.devices.c_str());
        DBS.LoadDB(query_rskdb);
        DBS.Setup();
        for (USERFIELD u : DBS.m_DA.m_UFs)
            if (u == UF_Undefined) { rsk_set_error("rsk_search_rskdb: invalid -columns field"); return RSK_E_INVALID; }
        if (prefilter_path && o.hits_digest) { rsk_set_error("rsk_search: hits_digest is not available on the -fast -db path"); return RSK_E_INVALID; }
        if (prefilter_path && o.shard_count > 1) {
            rsk_set_error("rsk_search: shards are not supported on the -fast -db path (the per-query top-B of the prefilter is a reduction over all targets)");
            return RSK_E_INVALID;
        }
        if (prefilter_path) {
            // cmd_search search.cpp:76-111: k-mer prefilter, then the candidates under the "sensitive" preset
            DBSearcher Src;
            Src.m_Params = &Params2;
            Src.m_SelfRevQueryFlavour = true;            // postmufilter.cpp:171
            Src.m_Opts = o;
            Src.m_Ctx = ctx;
            // `-search X -db X`: the two sides are the same file read under the same parameters and the same self-rev
            // flavour (both stages of cmd_search load it with DM_AlwaysSensitive) -- one load, the DB side is a view of it
            if (std::string(db_rskdb) == std::string(query_rskdb)) Src.MakeView(DBS, 0, DBS.GetDBChainCount());
            else Src.LoadDB(db_rskdb);
            // the candidates go from stage to stage in memory, in the hand-off file's order; the file itself
            // (rankedscoresbag.cpp:185-231) is written for -keeptmp only
            const std::string tmp = std::string(out_tsv) + ".prefilter.tmp";
            std::vector<uint32_t> pq, pt;
            MuPreFilterToPairs(DBS, Src, pq, pt, o.keeptmp || keep_tmp_env() ? tmp : std::string());
            if (pq.empty()) fprintf(stderr, "Warning: No hits found by mufilter pass\n");      // postmufilter.cpp:219-223 (no hits file)
            else PostMuFilterPairs(Params2, DBS, Src, pq, pt, out_tsv);
            if (nhits) *nhits = DBS.m_HitCount;
            if (stats8) {
                stats8[0] = DBS.m_ProcessedPairCount; stats8[1] = DBS.m_ProcessedPairCount - DBS.m_MKFPairCount; stats8[2] = DBS.m_MuFilterInputCount;
                stats8[3] = DBS.m_MuFilterDiscardCount; stats8[4] = DBS.m_MKFPairCount; stats8[5] = DBS.m_SWCount;
                stats8[6] = DBS.m_HitCount; stats8[7] = 1;
            }
            return RSK_OK;
        }
        HitsDigest Digest;
        FILE *f = o.hits_digest ? Digest.open() : fopen(out_tsv, "w");
        if (!f) { rsk_set_error("rsk_search_rskdb: cannot create %s", out_tsv); return RSK_E_INVALID; }
        FileCloser closer{ f };                          // closed on every exit path
        DBS.m_fTsv = f;
        if (o.shard_count > 1 && o.shard_index >= o.shard_count) { rsk_set_error("rsk_search: shard_index >= shard_count"); return RSK_E_INVALID; }
        if (!have_db) {
            if (o.shard_count > 1) DBS.RunSelfShard(o.shard_index, o.shard_count);
            else DBS.RunSelf();
        } else {
            const std::string dbfn = db_rskdb;
            if (dbfn.size() >= 4 && dbfn.compare(dbfn.size() - 4, 4, ".bca") == 0) {
                // Search_NoMuFilter search.cpp:39-60: the -db file streams through a ChainReader2
                ChainReader2 CR;
                if (o.shard_count > 1) {
                    // -db mode: contiguous target shards balanced by residues, the query set is replicated
                    BCAData B;
                    B.Open(dbfn);
                    uint64_t Lo, Hi;
                    DBSearcher::ResidueShardRange(B.m_SeqLengths.data(), B.GetChainCount(), o.shard_index, o.shard_count, Lo, Hi);
                    CR.OpenRange(dbfn, Lo, Hi);
                } else
                    CR.Open(dbfn);
                if (const char *e = getenv("RSK_STREAM_CHAINS")) { const long v = atol(e); if (v > 0) DBS.m_StreamBatchChains = (uint) v; }
                DBS.RunQuery(CR);
            } else {
            DBSearcher Src;
            Src.m_Params = &Params;
            Src.m_SelfRevQueryFlavour = true;            // runquery.cpp:43-44
            Src.m_Opts = o;
            Src.m_Ctx = ctx;
            Src.LoadDB(db_rskdb);
            if (o.shard_count > 1) {
                // -db mode: contiguous target shards balanced by residues, the query set is replicated
                const uint NS = Src.GetDBChainCount();
                std::vector<uint32_t> Lens(NS);
                for (uint i = 0; i < NS; ++i) Lens[i] = Src.m_DBChains[i]->GetSeqLength();
                uint64_t Lo, Hi;
                DBSearcher::ResidueShardRange(Lens.data(), NS, o.shard_index, o.shard_count, Lo, Hi);
                DBSearcher View;
                View.MakeView(Src, (uint) Lo, (uint) Hi);
                if (Hi > Lo) DBS.RunQuery(View);
            } else
                DBS.RunQuery(Src);
            }
        }
        closer.f = nullptr;
        if (fclose(f) != 0) { rsk_set_error("rsk_search: writing %s failed", out_tsv); return RSK_E_INVALID; }
        if (o.hits_digest) {
            if (!Digest.carry.empty()) Digest.line(Digest.carry.data(), Digest.carry.size());
            FILE *g = fopen(out_tsv, "w");
            if (!g) { rsk_set_error("rsk_search: cannot create %s", out_tsv); return RSK_E_INVALID; }
            fprintf(g, "digest\t%llu\t%llu\t%016llx\t%016llx\n", (unsigned long long) Digest.lines, (unsigned long long) Digest.bytes,
                    (unsigned long long) Digest.sum, (unsigned long long) Digest.x);
            fclose(g);
        }
        if (nhits) *nhits = DBS.m_HitCount;
        if (stats8) {
            stats8[0] = DBS.m_ProcessedPairCount; stats8[1] = DBS.m_AlnCount; stats8[2] = DBS.m_MuFilterInputCount;
            stats8[3] = DBS.m_MuFilterDiscardCount; stats8[4] = DBS.m_MKFPairCount; stats8[5] = DBS.m_SWCount;
            stats8[6] = DBS.m_HitCount; stats8[7] = 0;
        }
    } catch (const std::exception &e) {
        rsk_set_error("rsk_search_rskdb: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_search_rskdb(rsk_ctx *ctx, const char *query_rskdb, const char *db_rskdb, const char *mode, const char *columns,
                                double evalue, int noself, const char *out_tsv, uint64_t *nhits, uint64_t *stats8)
{
    if (!ctx || !query_rskdb || !out_tsv) { rsk_set_error("rsk_search_rskdb: NULL argument"); return RSK_E_INVALID; }
    SearchOptions o;
    if (!parse_mode(mode, o)) { rsk_set_error("rsk_search_rskdb: mode must be fast, sensitive or verysensitive"); return RSK_E_INVALID; }
    if (columns) o.columns = columns;
    if (evalue >= 0) { o.evalue_set = true; o.evalue = evalue; }
    o.noself = noself != 0;
    return search_impl(ctx, query_rskdb, db_rskdb, o, out_tsv, nhits, stats8);
}

extern "C" int rsk_search(rsk_ctx *ctx, const char *query_path, const char *db_path, const rsk_search_opts *opts, const char *out_tsv,
                          uint64_t *nhits, uint64_t *stats8)
{
    if (!ctx || !query_path || !out_tsv || !opts) { rsk_set_error("rsk_search: NULL argument"); return RSK_E_INVALID; }
    SearchOptions o;
    const int rc = reseek_amd::ParseSearchOpts(opts, o, "rsk_search");
    if (rc != RSK_OK) return rc;
    return search_impl(ctx, query_path, db_path, o, out_tsv, nhits, stats8);
}

extern "C" int rsk_abi_version(void) { return RSK_ABI_VERSION; }

extern "C" void rsk_shutdown(void) { reseek_amd::SecondaryCtx::Trim(-1); }

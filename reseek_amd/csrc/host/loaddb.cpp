// loaddb.cpp -- DBSearcher: chain sets in host memory and on the device (ProfileLoader::Load profileloader.cpp:72,
// DBSearcher::LoadDB / Setup dbsearcher.cpp:40-110): .bca / .rskdb reading, DSS featurisation of a batch (densities, SS,
// Conf letters, neighbours from the device: k_dss.hip), self-rev scores (alignpair.cpp:7) as one device batch, upload.
#include "host_internal.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace reseek_amd {
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
        // ~16 heap blocks per chain: released on the host threads (one thread: 20 ms for 11,211 chains, the tail of every call)
        const size_t n = std::max(std::max(m_DBChains.size(), m_DBProfiles.size()), std::max(m_DBMuLettersVec.size(), m_DBMuKmersVec.size()));
        rsk_parallel_for(n, 256, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                if (i < m_DBChains.size()) delete m_DBChains[i];
                if (i < m_DBProfiles.size()) delete m_DBProfiles[i];
                if (i < m_DBMuLettersVec.size()) delete m_DBMuLettersVec[i];
                if (i < m_DBMuKmersVec.size()) delete m_DBMuKmersVec[i];
            }
        });
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
    Kmers.reserve(L >= 3 ? L - 2 : 0);
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
    // RSKDB1 container: the file is mapped, a serial pass finds the chains' records (three counts per
    // record), the chain objects are built on the host threads.  (r01-r04b read it with ~20 fread calls and as many
    // allocations per chain, one chain after the other: 70 ms of a 0.43 s all-vs-all call.)
    PhaseTimer tm("LoadDB");
    // (mapped, not copied: the records are parsed out of the page cache)
    struct Mapped {
        const char *p = nullptr; size_t n = 0; int fd = -1;
        ~Mapped() { if (p) munmap((void *) p, n); if (fd >= 0) close(fd); }
        const char *data() const { return p; }
        size_t size() const { return n; }
    } buf;
    {
        buf.fd = open(DBFN.c_str(), O_RDONLY);
        if (buf.fd < 0) throw std::runtime_error("LoadDB: cannot open " + DBFN);
        struct stat sb;
        if (fstat(buf.fd, &sb) != 0 || sb.st_size < 16) throw std::runtime_error("LoadDB: truncated " + DBFN);
        void *m = mmap(nullptr, (size_t) sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, buf.fd, 0);
        if (m == MAP_FAILED) throw std::runtime_error("LoadDB: cannot map " + DBFN);
        buf.p = (const char *) m;
        buf.n = (size_t) sb.st_size;
    }
    if (memcmp(buf.data(), "RSKDB1\0\0", 8) != 0) throw std::runtime_error("LoadDB: " + DBFN + " is not an RSKDB1 container");
    uint32_t n, nfeat;
    memcpy(&n, buf.data() + 8, 4);
    memcpy(&nfeat, buf.data() + 12, 4);
    if (nfeat != RSK_NFEAT) throw std::runtime_error("LoadDB: feature count mismatch");
    // a record is at least 16 bytes (L, label length, self-rev score, k-mer count): a chain count the file cannot hold is an
    // error of the call before anything is sized by it
    if ((size_t) n > (buf.size() - 16) / 16) throw std::runtime_error("LoadDB: truncated " + DBFN + " (chain count exceeds the file)");
    // record: L, label length, label, residue characters[L], Mu letters[L], profile[nfeat][L], x[L], y[L], z[L] (float),
    // self-rev score, k-mer count, k-mers
    std::vector<size_t> rec((size_t) n + 1);
    {
        size_t o = 16;
        for (uint32_t k = 0; k < n; ++k) {
            rec[k] = o;
            if (o + 8 > buf.size()) throw std::runtime_error("LoadDB: truncated " + DBFN);
            uint32_t L, ll, nk;
            memcpy(&L, buf.data() + o, 4);
            memcpy(&ll, buf.data() + o + 4, 4);
            const size_t body = (size_t) ll + (size_t) L * (2 + nfeat) + 12 * (size_t) L + 4;
            if (o + 8 + body + 4 > buf.size()) throw std::runtime_error("LoadDB: truncated " + DBFN);
            memcpy(&nk, buf.data() + o + 8 + body, 4);
            o += 8 + body + 4 + 4 * (size_t) nk;
            if (o > buf.size()) throw std::runtime_error("LoadDB: truncated " + DBFN);
        }
        rec[n] = o;
    }
    const size_t base = m_DBChains.size();
    m_DBChains.resize(base + n); m_DBProfiles.resize(base + n); m_DBMuLettersVec.resize(base + n);
    m_DBMuKmersVec.resize(base + n); m_DBSelfRevScores.resize(base + n);
    std::atomic<bool> bad{false}, bad_letters{false};
    rsk_parallel_for(n, 64, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) {
            const char *p = buf.data() + rec[k];
            uint32_t L, ll;
            memcpy(&L, p, 4); memcpy(&ll, p + 4, 4);
            p += 8;
            PDBChain *C = new PDBChain;
            C->m_Label.assign(p, ll); p += ll;
            C->m_Seq.assign(p, L); p += L;
            auto *Mu = new std::vector<byte>((const byte *) p, (const byte *) p + L); p += L;
            auto *Prof = new std::vector<std::vector<byte> >(nfeat);
            for (uint32_t fi = 0; fi < nfeat; ++fi) { (*Prof)[fi].assign((const byte *) p, (const byte *) p + L); p += L; }
            // letters index the device tables (36 x 36 Mu matrix, 20 / 16-letter feature tables): out-of-range letters are an
            // error of the call, not an out-of-bounds read in a kernel
            {
                byte worst = 0;
                for (byte c : *Mu) worst = std::max(worst, c);
                if (worst >= 36) bad_letters = true;
                for (uint32_t fi = 0; fi < nfeat; ++fi) {
                    worst = 0;
                    for (byte c : (*Prof)[fi]) worst = std::max(worst, c);
                    if (worst >= (fi == 0 ? 20 : 16)) bad_letters = true;
                }
            }
            C->m_Xs.resize(L); C->m_Ys.resize(L); C->m_Zs.resize(L);
            memcpy(C->m_Xs.data(), p, 4 * (size_t) L); p += 4 * (size_t) L;
            memcpy(C->m_Ys.data(), p, 4 * (size_t) L); p += 4 * (size_t) L;
            memcpy(C->m_Zs.data(), p, 4 * (size_t) L); p += 4 * (size_t) L;
            float selfrev;
            memcpy(&selfrev, p, 4); p += 4;
            uint32_t nk;
            memcpy(&nk, p, 4); p += 4;
            auto *Kmers = new std::vector<uint>;
            GetMuKmers(*Mu, *Kmers);
            if (Kmers->size() != nk || (nk && memcmp(Kmers->data(), p, 4 * (size_t) nk) != 0)) bad = true;
            C->m_Idx = (uint) (base + k);
            m_DBChains[base + k] = C;
            m_DBProfiles[base + k] = Prof;
            m_DBMuLettersVec[base + k] = Mu;
            m_DBMuKmersVec[base + k] = Kmers;
            m_DBSelfRevScores[base + k] = m_Opts.selfrev0 ? 0.0f : selfrev;
        }
    });
    if (bad || bad_letters) {
        // the call fails as a whole: the chains it appended are released again
        for (size_t k = base; k < base + n; ++k) {
            delete m_DBChains[k]; delete m_DBProfiles[k]; delete m_DBMuLettersVec[k]; delete m_DBMuKmersVec[k];
        }
        m_DBChains.resize(base); m_DBProfiles.resize(base); m_DBMuLettersVec.resize(base);
        m_DBMuKmersVec.resize(base); m_DBSelfRevScores.resize(base);
        throw std::runtime_error(bad_letters ? "LoadDB: " + DBFN + " holds letters outside the Mu / feature alphabets"
                                             : "LoadDB: stored Mu k-mers disagree with the letters");
    }
    tm.lap("read + build chains");
}

// The inverse of LoadDB's container route: header "RSKDB1\0\0", chain count, feature count, then per chain L, label length, label,
// residue characters[L], Mu letters[L], profile[8][L], x[L], y[L], z[L] (float), self-rev score, k-mer count, k-mers.
void DBSearcher::WriteRskdb(const std::string &FN) const
{
    FILE *f = fopen(FN.c_str(), "wb");
    if (!f) throw std::runtime_error("WriteRskdb: cannot create " + FN);
    struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } closer{ f };
    std::string out;
    const uint32_t n = GetDBChainCount(), nfeat = RSK_NFEAT;
    out.append("RSKDB1\0\0", 8);
    out.append((const char *) &n, 4);
    out.append((const char *) &nfeat, 4);
    for (uint32_t k = 0; k < n; ++k) {
        const PDBChain &C = *m_DBChains[k];
        const uint32_t L = C.GetSeqLength(), ll = (uint32_t) C.m_Label.size();
        const std::vector<byte> &Mu = *m_DBMuLettersVec[k];
        const std::vector<std::vector<byte> > &Prof = *m_DBProfiles[k];
        if (Prof.size() != nfeat) throw std::runtime_error("WriteRskdb: chain without a profile");
        out.append((const char *) &L, 4);
        out.append((const char *) &ll, 4);
        out.append(C.m_Label);
        out.append(C.m_Seq.data(), L);
        if (Mu.size() == L) out.append((const char *) Mu.data(), L);
        else {
            // (-verysensitive keeps no Mu letters, dbsearcher.cpp:249-251: the container's Mu row is what the featuriser gives)
            std::vector<byte> M;
            DSS D;
            D.SetParams(*m_Params);
            D.Init(C);
            D.GetMuLetters(M);
            out.append((const char *) M.data(), L);
        }
        for (uint32_t fi = 0; fi < nfeat; ++fi) out.append((const char *) Prof[fi].data(), L);
        out.append((const char *) C.m_Xs.data(), 4 * (size_t) L);
        out.append((const char *) C.m_Ys.data(), 4 * (size_t) L);
        out.append((const char *) C.m_Zs.data(), 4 * (size_t) L);
        const float sr = m_DBSelfRevScores[k];
        out.append((const char *) &sr, 4);
        std::vector<uint> Kmers;
        {
            std::vector<byte> M((const byte *) out.data() + out.size() - 4 - 12 * (size_t) L - (size_t) nfeat * L - L,
                                (const byte *) out.data() + out.size() - 4 - 12 * (size_t) L - (size_t) nfeat * L);
            GetMuKmers(M, Kmers);
        }
        const uint32_t nk = (uint32_t) Kmers.size();
        out.append((const char *) &nk, 4);
        out.append((const char *) Kmers.data(), 4 * (size_t) nk);
        if (out.size() > ((size_t) 64 << 20)) {
            if (fwrite(out.data(), 1, out.size(), f) != out.size()) throw std::runtime_error("WriteRskdb: short write to " + FN);
            out.clear();
        }
    }
    if (fwrite(out.data(), 1, out.size(), f) != out.size()) throw std::runtime_error("WriteRskdb: short write to " + FN);
    closer.f = nullptr;
    if (fclose(f) != 0) throw std::runtime_error("WriteRskdb: writing " + FN + " failed");
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
    PhaseTimer tm("UploadToGpu");
    const uint n = GetDBChainCount();
    std::vector<uint32_t> len(n);
    std::vector<size_t> start((size_t) n + 1, 0);
    for (uint i = 0; i < n; ++i) { len[i] = m_DBChains[i]->GetSeqLength(); start[i + 1] = start[i] + len[i]; }
    const size_t tot = start[n];
    // the chains' vectors go straight into the set's staging buffer (r06; r01-r05 gathered them into flat arrays first: a second
    // pass over every byte, 60 MB of fresh pages per 11,211 chains and their release)
    rsk_chain_source from;
    from.mu = [&](uint32_t i, uint8_t *dst) { memcpy(dst, m_DBMuLettersVec[i]->data(), len[i]); };
    from.prof = [&](uint32_t i, int f, uint8_t *dst) { memcpy(dst, (*m_DBProfiles[i])[f].data(), len[i]); };
    from.xyz = [&](uint32_t i, int ax, float *dst) {
        const PDBChain &C = *m_DBChains[i];
        memcpy(dst, (ax == 0 ? C.m_Xs : ax == 1 ? C.m_Ys : C.m_Zs).data(), 4 * (size_t) len[i]);
    };
    for (uint i = 0; i < n; ++i)
        if (m_DBMuLettersVec[i]->size() != len[i] || m_DBProfiles[i]->size() != RSK_NFEAT || (*m_DBProfiles[i])[0].size() != len[i])
            throw std::runtime_error("UploadToGpu: chain " + std::to_string(i) + " has no Mu letters / profile of its length");
    check(rsk_db_create_from(m_Ctx, n, len.data(), from, m_DBSelfRevScores.data(), &m_Db), "rsk_db_create");
    tm.lap("rsk_db_create");
    // residue characters: the statistics kernel counts the identical columns of an alignment (GetPctId) while it walks the path
    {
        std::unique_ptr<char[]> seq(new char[tot + 1]);
        rsk_parallel_for(n, 512, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) memcpy(&seq[start[i]], m_DBChains[i]->m_Seq.data(), len[i]);
        });
        check(rsk_db_set_seq(m_Db, seq.get()), "rsk_db_set_seq");
    }
    tm.lap("residue characters");
}

}   // namespace reseek_amd

// TEST INFRASTRUCTURE: the reference's per-pair entry points driven the way its own threads drive them
// (runself.cpp:13-70: SetQuery once per i, SetTarget + AlignQueryTarget per j; postmufilter.cpp:116-155: ChainBag + AlignBags),
// against reseek_host.h + librsk.so.  Prints "<form>\t<hit line>" for every pair with E <= 10, both orientations as RunSelf.
//   usage: pair_main A.bca NA B.bca NB   (the first NA chains of A all-vs-all, the first NB chains of B all-vs-all)
#include <atomic>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <stdexcept>
#include <string>
#include <vector>

#include "reseek_host.h"

using namespace reseek_amd;

static void Run(const std::string &FN, uint N)
{
    DSSParams Params;
    Params.SetDSSParams(DM_AlwaysSensitive);
    DBSearcher DBS;                       // only as the loader: chains, profiles, Mu letters / k-mers, self-rev scores
    DBS.m_Params = &Params;
    DBS.LoadDB(FN);
    DBS.Setup();
    N = std::min(N, DBS.GetDBChainCount());
    DSSAligner DA;
    DA.SetParams(Params);
    DA.SetColumns("query+target+qlo+qhi+ql+tlo+thi+tl+pctid+pvalue+evalue+cigar+dpscore+lddt+newts+ids+gaps+aq");
    auto emit = [&](const char *Form, uint i, uint j) {
        if (DA.m_Path.empty()) return;
        for (int up = 1; up >= 0; --up) {
            if (!up && i == j) continue;
            if (DA.GetEvalue(up != 0) > 10) continue;
            std::string line;
            DA.AppendTsv(line, up != 0);
            printf("%s\t%s", Form, line.c_str());
        }
    };
    for (uint i = 0; i < N; ++i) {
        DA.SetQuery(*DBS.m_DBChains[i], DBS.m_DBProfiles[i], DBS.m_DBMuLettersVec[i], DBS.m_DBMuKmersVec[i], DBS.m_DBSelfRevScores[i]);
        for (uint j = i; j < N; ++j) {
            DA.SetTarget(*DBS.m_DBChains[j], DBS.m_DBProfiles[j], DBS.m_DBMuLettersVec[j], DBS.m_DBMuKmersVec[j], DBS.m_DBSelfRevScores[j]);
            DA.AlignQueryTarget();
            emit("AlignQueryTarget", i, j);
        }
        DA.UnsetQuery();
    }
    std::vector<ChainBag> Bags(N);
    for (uint i = 0; i < N; ++i) {
        Bags[i].m_ptrChain = DBS.m_DBChains[i];
        Bags[i].m_ptrProfile = DBS.m_DBProfiles[i];
        Bags[i].m_ptrMuLetters = DBS.m_DBMuLettersVec[i];
        Bags[i].m_ptrMuKmers = DBS.m_DBMuKmersVec[i];
        Bags[i].m_SelfRevScore = DBS.m_DBSelfRevScores[i];
    }
    for (uint i = 0; i < N; ++i)
        for (uint j = i; j < N; ++j) {
            DA.AlignBags(Bags[i], Bags[j]);
            emit("AlignBags", i, j);
        }
    // MuKmerFilter on its own: SetBagQ / AlignBag leave the HSP chain of the pair (mukmerfilter.h:27-39)
    MuKmerFilter MKF;
    MKF.SetParams(Params);
    uint nchains = 0;
    for (uint i = 0; i + 1 < N; ++i) {
        MKF.SetBagQ(Bags[i]);
        MKF.AlignBag(Bags[i + 1]);
        nchains += (uint) MKF.m_ChainHSPLois.size();
        MKF.ResetQ();
    }
    printf("# %s: %u chains, %u chained HSPs over neighbouring pairs\n", FN.c_str(), N, nchains);
}

// The reference's threading model on the per-pair entry points: T threads, ONE DSSAligner each (dbsearcher.cpp:98-106), rows
// dealt through a shared counter (runself.cpp:72-99) -- here all aligners share the default device context, whose batch-of-one
// calls the host layer serialises (CtxMutex).  Prints the same "AlignQueryTarget" rows as Run (in any order).
static void RunThreaded(const std::string &FN, uint N, uint T)
{
    DSSParams Params;
    Params.SetDSSParams(DM_AlwaysSensitive);
    DBSearcher DBS;
    DBS.m_Params = &Params;
    DBS.LoadDB(FN);
    DBS.Setup();
    N = std::min(N, DBS.GetDBChainCount());
    std::atomic<uint> next{0};
    std::mutex out_lock;
    std::vector<std::thread> ts;
    std::atomic<int> failed{0};
    for (uint t = 0; t < T; ++t)
        ts.emplace_back([&]() {
            try {
                DSSAligner DA;
                DA.SetParams(Params);
                DA.SetColumns("query+target+qlo+qhi+ql+tlo+thi+tl+pctid+pvalue+evalue+cigar+dpscore+lddt+newts+ids+gaps+aq");
                for (;;) {
                    const uint i = next.fetch_add(1);
                    if (i >= N) break;
                    DA.SetQuery(*DBS.m_DBChains[i], DBS.m_DBProfiles[i], DBS.m_DBMuLettersVec[i], DBS.m_DBMuKmersVec[i], DBS.m_DBSelfRevScores[i]);
                    for (uint j = i; j < N; ++j) {
                        DA.SetTarget(*DBS.m_DBChains[j], DBS.m_DBProfiles[j], DBS.m_DBMuLettersVec[j], DBS.m_DBMuKmersVec[j], DBS.m_DBSelfRevScores[j]);
                        DA.AlignQueryTarget();
                        if (DA.m_Path.empty()) continue;
                        for (int up = 1; up >= 0; --up) {
                            if (!up && i == j) continue;
                            if (DA.GetEvalue(up != 0) > 10) continue;
                            std::string line;
                            DA.AppendTsv(line, up != 0);
                            std::lock_guard<std::mutex> g(out_lock);
                            printf("AlignQueryTarget\t%s", line.c_str());
                        }
                    }
                    DA.UnsetQuery();
                }
            } catch (const std::exception &e) {
                fprintf(stderr, "pair_main thread: %s\n", e.what());
                failed = 1;
            }
        });
    for (auto &t : ts) t.join();
    if (failed) throw std::runtime_error("a worker thread failed");
}

int main(int argc, char **argv)
{
    try {
        if (argc >= 5 && std::string(argv[1]) == "-threads") {      // pair_main -threads T A.bca NA [B.bca NB ...]
            const uint T = (uint) atoi(argv[2]);
            for (int a = 3; a + 1 < argc; a += 2) RunThreaded(argv[a], (uint) atoi(argv[a + 1]), T);
            return 0;
        }
        for (int a = 1; a + 1 < argc; a += 2) Run(argv[a], (uint) atoi(argv[a + 1]));
    } catch (const std::exception &e) {
        fprintf(stderr, "pair_main: %s\n", e.what());
        return 1;
    }
    return 0;
}

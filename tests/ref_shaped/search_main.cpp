// TEST INFRASTRUCTURE: a stand-alone `-search` driver over reseek_host.h + librsk.so, used by tests/test_gpu_ref_shaped.py
// (compiled on the GPU box, its hit tables are diffed with the reference binary's goldens) and by tools/tsan_host.sh.
// It exercises the boundary classes of SURVEY 8b -- DBSearcher, ChainReader2, MuSeqSource, SeqDB, MuPreFilter,
// PostMuFilter, Open/CloseOutputFiles -- in the three shapes `reseek -search` has (all-vs-all, query set against a
// streamed -db file, two-stage -fast -db).  That the REFERENCE's own caller compiles against the same header is proved
// separately: oracle/Makefile.ref builds /root/reference/src/search.cpp against tests/ref_shaped/shim/ into
// oracle/_ref/search_refsrc (nothing of it is stored here).
//   usage: search_main QUERY [-db DB] -fast|-sensitive|-verysensitive -output HITS [-columns C] [-dbmu FA] [-keeptmp]
//                      [-noself] [-evalue E] [-devices 0,0,...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "cli_flags.h"
#include "reseek_host.h"

namespace ra = reseek_amd;

namespace {

// What the three shapes have in common: a loaded + set-up searcher over the positional chain file, hits file open
// while `body` runs.
struct LoadedSearch {
    ra::DSSParams params;
    ra::DBSearcher searcher;
    LoadedSearch()
    {
        params.SetDSSParams(ra::DM_UseCommandLineOption);
        searcher.m_Params = &params;
        searcher.LoadDB(ra::g_Arg1);
        searcher.Setup();
    }
    template <class F> void with_hits_file(F body)
    {
        ra::OpenOutputFiles();
        body(searcher);
        ra::CloseOutputFiles();
    }
};

void all_vs_all()
{
    LoadedSearch s;
    s.with_hits_file([](ra::DBSearcher &d) { d.RunSelf(); });
}

void against_streamed_db()
{
    LoadedSearch s;
    s.with_hits_file([](ra::DBSearcher &d) {
        ra::ChainReader2 targets;
        targets.Open(ra::g_Opts.db);
        d.RunQuery(targets);
    });
}

bool has_suffix(const std::string &s, const char *suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && !s.compare(s.size() - n, n, suf);
}

void two_stage_fast_db()
{
    const std::string db_file = ra::g_Opts.db, hits_file = ra::g_Opts.output;
    if (!has_suffix(db_file, ".bca")) throw std::runtime_error("-fast -db needs a .bca database");
    const std::string handoff = hits_file + ".prefilter.tmp";

    {   // stage 1: Mu letters of both sides, k-mer prefilter, candidate lists into the hand-off file
        ra::DSSParams prefilter_params;
        prefilter_params.SetDSSParams(ra::DM_UseCommandLineOption);
        if (prefilter_params.m_MuPrefPatternStr != "1110011") throw std::runtime_error("unexpected prefilter pattern");
        ra::MuSeqSource query_letters, db_letters;
        query_letters.OpenChains(ra::g_Arg1, prefilter_params);
        if (ra::g_Opts.dbmu.empty()) db_letters.OpenChains(db_file, prefilter_params);
        else db_letters.OpenFasta(ra::g_Opts.dbmu);
        ra::SeqDB query_index_input;
        query_index_input.FromSS(query_letters);
        ra::MuPreFilter(prefilter_params, query_index_input, db_letters, handoff);
    }
    {   // stage 2: the candidates aligned under the sensitive preset
        ra::DSSParams align_params;
        align_params.SetDSSParams(ra::DM_AlwaysSensitive);
        ra::PostMuFilter(align_params, handoff, ra::g_Arg1, db_file, hits_file);
    }
    if (!ra::g_Opts.keeptmp) remove(handoff.c_str());
}

}   // namespace

int main(int argc, char **argv)
{
    try {
        ref_shaped::parse_command_line(argc, argv);
        const bool db = !ra::g_Opts.db.empty();
        void (*const shape)() = !db ? all_vs_all : ra::g_Opts.fast_set() ? two_stage_fast_db : against_streamed_db;
        shape();
    } catch (const std::exception &e) {
        fprintf(stderr, "search_main: %s\n", e.what());
        return 1;
    }
    return 0;
}

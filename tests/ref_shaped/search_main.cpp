// TEST INFRASTRUCTURE: a caller written in the call sequence of the reference's `reseek -search` command
// (SelfSearch search.cpp:20-37, Search_NoMuFilter :39-60, cmd_search :62-111) against reseek_host.h + librsk.so.
// tests/test_gpu_ref_shaped.py compiles it on the GPU box and diffs its hit tables with the golden ones: the boundary test of
// SURVEY 8b ("signatures to keep").  Where the reference reads opt(x) / optset_x globals this file reads g_Opts.
//   usage: search_main QUERY [-db DB] -fast|-sensitive|-verysensitive -output HITS [-columns C] [-dbmu FA] [-keeptmp]
#include <cstring>
#include <stdexcept>
#include <string>

#include "reseek_host.h"

using namespace reseek_amd;
using std::string;

static void Die(const char *Msg) { throw std::runtime_error(Msg); }
static bool EndsWith(const string &s, const string &t) { return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }

static void SelfSearch()
{
    const string &QFN = g_Arg1;
    if (!g_Opts.db.empty()) Die("-db not used for -selfsearch");

    DBSearcher DBS;
    DSSParams Params;
    Params.SetDSSParams(DM_UseCommandLineOption);
    DBS.m_Params = &Params;

    DBS.LoadDB(QFN);
    DBS.Setup();

    OpenOutputFiles();
    DBS.RunSelf();
    CloseOutputFiles();
}

static void Search_NoMuFilter()
{
    if (g_Opts.db.empty()) Die("-db required");

    const string &QFN = g_Arg1;
    const string &DBFN = g_Opts.db;

    DBSearcher DBS;
    DSSParams Params;
    Params.SetDSSParams(DM_UseCommandLineOption);
    DBS.m_Params = &Params;

    DBS.LoadDB(QFN);
    DBS.Setup();

    OpenOutputFiles();
    ChainReader2 CR;
    CR.Open(DBFN);
    DBS.RunQuery(CR);
    CloseOutputFiles();
}

static void cmd_search()
{
    if (g_Opts.db.empty()) { SelfSearch(); return; }
    if (!g_Opts.fast_set()) { Search_NoMuFilter(); return; }

    const string &QueryFN = g_Arg1;
    const string DBFN = g_Opts.db;
    if (!EndsWith(DBFN, ".bca")) Die(".bca format required for -db");

    DSSParams Params;
    Params.SetDSSParams(DM_UseCommandLineOption);
    if (Params.m_MuPrefPatternStr != "1110011") Die("PatternStr");

    const string MuFilterTsvFN = g_Opts.output + ".prefilter.tmp";       // GetTmpFileName

    MuSeqSource QSS;
    MuSeqSource DBSS;
    QSS.OpenChains(QueryFN, Params);
    if (!g_Opts.dbmu.empty()) DBSS.OpenFasta(g_Opts.dbmu);
    else DBSS.OpenChains(DBFN, Params);

    SeqDB MuQueryDB;
    MuQueryDB.FromSS(QSS);

    MuPreFilter(Params, MuQueryDB, DBSS, MuFilterTsvFN);

    DSSParams Params2;
    Params2.SetDSSParams(DM_AlwaysSensitive);
    PostMuFilter(Params2, MuFilterTsvFN, QueryFN, DBFN, g_Opts.output);

    if (!g_Opts.keeptmp) remove(MuFilterTsvFN.c_str());
}

int main(int argc, char **argv)
{
    try {
        if (argc < 2) Die("usage: search_main QUERY [-db DB] -fast|-sensitive|-verysensitive -output HITS");
        g_Arg1 = argv[1];
        for (int i = 2; i < argc; ++i) {
            const string a = argv[i];
            auto val = [&]() -> string { if (i + 1 >= argc) Die("missing option value"); return argv[++i]; };
            if (a == "-db") g_Opts.db = val();
            else if (a == "-output") g_Opts.output = val();
            else if (a == "-columns") g_Opts.columns = val();
            else if (a == "-dbmu") g_Opts.dbmu = val();
            else if (a == "-fast") g_Opts.mode = AM_Fast;
            else if (a == "-sensitive") g_Opts.mode = AM_Sensitive;
            else if (a == "-verysensitive") g_Opts.mode = AM_VerySensitive;
            else if (a == "-keeptmp") g_Opts.keeptmp = true;
            else if (a == "-noself") g_Opts.noself = true;
            else if (a == "-evalue") { g_Opts.evalue_set = true; g_Opts.evalue = atof(val().c_str()); }
            else Die(("unknown option " + a).c_str());
        }
        cmd_search();
    } catch (const std::exception &e) {
        fprintf(stderr, "search_main: %s\n", e.what());
        return 1;
    }
    return 0;
}

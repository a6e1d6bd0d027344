// TEST INFRASTRUCTURE (build container only): include-path shim that lets the REFERENCE's own caller of the -search path,
// /root/reference/src/search.cpp, compile unmodified against reseek_host.h.  oracle/Makefile.ref feeds that file to the
// compiler through stdin (so its quoted includes resolve here, not next to it), wrapped in `namespace reseek_amd { }` with
// this header pre-included: the file's own declarations of MuPreFilter / PostMuFilter (search.cpp:9-18) then RE-declare the
// host layer's functions -- a signature that differs would not link -- and its class names resolve to the host layer's.
// The object is linked with shim_main.cpp + librsk.so into oracle/_ref/search_refsrc.  Nothing of the reference is stored in the repo: this directory only maps the
// names that file expects from its own headers (myutils.h, dss.h, seqdb.h, museqsource.h, dbsearcher.h, output.h,
// statsig.h) onto the host layer's.  If a signature of the boundary drifts from the reference's, this build breaks.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "reseek_host.h"

using std::string;
// the option globals of myutils.h:365-372 as views of g_Opts
#define opt(x) (reseek_amd::g_Opts.x)
#define optset_db (!reseek_amd::g_Opts.db.empty())
#define optset_dbmu (!reseek_amd::g_Opts.dbmu.empty())
#define optset_fast (reseek_amd::g_Opts.fast_set())

#define asserta(e) do { if (!(e)) throw std::runtime_error("assert failed: " #e); } while (0)

inline void Die(const char *Format, ...)
{
    char msg[1024];
    va_list ap;
    va_start(ap, Format);
    vsnprintf(msg, sizeof msg, Format, ap);
    va_end(ap);
    throw std::runtime_error(msg);
}
inline void Log(const char *, ...) {}
inline bool EndsWith(const string &s, const string &t) { return s.size() >= t.size() && !s.compare(s.size() - t.size(), t.size(), t); }
inline void GetTmpFileName(string &FN) { FN = reseek_amd::g_Opts.output + ".prefilter.tmp"; }
inline void DeleteStdioFile(const string &FN) { remove(FN.c_str()); }

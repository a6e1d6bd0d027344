// TEST INFRASTRUCTURE: the command-line switches the test drivers under tests/ref_shaped understand, as one table
// (switch, whether a value follows, what it sets in reseek_amd::g_Opts).  Shared by search_main.cpp and
// shim/shim_main.cpp.
#pragma once
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "reseek_host.h"

namespace ref_shaped {

struct Flag {
    const char *name;
    bool takes_value;
    std::function<void(reseek_amd::SearchOptions &, const char *)> apply;
};

inline const std::vector<Flag> &flag_table()
{
    using O = reseek_amd::SearchOptions;
    static const std::vector<Flag> t = {
        {"-db", true, [](O &o, const char *v) { o.db = v; }},
        {"-output", true, [](O &o, const char *v) { o.output = v; }},
        {"-columns", true, [](O &o, const char *v) { o.columns = v; }},
        {"-dbmu", true, [](O &o, const char *v) { o.dbmu = v; }},
        {"-evalue", true, [](O &o, const char *v) { o.evalue_set = true; o.evalue = atof(v); }},
        {"-devices", true, [](O &, const char *v) { setenv("RSK_DEVICES", v, 1); }},
        {"-fast", false, [](O &o, const char *) { o.mode = reseek_amd::AM_Fast; }},
        {"-sensitive", false, [](O &o, const char *) { o.mode = reseek_amd::AM_Sensitive; }},
        {"-verysensitive", false, [](O &o, const char *) { o.mode = reseek_amd::AM_VerySensitive; }},
        {"-keeptmp", false, [](O &o, const char *) { o.keeptmp = true; }},
        {"-noself", false, [](O &o, const char *) { o.noself = true; }},
    };
    return t;
}

// argv[1] = the positional chain file (g_Arg1), the rest switches of the table
inline void parse_command_line(int argc, char **argv)
{
    if (argc < 2) throw std::runtime_error("usage: PROG QUERY [-db DB] -fast|-sensitive|-verysensitive -output HITS [...]");
    reseek_amd::g_Arg1 = argv[1];
    for (int k = 2; k < argc; ++k) {
        const Flag *hit = nullptr;
        for (const Flag &f : flag_table())
            if (!strcmp(f.name, argv[k])) hit = &f;
        if (!hit) throw std::runtime_error(std::string("unknown option ") + argv[k]);
        if (hit->takes_value && k + 1 >= argc) throw std::runtime_error(std::string("no value after ") + argv[k]);
        hit->apply(reseek_amd::g_Opts, hit->takes_value ? argv[++k] : nullptr);
    }
}

}   // namespace ref_shaped

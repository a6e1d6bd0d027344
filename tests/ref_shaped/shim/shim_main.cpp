// TEST INFRASTRUCTURE (build container only): main() for oracle/_ref/search_refsrc = the reference's own search.cpp
// compiled against shim/ + this file + librsk.so.  See shim/myutils.h for how that file is compiled.
#include <cstdio>

#include "../cli_flags.h"
#include "myutils.h"

namespace reseek_amd { void cmd_search(); }      // defined by the reference's search.cpp (compiled inside the namespace)

int main(int argc, char **argv)
{
    try {
        ref_shaped::parse_command_line(argc, argv);
        reseek_amd::cmd_search();
    } catch (const std::exception &e) {
        fprintf(stderr, "search_refsrc: %s\n", e.what());
        return 1;
    }
    return 0;
}

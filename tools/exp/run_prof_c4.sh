#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/prof_search.sh r05_search_c4 qdb 1000 87500 verysensitive > gpurun_out/prof_c4.log 2>&1
tail -40 gpurun_out/prof_c4.log | cut -c1-160

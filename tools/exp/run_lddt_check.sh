#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_search.py tests/test_gpu_configs.py tests/test_gpu_xdrop.py -x -q 2>&1 | tail -4
python tools/bench_search.py qdb 1000 30000 verysensitive 2>/dev/null | grep seconds
bash tools/prof_search.sh r04b_search_c4s qdb 1000 30000 verysensitive 2>&1 | head -9

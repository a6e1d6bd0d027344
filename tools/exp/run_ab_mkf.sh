#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in 1 0 1 0; do echo "RSK_MKF_EARLY=$e"; RSK_MKF_EARLY=$e python tools/bench_search.py 0 sensitive 2>/dev/null | grep seconds; done
for e in 1 0; do echo "bca RSK_MKF_EARLY=$e"; RSK_MKF_EARLY=$e python tools/bench_search.py 0 sensitive bca 2>/dev/null | grep seconds; done
for e in 1 0; do echo "c3 RSK_MKF_EARLY=$e"; RSK_MKF_EARLY=$e python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | grep seconds; done

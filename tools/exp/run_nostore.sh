#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== shipped"; python tools/bench_align_groups.py 2>/dev/null | grep "sw_kernel_ms\|Tcells"
echo "== no trace stores (results wrong, timing only)"; RSK_LIB=$PWD/reseek_amd/librsk_nostore.so python tools/bench_align_groups.py 2>/dev/null | grep "sw_kernel_ms\|Tcells"

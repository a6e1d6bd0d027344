#!/bin/bash
# k_sw_float table-copy variants: bench.py --live-only per library
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== tree (16 copies, 8 waves, v_mad addresses)"; python tools/exp/live_ms.py 2>&1 | grep k_sw_float
for v in "$@"; do echo "== $v"; RSK_LIB=$PWD/build/var_$v/librsk.so python tools/exp/live_ms.py 2>&1 | grep k_sw_float; done

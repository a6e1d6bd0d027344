#!/bin/bash
# k_sw_qp lane-group geometry experiment: unaligned rows (G = 25) vs 256-byte-aligned rows (G = 16) with 16-lane groups
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=$PWD/build/var_g16/librsk.so
for w in "181 192" "169 180" "100 108"; do
  python tools/exp/swq_conflict.py $w
  RSK_SWQ_GS=16 python tools/exp/swq_conflict.py $w
  RSK_LIB=$V RSK_SWQ_GS=16 python tools/exp/swq_conflict.py $w
done
echo "== live bench, base"; python tools/exp/live_ms.py
echo "== live bench, g16 GS=16"; RSK_LIB=$V RSK_SWQ_GS=16 python tools/exp/live_ms.py

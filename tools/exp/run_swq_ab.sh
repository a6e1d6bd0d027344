#!/bin/bash
# same-box A/B of k_sw_qp libraries: run_swq_ab.sh <variant...> ("main" = reseek_amd/librsk.so), two rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/swq_ab.txt
for round in 1 2; do
for v in "$@"; do
  if [ $v = main ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_$v/librsk.so; fi
  echo "== $v" >> gpurun_out/swq_ab.txt
  timeout 600 python tools/exp/swq_bench.py 5 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.4f T cells/s  trace %.2f GB' % (d['median_ms'], d['Tcells_per_s'], d['trace_bytes']/1e9))" >> gpurun_out/swq_ab.txt
done
done
cat gpurun_out/swq_ab.txt

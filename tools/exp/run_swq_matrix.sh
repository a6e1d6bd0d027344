#!/bin/bash
# k_sw_qp variants x item sizes on two query samples (swq_bench: bench.py's; groups: bench_align_groups.py's, longer queries)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/swq_matrix.txt
run() {   # label, lib ("main" or variant), env assignments...
  label=$1; lib=$2; shift 2
  ( for v in "$@"; do export $v; done
    if [ $lib != main ]; then export RSK_LIB=$PWD/build/var_$lib/librsk.so; fi
    a=$(timeout 300 python tools/exp/swq_bench.py 3 2>&1 | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['median_ms'])")
    b=$(timeout 300 python tools/bench_align_groups.py 64 11211 2>&1 | grep sw_kernel_ms | tail -1 | grep -o '[0-9.]*' | head -1 | cut -c1-6)
    echo "$label: swq $a ms  groups $b ms" >> gpurun_out/swq_matrix.txt )
}
run "base (r04)" base
run "gs8 by-work" main
run "gs8 chunk1" main RSK_SWQ_CHUNK=1
run "gs8 chunk2" main RSK_SWQ_CHUNK=2
run "gs16 by-work" gs16
run "gs16 chunk2" gs16 RSK_SWQ_CHUNK=2
run "gs16 chunk4" gs16 RSK_SWQ_CHUNK=4
run "gs16 chunk2 nonpersistent" gs16 RSK_SWQ_CHUNK=2 RSK_SWQ_PERSIST=0
run "base (r04) again" base
cat gpurun_out/swq_matrix.txt

#!/bin/bash
# k_sw_qp / k_sw_float crossover on the -sensitive survivors of the SCOP40-shaped set: groups of at least
# RSK_SWQ_MIN_LANES / 16 pairs take the query-profile kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/crossover.txt
for ml in 4096 2048 1024 512 256 128; do
  RSK_SWQ_MIN_LANES=$ml timeout 300 python tools/bench_align.py 3 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('min_lanes %5s (groups >= %3d pairs)  sw kernels %.2f ms  %.3f T cells/s  call %.1f ms  trace %.2f GB' % (d['env'].get('RSK_SWQ_MIN_LANES'), int(d['env'].get('RSK_SWQ_MIN_LANES'))//16, d['sw_kernel_ms'], d['Tcells_per_s'], d['call_ms_incl_python'], d['trace_bytes']/1e9))" >> gpurun_out/crossover.txt
done
cat gpurun_out/crossover.txt

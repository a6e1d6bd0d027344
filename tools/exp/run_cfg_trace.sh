#!/bin/bash
# RSK_TRACE stage timings of the BASELINE configs[2..4] legs (one GPU): stderr of `bench.py --configs-only X`
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for c in ${1:-config2 config3 config4}; do
  RSK_TRACE=1 timeout 900 python bench.py --configs-only $c > gpurun_out/trace_$c.json 2> gpurun_out/trace_$c.err
  echo "== $c"; tail -c 1500 gpurun_out/trace_$c.json
done

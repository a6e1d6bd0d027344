#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
RSK_TRACE=1 timeout 600 python tools/bench_search.py 0 sensitive > gpurun_out/self_trace.json 2> gpurun_out/self_trace.err
grep -v "amdgpu.ids" gpurun_out/self_trace.err | tail -60 | cut -c1-170
tail -c 400 gpurun_out/self_trace.json

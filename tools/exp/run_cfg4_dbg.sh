#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ulimit -c 0
for v in RSK_NONE=1; do
( export $v; timeout 200 python bench.py --configs-only config3 > gpurun_out/cfg3_dbg.json 2> gpurun_out/cfg3_dbg.err; echo "$v rc=$?"; grep -i "fault\|error" gpurun_out/cfg3_dbg.err | head -3; tail -c 200 gpurun_out/cfg3_dbg.json )
done

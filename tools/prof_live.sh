#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + PMC passes of `bench.py --live-only` (k_mu_sw, k_sw_float, k_sw_qp on the
# SCOP40-shaped set) -> gpurun_out/prof_<tag>/{summary.txt, live_pmc.json}; copy them to profiles/<tag>_live_* and
# profiles/r05_live_pmc.json (read by bench.py, which checks the kernel sources' sha256 recorded in it).  Counters are collected in their own passes (no trace options with --pmc).
TAG=${1:-r05_live}
# rocprofv3 databases stay in /tmp on the box (gpurun merges back at most 64 MiB); the summaries are copied to gpurun_out/prof_<tag>/
OUT=/tmp/rsk_prof/prof_$TAG
KEEP=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $KEEP
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --live-only"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc -- $CMD > $OUT/pmc4.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_live_json.py $OUT > $OUT/live_pmc.json 2> $OUT/live_pmc.err
cat $OUT/live_pmc.json
cp $OUT/*.txt $OUT/*.json $OUT/*.log $OUT/*.err $KEEP/ 2>/dev/null

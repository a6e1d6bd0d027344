#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + PMC passes of the bench command; summaries -> gpurun_out/prof_<tag>/
TAG=${1:-r01}
# rocprofv3 databases stay in /tmp on the box (gpurun merges back at most 64 MiB); the summaries are copied to gpurun_out/prof_<tag>/
OUT=/tmp/rsk_prof/prof_$TAG
KEEP=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $KEEP
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-search --no-live --no-predict"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc -- $CMD > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_traffic_json.py $OUT > $OUT/traffic.json 2> $OUT/traffic.err
cat $OUT/summary.txt $OUT/traffic.json
cp $OUT/*.txt $OUT/*.json $OUT/*.log $OUT/*.err $KEEP/ 2>/dev/null

#!/bin/bash
# Runs on the GPU box: two PMC passes of one command for several library variants / environments.
#   prof_variants.sh TAG "CMD" name1:"ENV..." name2:"ENV..."   -> gpurun_out/prof_TAG/<name>/live_pmc.json (prof_live_json.py format)
TAG=$1; CMD=$2; shift 2
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  OUT=$ROOT/gpurun_out/prof_$TAG/$name
  mkdir -p $OUT
  ( cd /tmp
    env $envs rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
    env $envs rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1 )
  python3 $ROOT/tools/prof_live_json.py $OUT > $OUT/live_pmc.json 2> $OUT/live_pmc.err
  echo "== $name"; python3 - <<PY
import json
d=json.load(open("$OUT/live_pmc.json"))
for k,v in d.items():
    c=v["counters"]
    print(k, "ms", [round(x,2) for x in v.get("dispatch_ms_profiled",[])], "valu_issue %.3f lds_busy %.3f lds_conflict %.3f" % (v.get("valu_issue_frac",0), v.get("lds_busy_frac",0), v.get("lds_conflict_frac",0)),
          "INSTS_VALU %.3e INSTS_LDS %.3e SALU %.3e SMEM %.3e WAIT_INST_LDS %.3e WAIT_ANY %.3e WAVE_CYCLES %.3e" % (c.get("SQ_INSTS_VALU",0), c.get("SQ_INSTS_LDS",0), c.get("SQ_INSTS_SALU",0), c.get("SQ_INSTS_SMEM",0), c.get("SQ_WAIT_INST_LDS",0), c.get("SQ_WAIT_ANY",0), c.get("SQ_WAVE_CYCLES",0)))
PY
done

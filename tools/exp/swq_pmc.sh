#!/bin/bash
# PMC passes of k_sw_qp alone (tools/exp/swq_conflict.py over all chain lengths) -> gpurun_out/swq_pmc.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
OUT=/tmp/rsk_prof/swq_pmc
rm -rf $OUT; mkdir -p $OUT gpurun_out
export TMPDIR=/tmp
CMD="python $R/tools/exp/swq_conflict.py 0 100000 64"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
python3 $R/tools/prof_live_json.py $OUT > $R/gpurun_out/swq_pmc.json 2> $R/gpurun_out/swq_pmc.err
grep "^L" $OUT/pmc1.log
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/swq_pmc.json"))["k_sw_qp"]
print({k:v for k,v in d.items() if k.endswith("frac") or k=="cycles"})
c=d["counters"]; print({k:c[k] for k in sorted(c)})
PY

#!/bin/bash
# PMC passes of k_sw_qp alone (tools/exp/swq_bench.py) for the library variants given ("main" = reseek_amd/librsk.so) -> gpurun_out/swq_pmc_<variant>.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "$@"; do
  if [ $v = main ]; then unset RSK_LIB; else export RSK_LIB=$R/build/var_$v/librsk.so; fi
  OUT=/tmp/rsk_prof/swq_pmc_$v
  rm -rf $OUT; mkdir -p $OUT
  CMD="python $R/tools/exp/swq_bench.py 2"
  cd /tmp
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM -d $OUT/pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU2 -d $OUT/pmc3 -o pmc -- $CMD > $OUT/pmc3.log 2>&1
  python3 $R/tools/prof_live_json.py $OUT > $R/gpurun_out/swq_pmc_$v.json 2> $R/gpurun_out/swq_pmc_$v.err
  cd $R
  python3 - <<PY
import json
d=json.load(open("gpurun_out/swq_pmc_$v.json"))["k_sw_qp"]
c=d["counters"]; print("$v", {k:c[k] for k in sorted(c)})
PY
done

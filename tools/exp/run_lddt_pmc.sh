#!/bin/bash
# PMC counters of k_lddt / k_traceback on bench.py's structure-based live entries
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
OUT=/tmp/rsk_prof/lddt_pmc; rm -rf $OUT; mkdir -p $OUT gpurun_out
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc1 -o pmc -- python $R/bench.py --live-only > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- python $R/bench.py --live-only > $OUT/pmc2.log 2>&1
python3 $R/tools/prof_summary.py $OUT 2>&1 | grep "k_lddt(\|k_traceback" | cut -c1-60,100-200 > $R/gpurun_out/lddt_pmc.txt
cat $R/gpurun_out/lddt_pmc.txt

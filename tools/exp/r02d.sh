O=$GRAFT_REPO_ROOT/gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_align.py 1"
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc1 -o pmc -- $CMD > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $O/pmc2 -o pmc -- $CMD > $O/pmc2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $O > $O/summary.txt 2>&1
grep -v "k_mu_sw\|k_mf_\|k_len\|hipcub\|rocprim" $O/summary.txt | head -80
tail -3 $O/pmc2.log

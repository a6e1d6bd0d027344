#!/bin/bash
# prefilter seed walk: four posting loads in flight for short rows (as built) vs one at a time (-DPF_SHORTROW_SERIAL), both letter sets
cd ${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu 2>&1 | grep "passed\|failed"
for round in 1 2; do
for v in main pfserial; do
  if [ $v = main ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_$v/librsk.so; fi
  for set in scop40 syn; do
    echo "$v $set: $(timeout 600 python tools/exp/pf_bench.py $set 0 2 2>&1 | grep 'rep 1\|digest' | tr '\n' ' ' | cut -c1-230)"
  done
done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06_pf_ab.txt
for rep in 1 2; do for v in head pf1024 pf1024b; do
  if [ $v = head ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_$v/librsk.so; fi
  for set in scop40 syn; do echo "$v $set $(RSK_TRACE=1 timeout 600 python tools/exp/pf_bench.py $set 0 2 2>&1 | grep 'rep 1\|digest\|workgroup cycles' | tr '\n' ' ' | cut -c1-420)" >> gpurun_out/r06_pf_ab.txt; done
done; done
cat gpurun_out/r06_pf_ab.txt

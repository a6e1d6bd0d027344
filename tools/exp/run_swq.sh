#!/bin/bash
# k_sw_qp iteration loop: alignment parity tests, then the 64 x 11,211 benchmark (and variants under build/var_*)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/swq_tests.txt
cat gpurun_out/swq_tests.txt
: > gpurun_out/swq_bench.txt
for v in main "$@"; do
  if [ $v = main ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_$v/librsk.so; fi
  echo "== $v" >> gpurun_out/swq_bench.txt
  timeout 600 python tools/exp/swq_bench.py 5 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/swq_bench.txt
done
cat gpurun_out/swq_bench.txt

#!/bin/bash
# round-6 first GPU call: the N > 1 plumbing tests, the gapless / alignment tests touched by the ADVICE fixes, one full bench run
# (the compact line), and a probe of what the box exposes for clocks / power
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RSK_REQUIRE_REF=1
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_rccl.py tests/test_gpu_gapless.py tests/test_gpu_align.py -x -q > gpurun_out/r06a_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06a_tests.txt
tail -5 gpurun_out/r06a_tests.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06a_bench_detail.json ) > gpurun_out/r06a_bench.out 2> gpurun_out/r06a_bench.err
tail -n 1 gpurun_out/r06a_bench.out | wc -c
tail -n 1 gpurun_out/r06a_bench.out
tail -5 gpurun_out/r06a_bench.err
{
  echo "== sysfs"; for d in /sys/class/drm/card*/device; do echo $d; ls $d | tr '\n' ' '; echo; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo
    for f in $d/hwmon/*/freq1_input $d/hwmon/*/power1_cap $d/hwmon/*/power1_average $d/hwmon/*/power1_input $d/hwmon/*/power1_cap_max $d/gpu_busy_percent $d/pp_dpm_sclk; do echo "$f: $(cat $f 2>&1 | tr '\n' ' ')"; done; done
  echo "== rocm-smi"; time rocm-smi --showclocks --showpower --showmaxpower 2>&1 | head -40
  echo "== amd-smi"; time amd-smi metric -p -c 2>&1 | head -60
  python -c "import amdsmi; print('amdsmi python ok')" 2>&1 | tail -1
  nproc; cat /sys/fs/cgroup/cpu.max
} > gpurun_out/r06a_probe.txt 2>&1

#!/bin/bash
# round-6 second GPU call: (1) tests of the touched paths, (2) per-shard times of the live self search (predicted_scaling.search),
# (3) configs[2..4] with clocks / counters on the current library, (4) the configs[4] share on the libraries of 59f77c8 (r04 final),
# 70433b1, dc18e0f, b323138, cc115d6 (r05 final) and HEAD, alternating, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RSK_REQUIRE_REF=1
timeout 1200 python -m pytest tests/test_gpu_align.py tests/test_gpu_search.py tests/test_capi_exports.py -x -q > gpurun_out/r06b_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06b_tests.txt; tail -3 gpurun_out/r06b_tests.txt
timeout 900 python bench.py --search-scaling-only > gpurun_out/r06b_search_scaling.json 2> gpurun_out/r06b_search_scaling.err
tail -c 1500 gpurun_out/r06b_search_scaling.json; tail -3 gpurun_out/r06b_search_scaling.err
timeout 900 python bench.py --configs-only config2,config3,config4 > gpurun_out/r06b_configs.json 2> gpurun_out/r06b_configs.err
tail -3 gpurun_out/r06b_configs.err
: > gpurun_out/r06b_ab_c4.txt
for rep in 1 2; do
  for v in head r04final r05base lddt times r05final; do
    if [ $v = head ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_$v/librsk.so; fi
    timeout 600 python bench.py --configs-only config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']
for k,x in d.items():
    if isinstance(x,dict): print('$v', k.split('_')[0], '%.3f s' % x['seconds'], 'sclk', (x.get('clock') or {}).get('sclk_busy_mean_ghz'), 'W', (x.get('clock') or {}).get('power_busy_mean_w'), 'swqp GHz', x.get('swqp_clock_ghz'))
" >> gpurun_out/r06b_ab_c4.txt
  done
done
unset RSK_LIB
cat gpurun_out/r06b_ab_c4.txt

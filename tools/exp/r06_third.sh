#!/bin/bash
# round-6 third GPU call: window shards of the live self search, pinned-staging uploads, incremental k_traceback
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RSK_REQUIRE_REF=1
timeout 2400 python -m pytest tests/test_gpu_mu_sw.py tests/test_gpu_search.py tests/test_gpu_align.py tests/test_gpu_multidev.py tests/test_gpu_dist.py tests/test_gpu_db_goldens.py -x -q > gpurun_out/r06c_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06c_tests.txt; tail -15 gpurun_out/r06c_tests.txt
timeout 900 python bench.py --search-scaling-only > gpurun_out/r06c_search_scaling.json 2> gpurun_out/r06c_search_scaling.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r06c_search_scaling.json'))['predicted_scaling']['search']
for tag in ('rskdb','bca'):
    e=d[tag]; print(tag, e['one_gpu_seconds'], e['hits'])
    for k in ('n2','n4','n8'):
        if k in e: print('  ',k, e[k]['shard_seconds'], 'max/mean', e[k]['max_over_mean'], 'eff', e[k]['efficiency'], 'pairs max/mean', e[k]['pairs_max_over_mean'])
P
tail -3 gpurun_out/r06c_search_scaling.err
timeout 900 python bench.py --live-only > gpurun_out/r06c_live.json 2> gpurun_out/r06c_live.err
python - <<'P'
import json
for e in json.load(open('gpurun_out/r06c_live.json'))['roofline_live']:
    print(e['kernel'], round(e['kernel_ms'],3), 'frac', round(e.get('frac',0),3), e.get('ns_per_step_of_the_longest_walk'))
P
timeout 900 python bench.py --configs-only config2,config3,config4 > gpurun_out/r06c_configs.json 2> gpurun_out/r06c_configs.err
python - <<'P'
import json
c=json.load(open('gpurun_out/r06c_configs.json'))['configs']
for k,x in c.items():
    if isinstance(x,dict):
        print(k, {kk:x.get(kk) for kk in ('seconds','sw_pairs','hits','swqp_clock_ghz','sw_pairs_scored_frac','upload_copies','upload_MB','db_batches','loader_seconds','featurise_seconds')}, (x.get('clock') or {}).get('sclk_busy_mean_ghz'), x.get('vs_reference_on_sample',{}).get('identical'))
P

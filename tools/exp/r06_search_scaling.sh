#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_multidev.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
for rep in 1 2; do
timeout 600 python bench.py --search-scaling-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['predicted_scaling']['search']
for tag in ('rskdb','bca'):
    e=d[tag]; print(tag, e['one_gpu_seconds'], e['hits'])
    for k in ('n2','n4','n8'):
        if k in e: print('  ',k, e[k]['shard_seconds'], 'max/mean', e[k]['max_over_mean'], 'eff', e[k]['efficiency'])
"
done | tee gpurun_out/r06h_search_scaling.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
free -g | head -2; df -h /tmp | tail -1; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 1200 python tools/bench_configs_full.py --scale ${1:-0.01} > gpurun_out/cfgfull_small.json 2> gpurun_out/cfgfull_small.err
tail -5 gpurun_out/cfgfull_small.err
python - <<'P'
import json
d=json.load(open('gpurun_out/cfgfull_small.json'))
for k in ('config3','config4'):
    e=d[k]; print(k, e['union_8_equals_3_equals_1'], e['pairs_and_hits_add_up'], e.get('vs_reference_on_sample'))
    for S in (8,3,1):
        r=e['shards_%d'%S]; print(' S',S,'total %.2f max %.2f imb %.2f rss %.1f GB dev %.1f GB hits %d' % (r['seconds_total'], r['seconds_max_shard'], r['imbalance_max_over_mean'], r['peak_host_rss_gb'], r['peak_device_bytes_in_use_gb'], r['hits']))
P

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 3000 python tools/bench_configs_full.py > gpurun_out/r04_configs_full.json 2> gpurun_out/r04_configs_full.err
tail -5 gpurun_out/r04_configs_full.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r04_configs_full.json'))
for k in ('config3','config4'):
    e=d[k]; print(k, e['union_8_equals_3_equals_1'], e['pairs_and_hits_add_up'], e.get('vs_reference_on_sample'), 'gen %.0f s' % e['generation_seconds'])
    for S in (8,3,1):
        r=e['shards_%d'%S]; print(' S',S,'total %.2f max %.2f imb %.2f rss %.1f GB dev %.1f GB hits %d  %.1f M pairs/s' % (r['seconds_total'], r['seconds_max_shard'], r['imbalance_max_over_mean'], r['peak_host_rss_gb'], r['peak_device_bytes_in_use_gb'], r['hits'], r['chain_pairs_per_sec_one_gpu']/1e6), [round(s['seconds'],2) for s in r['shards']])
P

#!/bin/bash
# round-6 fifth GPU call: shard phases with the long-chain job under the filter + upload laps; search scaling again
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python - > gpurun_out/r06e_shard_trace.txt 2>&1 <<'P'
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tools')
import bench, bench_search, reseek_amd
seqs = bench.synth_mu_chains(0x5EED5EEC, None)
ctx = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
with tempfile.TemporaryDirectory() as td:
    db, out = os.path.join(td, "syn.rskdb"), os.path.join(td, "hits.tsv")
    bench_search.write_rskdb(db, seqs, np.random.default_rng(5))
    for k in (0, 3, 7):
        ctx.search(db, out, "sensitive", shard_index=k, shard_count=8)
    os.environ["RSK_TRACE"] = "1"
    for k in (3, 7):
        sys.stderr.write("==== shard %d of 8\n" % k); sys.stderr.flush()
        t0 = time.perf_counter(); ctx.search(db, out, "sensitive", shard_index=k, shard_count=8); sys.stderr.write("==== %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))
P
grep -v "amdgpu.ids\|rsk_align_pairs\]" gpurun_out/r06e_shard_trace.txt | head -80
timeout 900 python bench.py --search-scaling-only > gpurun_out/r06e_search_scaling.json 2> gpurun_out/r06e_search_scaling.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r06e_search_scaling.json'))['predicted_scaling']['search']
for tag in ('rskdb','bca'):
    e=d[tag]; print(tag, e['one_gpu_seconds'], e['hits'])
    for k in ('n2','n4','n8'):
        if k in e: print('  ',k, e[k]['shard_seconds'], 'max/mean', e[k]['max_over_mean'], 'eff', e[k]['efficiency'])
P

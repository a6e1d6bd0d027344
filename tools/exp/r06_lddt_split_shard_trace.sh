#!/bin/bash
# round-6 fourth GPU call: k_lddt split A/B, phases of one shard call (RSK_TRACE), copy counts of configs[3] under rocprofv3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RSK_REQUIRE_REF=1
timeout 1200 python -m pytest tests/test_gpu_align.py tests/test_gpu_vs_reference_binary.py tests/test_gpu_search.py -x -q > gpurun_out/r06d_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06d_tests.txt; tail -5 gpurun_out/r06d_tests.txt
for v in 1 0 1 0; do
  RSK_LDDT_SPLIT=$v timeout 600 python bench.py --live-only 2>/dev/null | python -c "
import json,sys
for e in json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline_live']:
    if e['kernel'] in ('k_lddt','k_traceback'): print('split=$v', e['kernel'], round(e['kernel_ms'],4), 'frac', round(e.get('frac',0),3))
"
done > gpurun_out/r06d_lddt_ab.txt
cat gpurun_out/r06d_lddt_ab.txt
for v in 1 0 1 0; do
  RSK_LDDT_SPLIT=$v timeout 600 python bench.py --configs-only config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['configs']
for k,x in d.items():
    if isinstance(x,dict): print('split=$v', k.split('_')[0], '%.3f s' % x['seconds'], 'swqp GHz', x.get('swqp_clock_ghz'))
"
done > gpurun_out/r06d_c4_ab.txt
cat gpurun_out/r06d_c4_ab.txt
python - > gpurun_out/r06d_shard_trace.txt 2>&1 <<'P'
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
    sys.stderr.write("==== whole\n")
    t0 = time.perf_counter(); ctx.search(db, out, "sensitive"); sys.stderr.write("==== %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))
P
grep -v amdgpu.ids gpurun_out/r06d_shard_trace.txt | head -120
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/r06d_c3prof -o c3 -- python $GRAFT_REPO_ROOT/bench.py --configs-only config3 > $GRAFT_REPO_ROOT/gpurun_out/r06d_c3prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - > gpurun_out/r06d_c3prof_summary.txt 2>&1 <<'PY'
import glob, os, sqlite3
for f in sorted(glob.glob("/tmp/r06d_c3prof/**/*.db", recursive=True)):
    c = sqlite3.connect(f)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    print(f, len(names), "tables/views")
    for n in names:
        if "copy" in n.lower() or n.startswith("top"):
            try:
                k = c.execute("select count(*) from \"%s\"" % n).fetchone()[0]
                print("  ", n, k, "rows")
                if k and k < 40:
                    for r in c.execute("select * from \"%s\"" % n).fetchall(): print("      ", r)
            except Exception as e:
                print("  ", n, "error", e)
for f in glob.glob("/tmp/r06d_c3prof/**/*.csv", recursive=True):
    print(f); print(open(f).read()[:3000])
PY
cat gpurun_out/r06d_c3prof_summary.txt | head -80

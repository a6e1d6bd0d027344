#!/bin/bash
# per-kernel times of one rsk_align_pairs call (64 queries x 11,211 chains, traceback + LDDT included), no overlap: base vs main
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/groups_trace.txt
for v in main base; do
  if [ $v = main ]; then unset RSK_LIB; else export RSK_LIB=$R/build/var_$v/librsk.so; fi
  OUT=/tmp/rsk_prof/groups_$v; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $R/tools/bench_align_groups.py 64 11211 > $OUT/log.txt 2>&1)
  echo "== $v" >> gpurun_out/groups_trace.txt
  python3 - $OUT >> gpurun_out/groups_trace.txt <<'PY'
import glob, os, sqlite3, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    for r in rows[:8]:
        print("   %-60s %5d %10.2f %10.3f %6.2f" % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
  grep -o '"sw_kernel_ms": [0-9.]*' $OUT/log.txt | tail -1 >> gpurun_out/groups_trace.txt
done
cat gpurun_out/groups_trace.txt

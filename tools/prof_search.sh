#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of one tools/bench_search.py command -> gpurun_out/prof_<tag>/summary.txt
# usage: prof_search.sh TAG qdb 1000 30000 verysensitive
TAG=$1; shift
# rocprofv3 databases stay in /tmp on the box (gpurun merges back at most 64 MiB); the summaries are copied to gpurun_out/prof_<tag>/
OUT=/tmp/rsk_prof/prof_$TAG
KEEP=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $KEEP
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_search.py "$@" > $OUT/trace.log 2>&1 < /dev/null
{
  echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_search.py $*   (two runs of the call)"
  python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    print("kernel stats (ms): name, calls, total, average, pct")
    for r in rows[:24]:
        print("   %-66s %7d %14.2f %14.3f %6.2f" % (r[0][:66], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
  grep '"seconds"' $OUT/trace.log
} > $OUT/summary.txt 2>&1 < /dev/null
cat $OUT/summary.txt
cp $OUT/*.txt $OUT/*.json $OUT/*.log $OUT/*.err $KEEP/ 2>/dev/null

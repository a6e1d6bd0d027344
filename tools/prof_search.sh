#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of one query-vs-database -search call (all kernels of the path incl. the
# long-chain stage: k_mkf_seed, k_xdrop, k_lddt_long) -> gpurun_out/prof_<tag>/summary.txt
TAG=${1:-r01_search}
NQ=${2:-256}; ND=${3:-30000}; MODE=${4:-sensitive}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
RSK_TRACE=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_search.py qdb $NQ $ND $MODE > $OUT/trace.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_search.py qdb $NQ $ND $MODE   (two runs of the call)";
  python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    print("kernel stats (us): name, calls, total, average, pct")
    for r in rows[:24]:
        print("   %-64s %6d %14.0f %14.0f %6.2f" % (r[0][:64], r[1], r[2], r[3], r[4]))
PY
  echo "# stdout / RSK_TRACE of the same command"; grep -v "^\[rocprof\|^W2\|^I2\|^E2" $OUT/trace.log | tail -120; } > $OUT/summary.txt 2>&1

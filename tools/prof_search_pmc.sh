#!/bin/bash
# Runs on the GPU box: two rocprofv3 --pmc passes (no trace options) of one tools/bench_search.py command; per-kernel
# counter averages -> gpurun_out/prof_<tag>/pmc_summary.txt.   usage: prof_search_pmc.sh TAG qdb 256 30000 sensitive
TAG=$1; shift
OUT=/tmp/rsk_prof/prof_$TAG   # rocprofv3 databases stay in /tmp on the box; only the summary goes back
KEEP=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $KEEP
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc1 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_search.py "$@" > $OUT/pmc1.log 2>&1 < /dev/null
timeout 900 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_search.py "$@" > $OUT/pmc2.log 2>&1 < /dev/null
{
echo "# rocprofv3 --pmc (two passes) -- python tools/bench_search.py $*   per-dispatch averages; issue = SQ_INSTS_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"
python3 - "$OUT" <<'PY'
import collections, glob, os, sqlite3, sys
acc = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*", "*.db"))):
    c = sqlite3.connect(f)
    for name, ctr, avg, n, dur in c.execute("select kernel_name,counter_name,avg(value),count(*),avg(duration) from counters_collection group by kernel_name,counter_name"):
        if name.startswith("k_") or "k_sw_" in name:
            acc[name][ctr] = avg; acc[name]["_n"] = n; acc[name]["_ms"] = dur / 1e6
for name, d in sorted(acc.items(), key=lambda kv: -kv[1]["_ms"] * kv[1]["_n"]):
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    issue = d.get("SQ_INSTS_VALU", 0) * 4 / (cyc * 1024) if cyc else float("nan")
    print("%-44s n=%-4d %8.3f ms  VALU %.3g  SALU %.3g  SMEM %.3g  LDS %.3g  VMEM rd/wr %.3g/%.3g  waves %.3g  VALU issue %.2f" % (
        name[:44], d["_n"], d["_ms"], d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_SALU", 0), d.get("SQ_INSTS_SMEM", 0), d.get("SQ_INSTS_LDS", 0),
        d.get("SQ_INSTS_VMEM_RD", 0), d.get("SQ_INSTS_VMEM_WR", 0), d.get("SQ_WAVES", 0), issue))
PY
} > $OUT/pmc_summary.txt 2>&1 < /dev/null
cat $OUT/pmc_summary.txt
cp $OUT/pmc_summary.txt $KEEP/

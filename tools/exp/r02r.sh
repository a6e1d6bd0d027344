RSK_TRACE=1 python tools/bench_search.py qdb 1000 30000 verysensitive > gpurun_out/r02r.json 2> gpurun_out/r02r.err
python - <<'PY'
import re,collections
tot=collections.defaultdict(float); cnt=collections.Counter()
for ln in open('gpurun_out/r02r.err'):
    m=re.match(r'\[(\w+)\]\s+(.*?)\s+([\d.]+) ms', ln)
    if m:
        k=m.group(1)+': '+m.group(2); tot[k]+=float(m.group(3)); cnt[k]+=1
for k,v in sorted(tot.items(), key=lambda x:-x[1])[:30]: print("%9.1f ms  x%-4d %s"%(v,cnt[k],k))
PY
grep seconds gpurun_out/r02r.json

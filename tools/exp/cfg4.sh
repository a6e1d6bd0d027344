for f in 3 2; do
RSK_ALIGN_INFLIGHT=$f python bench.py --configs-only config4 2>&1 | tail -1 > gpurun_out/cfg4_$f.json
python - <<PY
import json
d=json.load(open("gpurun_out/cfg4_$f.json"))
def walk(d):
    for k,v in d.items():
        if isinstance(v,dict):
            if "seconds" in v: print("inflight $f", k, round(v["seconds"],2))
            else: walk(v)
walk(d)
PY
done

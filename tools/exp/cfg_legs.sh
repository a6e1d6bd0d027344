python -m pytest tests/test_gpu_search.py tests/test_gpu_configs.py tests/test_gpu_multidev.py -x -q 2>&1 | tail -2
python bench.py --configs-only config3,config4 2>&1 | tail -1 > gpurun_out/cfg.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/cfg.json"))
for k,v in d.items():
    if isinstance(v,dict) and "seconds" in v: print(k, round(v["seconds"],2), v.get("vs_reference_on_sample"))
PY

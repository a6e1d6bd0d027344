timeout 1200 python -m pytest tests/test_gpu_align.py tests/test_gpu_search.py tests/test_gpu_vs_reference_binary.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8
timeout 300 python bench.py --live-only > gpurun_out/r02x_live.out 2>gpurun_out/r02x_live.err; echo rc=$?
python - <<'PY'
import json
r=json.load(open('gpurun_out/r02x_live.out'))['roofline_live']
for k in r: print(k['kernel'], round(k['kernel_ms'],2), k.get('cells_per_s'))
PY

timeout 900 python -m pytest tests/test_gpu_mu_sw.py tests/test_gpu_search.py tests/test_gpu_vs_reference_binary.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5
python tools/bench_kernels.py 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin)
for k in ('mu_sw_fwd','mu_sw_rev','mu_filter_sensitive','mu_filter_fast'): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in r[k].items()})"
RSK_MUSW_EARLY_EXIT=0 python tools/bench_kernels.py 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin)
for k in ('mu_filter_sensitive','mu_filter_fast'): print('no-early-exit', k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in r[k].items()})"

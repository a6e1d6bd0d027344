export RSK_LIB=$PWD/reseek_amd/librsk_sst.so
timeout 300 python bench.py --live-only > gpurun_out/r02w_sst.out 2>gpurun_out/r02w_sst.err; echo rc=$?
python - <<'PY'
import json
r=json.load(open('gpurun_out/r02w_sst.out'))['roofline_live']
for k in r: print(k['kernel'], round(k['kernel_ms'],2), k.get('cells_per_s'))
PY
tail -3 gpurun_out/r02w_sst.err
unset RSK_LIB
RSK_TRACE=1 timeout 600 python tools/bench_search.py qdb 256 125000 sensitive > gpurun_out/r02w_c3.out 2> gpurun_out/r02w_c3.err
grep '"seconds"' gpurun_out/r02w_c3.out
grep "^\[RunQuery\]\|^\[RunPairs\]\|^\[LoadChains\]\|^\[search\]" gpurun_out/r02w_c3.err | tail -40

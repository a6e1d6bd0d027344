RSK_TRACE=1 timeout 600 python tools/bench_search.py qdb 1000 30000 verysensitive > gpurun_out/r02x_c4.out 2> gpurun_out/r02x_c4.err
grep '"seconds"' gpurun_out/r02x_c4.out
grep "^\[pool\]" gpurun_out/r02x_c4.err | awk -F'took' '{ if ($2+0 > 5) print }' | cut -c1-140
grep "format (threads)" gpurun_out/r02x_c4.err | sed 's/.*@//' | awk 'NR>1{printf "%.0f ", $1-p} {p=$1}' | fold -w 200
timeout 600 python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | grep '"seconds"'
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_align.py -x -q 2>&1 | tail -3

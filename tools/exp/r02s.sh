RSK_TRACE=1 python tools/bench_search.py qdb 1000 30000 verysensitive > gpurun_out/r02s.json 2> gpurun_out/r02s.err
grep "^\[pool\]" gpurun_out/r02s.err | head -40; grep -c "^\[pool\]" gpurun_out/r02s.err
grep "alloc+h2d" gpurun_out/r02s.err | head -45 | awk '{print $3}' | tr '\n' ' '

timeout 900 python -m pytest tests/test_gpu_dss_density.py -x -q 2>&1 | tail -8
RSK_TRACE=1 timeout 600 python tools/bench_search.py qdb 256 125000 sensitive > gpurun_out/r03l.out 2> gpurun_out/r03l.err
grep '"seconds"' gpurun_out/r03l.out
grep "^\[LoadChains\]" gpurun_out/r03l.err | tail -8 | cut -c1-160
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3

timeout 900 python -m pytest tests/test_gpu_mkf.py tests/test_gpu_search.py -x -q 2>&1 | tail -3
RSK_MKF_OVERLAP=0 RSK_TRACE=1 timeout 300 python tools/bench_search.py qdb 256 30000 sensitive 2> gpurun_out/r02z_s.err | grep '"seconds"'
grep "rsk_mkf_seed_pairs\]" gpurun_out/r02z_s.err | cut -c1-200

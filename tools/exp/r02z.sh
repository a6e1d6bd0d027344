timeout 900 python -m pytest tests/test_gpu_xdrop.py tests/test_gpu_mkf.py tests/test_gpu_search.py -x -q 2>&1 | tail -3
timeout 300 python tools/bench_search.py qdb 256 30000 sensitive 2>/dev/null | grep '"seconds"'
timeout 600 python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | grep '"seconds"'

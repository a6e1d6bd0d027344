timeout 600 python -m pytest tests/test_gpu_xdrop.py tests/test_gpu_mkf.py -x -q 2>&1 | tail -3
RSK_MKF_OVERLAP=0 bash tools/prof_search.sh r02z_c3s qdb 256 30000 sensitive | head -8

timeout 1700 python -m pytest tests/test_gpu_configs.py tests/test_gpu_gapless.py tests/test_gpu_vs_reference_binary.py tests/test_gpu_mkf.py -x -q 2>&1 | tail -15

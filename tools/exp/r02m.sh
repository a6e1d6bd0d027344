timeout 900 bash tools/tsan_host.sh 2>&1 | tail -60
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8

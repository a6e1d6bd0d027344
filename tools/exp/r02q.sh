python tools/bench_search.py qdb 1000 87500 verysensitive 2>/dev/null | grep '"seconds"'
python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | grep '"seconds"'
RSK_TRACE=1 python tools/bench_search.py qdb 1000 87500 verysensitive 2>&1 | grep "^\[RunQuery\|^\[RunPairs\|^\[LoadChains\|ReplayBatch\] format\|kernels+d2h\|classify\|paths d2h\|host stats\|offsets" | tail -60

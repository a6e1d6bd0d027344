python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | grep -A3 '"run1"' | tail -3
python tools/bench_search.py 0 sensitive bca 2>/dev/null | grep -A3 '"run1"' | tail -3
python tools/bench_search.py 0 sensitive 2>/dev/null | grep -A3 '"run1"' | tail -3
python tools/bench_search.py qdb 1000 87500 verysensitive 2>/dev/null | grep -A3 '"run1"' | tail -3
RSK_TRACE=1 python tools/bench_search.py qdb 1000 30000 verysensitive 2>&1 | grep -v "^\[rsk_align\|ReplayBatch\|LoadChains\|SelfRev\|LoadBCA" | tail -30

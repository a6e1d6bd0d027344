for sc in 131072 32768 16384; do
echo "== RSK_STREAM_CHAINS=$sc"
RSK_STREAM_CHAINS=$sc python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | grep -A1 '"run1"' | tail -1
done
RSK_STREAM_CHAINS=32768 RSK_TRACE=1 python tools/bench_search.py qdb 256 125000 sensitive 2>&1 | grep "^\[RunQuery\|^\[RunPairs\|^\[LoadChains\|^\[SelfRev\] GPU\|RunMKFPairs\]\|mkf_align" | tail -40

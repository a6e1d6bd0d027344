for sc in 131072 32768 16384 32768 131072 16384; do
echo "== RSK_STREAM_CHAINS=$sc"
RSK_STREAM_CHAINS=$sc python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | grep '"seconds"'
done

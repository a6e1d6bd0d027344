RSK_TRACE=1 timeout 900 python tools/bench_search.py qdb 1000 87500 verysensitive > gpurun_out/r03c.out 2> gpurun_out/r03c.err
grep '"seconds"' gpurun_out/r03c.out
grep "^\[pool\]" gpurun_out/r03c.err | awk -F'took' '{ if ($2+0 > 5) print }' | cut -c1-140 | head
grep "^\[RunPairs\]\|^\[RunQuery\]\|^\[LoadChains\]" gpurun_out/r03c.err | tail -30 | cut -c1-120
grep "format (threads)" gpurun_out/r03c.err | sed 's/.*@//' | awk 'NR>1{printf "%.0f ", $1-p} {p=$1}' | fold -w 220 | tail -8

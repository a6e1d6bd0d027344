RSK_TRACE=1 timeout 600 python tools/bench_search.py qdb 256 125000 sensitive > gpurun_out/r03j.out 2> gpurun_out/r03j.err
grep '"seconds"' gpurun_out/r03j.out
grep "^\[RunQuery\]\|^\[LoadChains\]\|^\[LoadBCA\]\|^\[SelfRev\]\|^\[RunPairs\] align\|^\[RunPairs\] filter" gpurun_out/r03j.err | tail -42 | cut -c1-120

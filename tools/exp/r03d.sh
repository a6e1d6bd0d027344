for t in 8192 6144 4096 3072; do
echo "== RSK_SWQ_MIN_LANES=$t"
RSK_SWQ_MIN_LANES=$t RSK_TRACE=1 timeout 300 python tools/bench_search.py qdb 1000 30000 verysensitive 2> gpurun_out/r03d.err | grep '"seconds"'
grep "kernels+d2h" gpurun_out/r03d.err | awk '{s+=$3; n++} END {print "kernels+d2h mean", s/n, n}'
grep "classes" gpurun_out/r03d.err | sed -n 5p | cut -c1-200
done

timeout 1200 python -m pytest tests/test_gpu_mkf.py tests/test_gpu_search.py tests/test_gpu_vs_reference_binary.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
for w in 0 1; do
echo "== RSK_XDROP_WAVE=$w"
RSK_XDROP_WAVE=$w RSK_TRACE=1 timeout 300 python tools/bench_search.py qdb 256 30000 sensitive 2> gpurun_out/r02z_$w.err | grep '"seconds"'
grep "rsk_mkf_align_pairs\]\|RunMKFPairs\] .*chaining" gpurun_out/r02z_$w.err | cut -c1-260
done

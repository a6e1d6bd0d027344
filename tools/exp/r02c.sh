mkdir -p gpurun_out/r02c; O=gpurun_out/r02c
RSK_SWQ_MIN_LANES=1 timeout 300 python -m pytest tests/test_gpu_align.py tests/test_gpu_search.py -x -q 2>&1 | tail -3 > $O/tests.txt
for ml in 1000000000 64 160 320 640 1280 2560 5120; do
  RSK_SWQ_MIN_LANES=$ml RSK_TRACE=1 timeout 120 python tools/bench_align.py 3 >> $O/align.jsonl 2>> $O/align.err
done
cat $O/tests.txt; cat $O/align.jsonl; grep "bucket\|classes" $O/align.err | sort | uniq -c | sort -k2 | head -60

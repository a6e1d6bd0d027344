mkdir -p gpurun_out/r02b; O=gpurun_out/r02b
timeout 300 python -m pytest tests/test_gpu_align.py tests/test_gpu_search.py -x -q 2>&1 | tail -3 > $O/tests.txt
RSK_SWQ_MIN_COUNT=1 RSK_SWQ_SMALL_BUCKET=3 timeout 300 python -m pytest tests/test_gpu_align.py tests/test_gpu_search.py -x -q 2>&1 | tail -3 >> $O/tests.txt
RSK_SWQ_MIN_COUNT=1 RSK_SWQ_SMALL_BUCKET=2 RSK_SWQ_MIN_LANES=1000000 timeout 300 python -m pytest tests/test_gpu_align.py tests/test_gpu_search.py -x -q 2>&1 | tail -3 >> $O/tests.txt
RSK_SWQ_MIN_COUNT=1 RSK_SWQ_SMALL_BUCKET=1 RSK_SWQ_MIN_LANES=1000000 timeout 300 python -m pytest tests/test_gpu_align.py -x -q 2>&1 | tail -3 >> $O/tests.txt
for cfg in "RSK_SWQ_MIN_COUNT=0" "RSK_SWQ_MIN_COUNT=8 RSK_SWQ_SMALL_BUCKET=2" "RSK_SWQ_MIN_COUNT=8 RSK_SWQ_SMALL_BUCKET=3" "RSK_SWQ_MIN_COUNT=8 RSK_SWQ_SMALL_BUCKET=1" "RSK_SWQ_MIN_COUNT=4 RSK_SWQ_SMALL_BUCKET=2" "RSK_SWQ_MIN_COUNT=16 RSK_SWQ_SMALL_BUCKET=2" "RSK_SWQ_MIN_COUNT=8 RSK_SWQ_SMALL_BUCKET=2 RSK_SWQ_MIN_LANES=1000000" "RSK_SWQ_MIN_COUNT=8 RSK_SWQ_SMALL_BUCKET=3 RSK_SWQ_MIN_LANES=1000000" "RSK_SWQ_MIN_COUNT=0 RSK_SWQ_MIN_LANES=1000000"; do
  env $cfg RSK_TRACE=1 timeout 120 python tools/bench_align.py 3 >> $O/align.jsonl 2>> $O/align.err
done
cat $O/tests.txt; cat $O/align.jsonl

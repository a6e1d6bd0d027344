O=gpurun_out/r02f; mkdir -p $O
for v in r16w2 r12w2 r20w2 r24w2; do
  if [ -n "$v" ]; then export RSK_LIB=$PWD/reseek_amd/librsk_$v.so; fi
  echo "variant=$v" >> $O/align.jsonl
  timeout 300 python -m pytest tests/test_gpu_align.py -x -q 2>&1 | tail -1 >> $O/align.jsonl
  timeout 120 python tools/bench_align.py 3 >> $O/align.jsonl 2>> $O/align.err
done
cat $O/align.jsonl

O=gpurun_out/r02g; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -5 $O/bench.err
bash tools/prof_live.sh r02_live > $O/prof_live.out 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

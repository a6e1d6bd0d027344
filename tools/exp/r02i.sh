O=gpurun_out/r02i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fast_shards.py tests/test_gpu_dist.py -x -q 2>&1 | tail -30
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-search --no-live > $O/bench1.json 2> $O/bench1.err; tail -2 $O/bench1.err; python -c "
import json; r=json.load(open('$O/bench1.json')); print(r['value'], r['ms_per_step'], r['scaling'], r['config']['pairs_total'], r['roofline']['frac'], r['roofline']['hbm']['frac'])"

O=gpurun_out/r02u; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/prof_live.sh r02u_live > $O/prof_live.out 2>&1
bash tools/prof_bench.sh r02u_bench > $O/prof_bench.out 2>&1
tail -3 $O/prof_bench.out

bash tools/prof_search.sh r02l_search 256 30000 sensitive > /dev/null 2>&1
head -30 gpurun_out/prof_r02l_search/summary.txt
grep "RunMKFPairs\|rsk_mkf_align\|RunPairs\]\|RunQuery\]" gpurun_out/prof_r02l_search/summary.txt | tail -12
python tools/bench_search.py qdb 256 125000 sensitive 2>/dev/null | tail -22
python tools/bench_search.py 0 sensitive bca 2>/dev/null | tail -14

# end-of-round measurement set: live-kernel PMC, bench line, search traces of the config-3 / config-4 shapes
timeout 1500 bash tools/prof_live.sh r02z_live > gpurun_out/r03a_live.log 2>&1 < /dev/null
timeout 900 python bench.py > gpurun_out/r02z_bench.json 2> gpurun_out/r02z_bench.err < /dev/null
tail -c 600 gpurun_out/r02z_bench.json
timeout 600 bash tools/prof_search.sh r02z_c3 qdb 256 30000 sensitive > /dev/null 2>&1 < /dev/null
timeout 600 bash tools/prof_search.sh r02z_c4 qdb 1000 30000 verysensitive > /dev/null 2>&1 < /dev/null
echo done

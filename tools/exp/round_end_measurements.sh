# end-of-round measurement set: live-kernel PMC, bench line, search traces of the config-3 / config-4 shapes
# (copy gpurun_out/prof_<tag>/summary.txt and the bench json into profiles/ afterwards)
TAG=${1:-r02z}
if [ "$2" = "pmc" ]; then timeout 1500 bash tools/prof_live.sh ${TAG}_live > gpurun_out/${TAG}_live.log 2>&1 < /dev/null; fi
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err < /dev/null
tail -c 400 gpurun_out/${TAG}_bench.json
timeout 600 bash tools/prof_search.sh ${TAG}_c3 qdb 256 30000 sensitive > /dev/null 2>&1 < /dev/null
timeout 600 bash tools/prof_search.sh ${TAG}_c4 qdb 1000 30000 verysensitive > /dev/null 2>&1 < /dev/null
echo done

# end-of-round measurement set: live-kernel PMC, bench line, search traces of the config-2/3/4 shapes and the self search
# (copy gpurun_out/prof_<tag>/summary.txt and the bench json into profiles/ afterwards)
TAG=${1:-r03}
timeout 1500 bash tools/prof_live.sh ${TAG}_live > gpurun_out/${TAG}_live.log 2>&1 < /dev/null
timeout 1700 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err < /dev/null
tail -c 300 gpurun_out/${TAG}_bench.json
timeout 600 bash tools/prof_search.sh ${TAG}_self 0 sensitive > /dev/null 2>&1 < /dev/null
timeout 600 bash tools/prof_search.sh ${TAG}_c3 qdb 256 125000 sensitive > /dev/null 2>&1 < /dev/null
timeout 600 bash tools/prof_search.sh ${TAG}_c4 qdb 1000 87500 verysensitive > /dev/null 2>&1 < /dev/null
timeout 900 bash tools/prof_search.sh ${TAG}_c2 0 fast bca db > /dev/null 2>&1 < /dev/null
RSK_ALIGN_INFLIGHT=1 timeout 600 bash tools/prof_search.sh ${TAG}_c4serial qdb 1000 30000 verysensitive > /dev/null 2>&1 < /dev/null
echo done

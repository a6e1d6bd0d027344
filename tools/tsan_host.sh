#!/bin/bash
# ThreadSanitizer pass over the threaded HOST code of librsk: the host/*.cpp sources are rebuilt with
# clang++ -fsanitize=thread (the .hip objects stay as built: their host parts run on the caller's thread or under
# rsk_parallel_for, the HIP runtime itself is not instrumented) into reseek_amd/librsk_tsan.so, and the reference-shaped
# driver tests/ref_shaped/search_main.cpp (pure C++, no Python) runs the searches that put the most threads in flight:
# two GPU stages on two contexts + the long-chain job beside them + the streaming loader + the worker pools.
# Runs on the GPU box; report -> gpurun_out/tsan/report.txt (copy to profiles/).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/tsan; mkdir -p $OUT build/tsan
HIPINC=/opt/rocm/include
CXX=/opt/rocm/lib/llvm/bin/clang++
for f in reseek_amd/csrc/host/*.cpp; do
  $CXX -std=c++17 -O1 -g -fPIC -fsanitize=thread -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I include -I $HIPINC -c $f -o build/tsan/$(basename $f).o || exit 1
done
[ -f build/obj/rsk_api.hip.o ] || python -c "import __graft_entry__ as g; g.build()"
OBJS=$(ls build/obj/*.hip.o)
$CXX -shared -fPIC -fsanitize=thread $OBJS build/tsan/*.cpp.o -L/opt/rocm/lib -lamdhip64 -o reseek_amd/librsk_tsan.so || exit 1
$CXX -std=c++17 -O1 -g -fsanitize=thread -I reseek_amd/csrc/host tests/ref_shaped/search_main.cpp -L reseek_amd -lrsk_tsan -Wl,-rpath,$PWD/reseek_amd -Wl,-rpath,/opt/rocm/lib -pthread -o build/tsan/search_main || exit 1
W=$(mktemp -d)
for n in q100 palms; do gzip -dc tests/golden/$n.bca.gz > $W/$n.bca; done
for n in q100_sensitive palms_sensitive; do gzip -dc tests/golden/$n.rskdb.gz > $W/$n.rskdb; done      # RSKDB1 containers: the loader builds the chains on the host threads
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan.supp"
export RSK_BATCH_PAIRS=700 RSK_STREAM_CHAINS=9        # many alignment batches and streamed DB batches on small inputs
: > $OUT/report.txt
run() { echo "== search_main $*" >> $OUT/report.txt; ( cd $W && $OLDPWD/build/tsan/search_main "$@" ) >> $OUT/report.txt 2>&1; echo "exit $?" >> $OUT/report.txt; }
run q100.bca -sensitive -output $W/a.tsv
run palms.bca -sensitive -output $W/b.tsv
run q100.bca -db q100.bca -sensitive -output $W/c.tsv
run q100.bca -db palms.bca -sensitive -output $W/d.tsv
run q100.bca -db q100.bca -fast -output $W/e.tsv
run q100.bca -verysensitive -output $W/f.tsv
# r03: one searcher driving several device contexts (host thread + context per list entry, concurrent writers of one hits file)
run q100.bca -sensitive -devices 0,0,0 -output $W/g.tsv
run palms.bca -sensitive -devices 0,0 -output $W/h.tsv
run q100.bca -db palms.bca -sensitive -devices 0,0,0 -output $W/i.tsv
run q100.bca -db q100.bca -fast -devices 0,0 -output $W/j.tsv
# r04: the container loader (chains built in parallel) and the parallel release of a chain set
run q100_sensitive.rskdb -sensitive -output $W/k.tsv
run palms_sensitive.rskdb -sensitive -output $W/l.tsv
wc -l $W/*.tsv >> $OUT/report.txt
echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' $OUT/report.txt)" | tee -a $OUT/report.txt
grep -A12 "WARNING: ThreadSanitizer" $OUT/report.txt | head -80

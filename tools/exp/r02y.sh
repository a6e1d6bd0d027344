for lib in old new; do
  if [ $lib = old ]; then export RSK_LIB=$PWD/reseek_amd/librsk_old.so; else unset RSK_LIB; fi
  RSK_TRACE=1 timeout 300 python tools/bench_search.py qdb 256 30000 sensitive 2> gpurun_out/r02y_$lib.err | grep '"seconds"'
  echo "== $lib"; grep "kernels+d2h\|classes" gpurun_out/r02y_$lib.err | tail -8 | cut -c1-200
done

#!/bin/bash
# bench the headline kernel with the in-tree library and with build/var_<name>/librsk.so for every name given
for v in base "$@"; do
  if [ $v = base ]; then unset RSK_LIB; else export RSK_LIB=$PWD/build/var_$v/librsk.so; fi
  timeout 280 python bench.py --no-cpu-baseline --no-search --no-live --no-configs > gpurun_out/b_$v.json 2> gpurun_out/b_$v.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/b_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('slot_efficiency'))
"
done

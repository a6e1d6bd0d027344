#!/bin/bash
# build/var_<name>/librsk.so = the library as it was at git revision <rev> (every source of that revision, same flags as
# __graft_entry__.build); RSK_LIB selects it in an A/B run (capi.py skips symbols the old build lacks).
# usage: tools/exp/build_rev.sh <name> <rev>
set -e
cd "$(dirname "$0")/../.."
name=$1; rev=$2
d=build/rev_$name
rm -rf $d; mkdir -p $d build/var_$name
git archive $rev reseek_amd/csrc include | tar -x -C $d
pids=()
for src in $d/reseek_amd/csrc/*.hip $d/reseek_amd/csrc/host/*.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I $d/include -c $src -o $d/$(basename $src).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o build/var_$name/librsk.so
rm -rf $d
echo build/var_$name/librsk.so $rev

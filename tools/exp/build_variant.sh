#!/bin/bash
# build/var_<name>/librsk.so = the library with ONE source recompiled under extra -D flags (timing experiments; RSK_LIB selects it)
# usage: tools/exp/build_variant.sh <name> <source.hip> <flags...>
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
python __graft_entry__.py > /dev/null
mkdir -p build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I include "$@" -c reseek_amd/csrc/$src -o build/var_$name/$src.o
objs=$(ls build/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/var_$name/$src.o -o build/var_$name/librsk.so
echo build/var_$name/librsk.so

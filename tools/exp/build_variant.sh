#!/bin/bash
# build_variant.sh NAME "-DSWF_R=12 -DSWF_WPE=3" FILE.hip : librsk_NAME.so with one source recompiled under extra flags
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FLAGS=$2; SRC=$3
mkdir -p build/var_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I include $FLAGS -c reseek_amd/csrc/$SRC -o build/var_$NAME/$SRC.o
OBJS=$(ls build/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/var_$NAME/$SRC.o -o reseek_amd/librsk_$NAME.so
echo built reseek_amd/librsk_$NAME.so

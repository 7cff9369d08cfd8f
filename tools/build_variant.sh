#!/bin/bash
# A/B builds of the kernel library: tools/build_variant.sh NAME "EXTRA FLAGS" [TUs to recompile, default all five]
#   -> cudasw4_amd/lib_NAME/libcudasw4_amd.so (the other objects are taken from the main build: build that first).
# Use with CUDASW4_AMD_LIB=cudasw4_amd/lib_NAME/libcudasw4_amd.so (capi.py; the Python mirror and the sweep tools).
set -e
cd "$(dirname "$0")/../cudasw4_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
TUS=${*:-"sw_api sw_kind_f16x2 sw_kind_i16x2 sw_kind_i32 sw_kind_f32"}
OUT=../lib_$NAME; mkdir -p $OUT/obj
cp -n ../lib/obj/*.o $OUT/obj/   # objects this variant has already built stay
for t in $TUS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None $FLAGS -c $t.hip -o $OUT/obj/$t.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj/*.o -o $OUT/libcudasw4_amd.so
ls -la $OUT/libcudasw4_amd.so

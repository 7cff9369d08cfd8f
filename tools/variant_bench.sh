#!/bin/bash
# A/B of kernel-library variants (tools/build_variant.sh NAME ...) in ONE GPU session: the peak benchmark through the
# Python mirror with per-query rates (tools/peak_sweep.py), half2, L = 512 (and what else is given).
#   tools/variant_bench.sh "A B C" [lengths] [kernels]
VARIANTS=${1:-"A"}; LENGTHS=${2:-512}; KERNELS=${3:-half2}
mkdir -p gpurun_out/variants
for v in main $VARIANTS; do
    if [ $v = main ]; then lib=cudasw4_amd/lib/libcudasw4_amd.so; else lib=cudasw4_amd/lib_$v/libcudasw4_amd.so; fi
    [ -f $lib ] || { echo "$v: no $lib"; continue; }
    for rep in 1 2; do
        CUDASW4_AMD_LIB=$PWD/$lib python tools/peak_sweep.py --lengths $LENGTHS --kernels $KERNELS --json gpurun_out/variants/$v.$rep.json > gpurun_out/variants/$v.$rep.txt 2>&1
        echo "== $v rep $rep"; grep -E "gcups" gpurun_out/variants/$v.$rep.txt | head -12
    done
done

#!/bin/bash
# one variant library, several values of an environment knob:  tools/variant_bench2.sh C CUDASW4_AMD_MAX_ROWS_MULTI "0 32 36 40"
V=$1; KNOB=$2; VALUES=$3; LENGTHS=${4:-512}
lib=$PWD/cudasw4_amd/lib_$V/libcudasw4_amd.so; [ $V = main ] && lib=$PWD/cudasw4_amd/lib/libcudasw4_amd.so
mkdir -p gpurun_out/variants
for val in $VALUES; do
    for rep in 1 2; do
        echo "== $V $KNOB=$val rep $rep"
        env $KNOB=$val CUDASW4_AMD_LIB=$lib python tools/peak_sweep.py --lengths $LENGTHS --kernels half2 2>&1 | grep gcups
    done
done

#!/bin/bash
# Every counter-backed figure bench.py reports, in one gpurun call (profiles/kernel_counters.json is extended entry by
# entry, so all configurations must be collected on the same kernel sources):
#   tools/collect_all_profiles.sh r03 <commit>      -> gpurun_out/profiles_r03/  (copy into profiles/)
TAG=${1:-r03}; COMMIT=${2:-unknown}
rm -f profiles/kernel_counters.json
bash tools/collect_profiles.sh $TAG $COMMIT > /dev/null 2>&1                                              # BASELINE config 2 (headline)
bash tools/collect_profiles.sh $TAG $COMMIT --workload sprot-like > /dev/null 2>&1                        # config 3
export PASSES="stats traffic valu"
bash tools/collect_profiles.sh $TAG $COMMIT --kernel float > /dev/null 2>&1
bash tools/collect_profiles.sh $TAG $COMMIT --kernel dpxs32 > /dev/null 2>&1                              # int32 results in fp32 lanes
CUDASW4_AMD_I32_NATIVE=1 bash tools/collect_profiles.sh $TAG $COMMIT --kernel dpxs32 > /dev/null 2>&1     # the int32 kernels themselves
export PASSES="traffic valu"
bash tools/collect_profiles.sh $TAG $COMMIT --kernel dpxs16 > /dev/null 2>&1
bash tools/collect_profiles.sh $TAG $COMMIT --max-gpu-mem 600M > /dev/null 2>&1                           # hybrid residency, half2
bash tools/collect_profiles.sh $TAG $COMMIT --kernel dpxs32 --max-gpu-mem 600M > /dev/null 2>&1           # config 5's route on the peak DB
bash tools/collect_profiles.sh $TAG $COMMIT --workload sprot-like --kernel dpxs32 > /dev/null 2>&1        # config 5's kernels on ragged subjects
bash tools/collect_profiles.sh $TAG $COMMIT --workload sprot-like --max-gpu-mem 260M --max-batch-bytes 16M > /dev/null 2>&1   # hybrid, ragged
cp profiles/kernel_counters.json gpurun_out/profiles_$TAG/kernel_counters.json
ls gpurun_out/profiles_$TAG; cat gpurun_out/profiles_$TAG/kernel_counters.json | head -60

#!/bin/bash
# The counter passes bench.py's default line and the GPU suite need, when the full tools/collect_all_profiles.sh (10
# configurations, ~3 minutes per rocprofv3 pass) does not fit the GPU budget: peak half2 (headline), Swiss-Prot-like dpx,
# peak float, peak dpxs32 — VALU instructions and HBM-side traffic each, kernel stats for the first two.
#   tools/collect_min_profiles.sh r04 <commit> [skip-first]
set -u
TAG=${1:-r04}; COMMIT=${2:-unknown}; SKIP_FIRST=${3:-0}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT/logs
export KERNEL_COUNTERS_OUT=$PWD/$OUT/kernel_counters.building.json
if [ "$SKIP_FIRST" = 0 ]; then
    rm -f $KERNEL_COUNTERS_OUT
    PASSES="stats traffic valu lds clock" bash tools/collect_profiles.sh $TAG $COMMIT > $OUT/logs/min_01_peak_half2.log 2>&1
fi
PASSES="stats traffic valu" bash tools/collect_profiles.sh $TAG $COMMIT --workload sprot-like > $OUT/logs/min_02_sprot_dpx.log 2>&1
PASSES="traffic valu" bash tools/collect_profiles.sh $TAG $COMMIT --kernel float > $OUT/logs/min_03_float.log 2>&1
PASSES="traffic valu" bash tools/collect_profiles.sh $TAG $COMMIT --kernel dpxs32 > $OUT/logs/min_04_dpxs32.log 2>&1
python3 - "$KERNEL_COUNTERS_OUT" <<'PY'
import json, sys
have = json.load(open(sys.argv[1])).get("valu_instr_per_unit", {})
print("entries:", sorted(have))
PY
cp $KERNEL_COUNTERS_OUT $OUT/kernel_counters.json
ls $OUT

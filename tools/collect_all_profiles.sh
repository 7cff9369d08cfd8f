#!/bin/bash
# Every counter-backed figure bench.py reports, in one gpurun call (kernel_counters.json is extended entry by entry, so
# all configurations must be collected on the same kernel sources):
#   tools/collect_all_profiles.sh r04 <commit>      -> gpurun_out/profiles_r04/  (copy into profiles/)
# The counters file is BUILT under gpurun_out/ (KERNEL_COUNTERS_OUT): the tracked profiles/kernel_counters.json is only
# replaced at the end, and only when every configuration below has produced its entry; the log of every pass is kept
# under gpurun_out/profiles_<tag>/logs/, and a failing pass is reported (exit code 1) instead of discarded.
set -u
TAG=${1:-r04}; COMMIT=${2:-unknown}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT/logs
export KERNEL_COUNTERS_OUT=$PWD/$OUT/kernel_counters.building.json
rm -f $KERNEL_COUNTERS_OUT
FAILED=0
N=0
run() {  # expected counters key, then collect_profiles.sh arguments
    local key=$1; shift
    N=$((N + 1))
    local log=$OUT/logs/$(printf %02d $N)_$(echo "$key" | tr ':' '_').log
    if ! bash tools/collect_profiles.sh $TAG $COMMIT "$@" > $log 2>&1; then echo "FAILED: $key (see $log)"; FAILED=1; fi
    EXPECT="$EXPECT $key"
}
EXPECT=""
run peak:half2:resident                                                      # BASELINE config 2 (headline)
run sprot-like:dpx:resident --workload sprot-like                            # config 3
export PASSES="stats traffic valu"
run peak:float:resident --kernel float
run peak:dpxs32:resident --kernel dpxs32                                     # int32 results in fp32 lanes
CUDASW4_AMD_I32_NATIVE=1 run peak:dpxs32:resident:i32native --kernel dpxs32  # the int32 kernels themselves
export PASSES="traffic valu"
run peak:dpxs16:resident --kernel dpxs16
run peak:half2:hybrid --max-gpu-mem 600M                                     # hybrid residency, half2
run peak:dpxs32:hybrid --kernel dpxs32 --max-gpu-mem 600M                    # config 5's route on the peak DB
run sprot-like:dpxs32:resident --workload sprot-like --kernel dpxs32         # config 5's kernels on ragged subjects
run sprot-like:dpx:hybrid --workload sprot-like --max-gpu-mem 260M --max-batch-bytes 16M   # hybrid, ragged
python3 - "$KERNEL_COUNTERS_OUT" $EXPECT <<'PY'
import json, sys
path, expect = sys.argv[1], sys.argv[2:]
try:
    have = json.load(open(path)).get("valu_instr_per_unit", {})
except (OSError, ValueError):
    have = {}
missing = [k for k in expect if k not in have]
print("counters entries: %d of %d%s" % (len(expect) - len(missing), len(expect), "" if not missing else "  MISSING: " + " ".join(missing)))
sys.exit(1 if missing else 0)
PY
COMPLETE=$?
if [ $COMPLETE -eq 0 ] && [ $FAILED -eq 0 ]; then
    mv $KERNEL_COUNTERS_OUT $OUT/kernel_counters.json
    cp $OUT/kernel_counters.json profiles/kernel_counters.json
    echo "profiles/kernel_counters.json replaced"
else
    echo "counters file incomplete: left at $KERNEL_COUNTERS_OUT, profiles/kernel_counters.json untouched"
fi
ls $OUT
exit $(( FAILED | COMPLETE ))

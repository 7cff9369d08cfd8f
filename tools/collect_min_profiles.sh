#!/bin/bash
# The counter passes bench.py's default line and the GPU suite need, when the full tools/collect_all_profiles.sh (10
# configurations, ~3 minutes per rocprofv3 pass) does not fit the GPU budget: peak half2 (headline), Swiss-Prot-like dpx,
# peak float, peak dpxs32 — VALU instructions and HBM-side traffic each, kernel stats for the first two.
#   tools/collect_min_profiles.sh r05 <commit> [skip-first]
# Like collect_all_profiles.sh: every pass's exit code is tracked, the four expected keys are verified, and the counters file
# is copied out (gpurun_out/profiles_<tag>/kernel_counters.json) only when it is complete; exit code 1 otherwise.
set -u
TAG=${1:-r05}; COMMIT=${2:-unknown}; SKIP_FIRST=${3:-0}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT/logs
export KERNEL_COUNTERS_OUT=$PWD/$OUT/kernel_counters.building.json
FAILED=0
EXPECT=""
run() {  # expected counters key, log name, PASSES, then collect_profiles.sh arguments
    local key=$1 log=$OUT/logs/$2 passes=$3; shift 3
    if ! PASSES="$passes" bash tools/collect_profiles.sh $TAG $COMMIT "$@" > $log 2>&1; then echo "FAILED: $key (see $log)"; FAILED=1; fi
    EXPECT="$EXPECT $key"
}
if [ "$SKIP_FIRST" = 0 ]; then
    rm -f $KERNEL_COUNTERS_OUT
    run peak:half2:resident min_01_peak_half2.log "stats traffic valu lds clock"
else
    EXPECT="peak:half2:resident"
fi
run sprot-like:dpx:resident min_02_sprot_dpx.log "stats traffic valu" --workload sprot-like
run peak:float:resident min_03_float.log "traffic valu" --kernel float
run peak:dpxs32:resident min_04_dpxs32.log "traffic valu" --kernel dpxs32
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
    cp $KERNEL_COUNTERS_OUT $OUT/kernel_counters.json
    echo "$OUT/kernel_counters.json written"
else
    echo "counters file incomplete: left at $KERNEL_COUNTERS_OUT, nothing copied"
fi
ls $OUT
exit $(( FAILED | COMPLETE ))

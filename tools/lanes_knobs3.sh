#!/bin/bash
# re-score grid cap and slot reserve with two lanes: 125 000- and 62 500-subject shards of the peak DB, short-query streams
# (an experiment of round 4: CUDASW4_AMD_RESCORE_CAP belonged to a library build that capped the re-score launches' grids by
# the recent re-score counts — measured without effect and not kept, profiles/r04_results.md; the variable is ignored now)
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["verified"], d["config"]["tail_overlaps"], d["ms_per_step"])'
run() { echo "== $* size $SIZE"; env "$@" BENCH_PIPELINE=1 timeout 400 python bench.py --no-sweep --no-secondary --no-cpu-baseline --steps 6 --warmup 2 --workload peak --db-size $SIZE 2>/dev/null | python -c "$P"; }
for SIZE in 125000 62500; do
run CUDASW4_AMD_RESCORE_CAP=0 CUDASW4_AMD_LANE_RESERVE=0
run CUDASW4_AMD_RESCORE_CAP=1 CUDASW4_AMD_LANE_RESERVE=0
run CUDASW4_AMD_RESCORE_CAP=1 CUDASW4_AMD_LANE_RESERVE=4
run CUDASW4_AMD_RESCORE_CAP=1 CUDASW4_AMD_LANE_RESERVE=16
done
for e in "CUDASW4_AMD_RESCORE_CAP=0 CUDASW4_AMD_LANE_RESERVE=0" "CUDASW4_AMD_RESCORE_CAP=1 CUDASW4_AMD_LANE_RESERVE=0" "CUDASW4_AMD_RESCORE_CAP=1 CUDASW4_AMD_LANE_RESERVE=4" "CUDASW4_AMD_RESCORE_CAP=1 CUDASW4_AMD_LANE_RESERVE=16"; do
echo "== $e"; env $e timeout 300 python tools/short_query_pipeline.py --lengths 48,96,144,222,300 2>&1 | grep -E "query residues|driver's rule|one at a time"
done

#!/bin/bash
# the tail hand-over's two knobs on a 125 000-subject shard of the peak DB: slots left free, and the gate itself
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["verified"], d["config"]["tail_overlaps"], d["ms_per_step"])'
run() { echo "== $*"; env "$@" BENCH_PIPELINE=1 timeout 400 python bench.py --no-sweep --no-secondary --no-cpu-baseline --steps 6 --warmup 2 --workload peak --db-size ${SIZE:-125000} 2>/dev/null | python -c "$P"; }
for r in 0 4 16 48 128; do run CUDASW4_AMD_LANE_RESERVE=$r; done
run CUDASW4_AMD_TAIL_GATE=0
run CUDASW4_AMD_TAIL_GATE=0 CUDASW4_AMD_LANE_RESERVE=0
run CUDASW4_AMD_TAIL_OVERLAP=0
SIZE=500000
run CUDASW4_AMD_TAIL_OVERLAP=0
run CUDASW4_AMD_TAIL_OVERLAP=1

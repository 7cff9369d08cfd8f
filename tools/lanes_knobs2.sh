#!/bin/bash
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["verified"], d["config"]["tail_overlaps"], d["ms_per_step"])'
run() { echo "== $* $Q"; env "$@" BENCH_PIPELINE=1 timeout 400 python bench.py --no-sweep --no-secondary --no-cpu-baseline --steps 20 --warmup 4 --workload peak --db-size 125000 --queries $Q 2>/dev/null | python -c "$P"; }
for Q in 0,1,2,3,4 5,6,7,8,9; do
for r in 0 16; do run CUDASW4_AMD_LANE_RESERVE=$r; done
run CUDASW4_AMD_TAIL_GATE=0 CUDASW4_AMD_LANE_RESERVE=0
run CUDASW4_AMD_TAIL_OVERLAP=0
done

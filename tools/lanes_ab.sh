#!/bin/bash
# A/B of the tail hand-over on the full-size workloads
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["verified"], d["config"]["tail_overlaps"], d["ms_per_step"])'
run() { echo "== $*"; env "$@" timeout 400 python bench.py --no-sweep --no-secondary --no-cpu-baseline --steps 4 --warmup 2 $EXTRA 2>/dev/null | python -c "$P"; }
EXTRA="--workload peak"
run BENCH_PIPELINE=0
run BENCH_PIPELINE=1 CUDASW4_AMD_TAIL_OVERLAP=0
run BENCH_PIPELINE=1 CUDASW4_AMD_TAIL_OVERLAP=1
EXTRA="--workload sprot-like --kernel dpx"
run BENCH_PIPELINE=0
run BENCH_PIPELINE=1 CUDASW4_AMD_TAIL_OVERLAP=0
run BENCH_PIPELINE=1 CUDASW4_AMD_TAIL_OVERLAP=1
EXTRA="--workload sprot-like --kernel half2"
run BENCH_PIPELINE=0
run BENCH_PIPELINE=1 CUDASW4_AMD_TAIL_OVERLAP=1
EXTRA="--workload peak --db-size 250000"
run BENCH_PIPELINE=1 CUDASW4_AMD_TAIL_OVERLAP=0
run BENCH_PIPELINE=1
EXTRA="--workload peak --db-size 62500"
run BENCH_PIPELINE=1 CUDASW4_AMD_TAIL_OVERLAP=0
run BENCH_PIPELINE=1

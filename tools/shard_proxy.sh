#!/bin/bash
# What each of N GPUs gets from the two benchmark DBs, measured on ONE GPU: a DB of 1/N of the subjects (same length
# distribution), 20 queries, two queries in flight as in multi-rank runs; with the tail hand-over (default) and without.
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%s GCUPS verified=%s gated=%s ms/step=%s" % (d["value"], d["verified"], d["config"]["tail_overlaps"], d["ms_per_step"]))'
run() { echo "== $* : $EXTRA"; env "$@" BENCH_PIPELINE=1 timeout 400 python bench.py --no-sweep --no-secondary --no-cpu-baseline --steps 6 --warmup 2 $EXTRA 2>/dev/null | python -c "$P"; }
for n in 285000 142500 71250; do
    EXTRA="--workload sprot-like --kernel dpx --db-size $n"
    run CUDASW4_AMD_TAIL_OVERLAP=0
    run X=1
done
for n in 500000 250000 125000 62500; do
    EXTRA="--workload peak --db-size $n"
    run CUDASW4_AMD_TAIL_OVERLAP=0
    run X=1
done

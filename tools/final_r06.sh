#!/bin/bash
# Round 6, one gpurun call: GPU suite, rocprofv3 passes of every configuration (-> kernel_counters.json, so that the bench
# lines behind them carry their counter-backed fractions), the default bench line under a clock, the binding bench at full
# size, a marker trace, and the full measurement pass.   tools/final_r06.sh <commit>
set -u
export TMPDIR=/tmp
COMMIT=${1:-unknown}
O=gpurun_out/final; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.txt
tail -2 $O/pytest_gpu.txt
bash tools/collect_all_profiles.sh r06 $COMMIT > $O/collect.log 2>&1
tail -3 $O/collect.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt
tail -3 $O/bench_default_time.txt
timeout 1500 python tools/binding_bench.py > $O/binding_bench.txt 2>&1
cat $O/binding_bench.txt | tail -3
# marker trace (roctx ranges of the host driver: query / batch / launch set / top-K) of three queries
R=gpurun_out/markers_raw; rm -rf $R
timeout 600 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $R -o m -- python3 bench.py --no-sweep --no-shard-proxy --no-secondary --no-cpu-baseline --no-verify --steps 1 --warmup 0 --queries 0,9,19 > $O/markers_run.log 2>&1
python3 tools/marker_summary.py $R > $O/r06_bench_markers.txt 2>&1
rm -rf $R
head -5 $O/r06_bench_markers.txt
SKIP_PYTEST=1 bash tools/full_measurement.sh > $O/full_measurement.log 2>&1
tail -30 $O/full_measurement.log

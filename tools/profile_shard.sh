#!/bin/bash
# rocprofv3 kernel statistics of a small shard of the Swiss-Prot-like DB (latency mode, the row-parallel kernel for the
# giants): tools/profile_shard.sh [subjects]  -> gpurun_out/shard_profile/ (copy the summary into profiles/)
cd /tmp && export TMPDIR=/tmp
N=${1:-71250}
OUT=$GRAFT_REPO_ROOT/gpurun_out/shard_profile
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export CUDASW4_AMD_NO_HANDSHAKE=1 BENCH_PIPELINE=0
rocprofv3 --kernel-trace --stats -d $OUT/raw -o shard -- python3 bench.py --no-sweep --no-secondary --no-cpu-baseline --steps 3 --warmup 1 --workload sprot-like --kernel dpx --db-size $N > $OUT/bench.log 2>&1
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/shard_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("rocprofv3 --kernel-trace --stats: bench.py --workload sprot-like --kernel dpx --db-size (a 1/8 shard), handshake off under the profiler")
print("%-110s %8s %12s %12s %7s" % ("kernel", "calls", "total ms", "avg ms", "%"))
for r in rows[:14]:
    print("%-110s %8s %12.3f %12.4f %7.2f" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
cat $OUT/shard_kernel_stats.txt | cut -c1-170

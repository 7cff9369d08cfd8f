#!/bin/bash
# BASELINE configs 4 and 5 at their own scale on ONE MI355X: a UniRef50-sized synthetic DB
# (6e7 sequences, 2.3e10 residues, 23 GB of chars) through bench.py / the C++ driver:
#   config 4  half2, resident — as one shard and as 8 in-process shards of the device (the shape of an 8-GPU node);
#   config 5  the int32 configuration with the memory limit far below the DB: hybrid residency, and everything streamed.
# Three queries (144, 1000 and 5478 residues); every score of a seeded sample of subjects (incl. the longest) is checked
# against the CPU oracle, the top-10 against the top of all 6e7 scores; the line reports the H2D subject bytes per step.
#   tools/scale_legs.sh [db-size] [tag]    -> gpurun_out/scale_<tag>/*.json
N=${1:-60000000}; TAG=${2:-r04}
OUT=gpurun_out/scale_$TAG; mkdir -p $OUT
Q="--queries 0,9,19 --steps 1 --warmup 1 --no-secondary --no-sweep --cpu-sample-subjects 3000"
run() { name=$1; shift; echo "== $name: $*"; ( time python bench.py --workload uniref50-like --db-size $N $Q "$@" ) > $OUT/$name.json 2> $OUT/$name.err; tail -c 400 $OUT/$name.json; grep real $OUT/$name.err; }
run config4_half2_resident_1shard
run config4_half2_resident_8shards --shards-per-gpu 8
run config5_int32_hybrid_16G --kernel dpxs32 --max-gpu-mem 16G
CUDASW4_AMD_NO_HYBRID=1 run config5_int32_streamed_16G --kernel dpxs32 --max-gpu-mem 16G
run config5_int32_streamed_8shards --kernel dpxs32 --max-gpu-mem 3G --shards-per-gpu 8
python - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
for f in sorted(os.listdir(out)):
    if f.endswith(".json"):
        try:
            d = json.loads([l for l in open(os.path.join(out, f)).read().splitlines() if l.startswith("{")][-1])
            print("%-40s %9.1f GCUPS  %8.1f ms/step  verified=%s  residency=%s  h2d/step=%.2f GB" % (
                f[:-5], d["value"], d["ms_per_step"], d["verified"], d["config"]["residency"], d["config"]["h2d_subject_bytes_per_step"] / 1e9))
        except Exception as e:
            print(f, "no result line:", e)
PY

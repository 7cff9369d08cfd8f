#!/bin/bash
# Peak benchmark of the reference (runpeakbenchmark.sh:26-83) for this build: every kernel type against the
# simulated equal-length DB, all 20 queries, no result output.  Prints the "Total time ... GCUPS" lines.
#   scripts/runpeakbenchmark.sh [query.fasta] [num_subjects]
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
ALIGN=$HERE/cudasw4_amd/lib/align
QUERY=${1:-$HERE/tests/golden/allqueries.fasta}
NUM=${2:-1000000}
COMMON="--query $QUERY --top 0 --verbose --uploadFull --prefetchDBFile --mat blosum62"
run() {  # name, lengths, kernel flags
    local name=$1 lengths=$2; shift 2
    for L in $lengths; do
        echo -n "$name L=$L : "
        $ALIGN $COMMON --pseudodb $NUM $L "$@" | grep "Total time"
    done
}
run half2  "128 256 512 768 1024 2048" --singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float
run dpxs16 "128 256 512 768 1024 2048" --singlePassType DPXs16 --manyPassType_small DPXs16 --manyPassType_large DPXs32 --overflowType DPXs32
run dpxs32 "128 256 512 768 1024"      --singlePassType DPXs32 --manyPassType_small DPXs16 --manyPassType_large DPXs32 --overflowType DPXs32
run float  "128 256 512 768 1024"      --singlePassType Float --manyPassType_small Half2 --manyPassType_large Float

#!/bin/bash
# Real-database benchmarks of the reference (run{sprot,uniref50,trembl}benchmark.sh) for this build.
#   scripts/rundbbenchmark.sh <proteins.fasta[.gz]> <dbprefix> [half2|dpx] [extra align flags...]
# e.g. uniprot_sprot.fasta.gz (benchmarksetup.sh of the reference downloads it; there is no network on the
# build / GPU boxes of this project, so the file has to be provided).
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
FASTA=$1; PREFIX=$2; MODE=${3:-half2}; shift 3 || true
[ -f "${PREFIX}0chars" ] || $HERE/cudasw4_amd/lib/makedb "$FASTA" "$PREFIX"
KFLAGS="--singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float"
[ "$MODE" = dpx ] && KFLAGS="--dpx"
$HERE/cudasw4_amd/lib/align --query $HERE/tests/golden/allqueries.fasta --db "$PREFIX" --top 0 --verbose --uploadFull \
    --prefetchDBFile --mat blosum62 $KFLAGS "$@" | grep -E "Scan time|Total time"

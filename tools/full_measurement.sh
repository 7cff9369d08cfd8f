#!/bin/bash
# Full measurement pass on the GPU box (via gpurun): GPU tests, bench line, peak sweep, Swiss-Prot-like DB, the align
# command line resident and streamed, rocprof stats + PMC.  Results under gpurun_out/final/ (copy what is to be kept into profiles/).
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python tools/peak_sweep.py --json $O/peak_sweep.json > $O/peak_sweep.txt 2>&1
python tools/synth_db_bench.py --config dpx > $O/synth_dpx.txt 2>&1
python tools/synth_db_bench.py --config half2 > $O/synth_half2.txt 2>&1
A=cudasw4_amd/lib/align
$A --query tests/golden/allqueries.fasta --pseudodb 1000000 512 --top 0 --verbose --uploadFull --prefetchDBFile --mat blosum62 --singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float --overflowType Float > $O/align_peak.txt 2>&1
$A --query tests/golden/allqueries.fasta --pseudodb 1000000 512 --top 0 --verbose --maxGpuMem 600M --maxBatchBytes 32M --mat blosum62 --singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float --overflowType Float > $O/align_stream.txt 2>&1
bash tools/collect_profiles.sh r01 > $O/collect.txt 2>&1
cp gpurun_out/profiles_r01/* $O/ 2>/dev/null
tail -3 $O/pytest_gpu.txt; cat $O/bench_line.json; tail -2 $O/align_peak.txt $O/align_stream.txt

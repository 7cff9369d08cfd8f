#!/bin/bash
# Full measurement pass on the GPU box (via gpurun): GPU tests, the bench line (peak + Swiss-Prot-like workload), other kernel
# configurations, hybrid / streamed residency, the multi-rank paths on one GPU, peak and short-query sweeps, the align
# command line.  Results under gpurun_out/final/ (copy what is to be kept into profiles/).  The rocprofv3 passes are
# separate: tools/collect_all_profiles.sh r06 <commit> (tools/final_r06.sh runs both).
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
[ "${SKIP_PYTEST:-0}" = 1 ] || python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
python bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_err.txt
for k in dpxs16 dpxs32 float; do python bench.py --kernel $k --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$k.json; done
CUDASW4_AMD_I32_NATIVE=1 python bench.py --kernel dpxs32 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dpxs32_native.json
python bench.py --max-gpu-mem 600M --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_hybrid.json
CUDASW4_AMD_NO_HYBRID=1 python bench.py --max-gpu-mem 600M --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_streamed.json
python bench.py --kernel dpxs32 --max-gpu-mem 600M --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dpxs32_hybrid.json
python bench.py --workload sprot-like --max-gpu-mem 260M --max-batch-bytes 16M --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_sprot_hybrid.json
CUDASW4_AMD_NO_HYBRID=1 python bench.py --workload sprot-like --max-gpu-mem 260M --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_sprot_streamed.json
python bench.py --workload sprot-like --kernel dpxs32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_sprot_dpxs32.json
python bench.py --db-size 125000 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_125k.json
BENCH_PIPELINE=1 python bench.py --db-size 125000 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_125k_pipeline.json
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_2ranks_1gpu.json
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo python bench.py --gpus 8 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_8ranks_1gpu.json
python tools/peak_sweep.py --json $O/peak_sweep.json > $O/peak_sweep.txt 2>&1
python tools/ragged_query_sweep.py > $O/ragged_query_sweep.txt 2>&1
A=cudasw4_amd/lib/align
$A --query tests/golden/allqueries.fasta --pseudodb 1000000 512 --top 0 --verbose --uploadFull --prefetchDBFile --mat blosum62 --singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float --overflowType Float > $O/align_peak.txt 2>&1
$A --query tests/golden/allqueries.fasta --pseudodb 1000000 512 --top 0 --verbose --maxGpuMem 600M --maxBatchBytes 32M --mat blosum62 --singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float --overflowType Float > $O/align_hybrid.txt 2>&1
tail -3 $O/pytest_gpu.txt; for f in $O/bench_*.json; do python3 - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    v = d["valu_roofline"]
    print(sys.argv[1], d["value"], d["n_gpus"], d["scaling"], d["verified"], d["config"]["residency"], "kernel", v["kernel_gcups"], "valu frac", v["frac"],
          "sprot", d.get("sprot_like", {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -n 2 $O/align_peak.txt; tail -n 2 $O/align_hybrid.txt

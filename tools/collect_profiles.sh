#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the default
# bench.py command, summarised into gpurun_out/profiles_<tag>/ (copy the summaries into profiles/).
#   tools/collect_profiles.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
RAW=gpurun_out/prof_raw_$TAG
mkdir -p $OUT $RAW
export TMPDIR=/tmp
BENCH="bench.py --steps 1 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $RAW/stats -o bench -- python3 $BENCH > $OUT/bench_under_rocprof_stats.log 2>&1
python3 tools/rocprof_summary.py stats $RAW/stats/bench_results.db > $OUT/${TAG}_bench_half2_kernel_stats.txt 2>&1
: > $OUT/${TAG}_bench_half2_pmc.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
    N=$(echo $C | cut -d" " -f1)
    timeout 600 rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc_$N -o bench -- python3 $BENCH > $RAW/pmc_$N.log 2>&1
    python3 tools/rocprof_summary.py pmc $RAW/pmc_$N/bench_results.db "swk::sw_s" >> $OUT/${TAG}_bench_half2_pmc.txt 2>&1
done
python3 tools/rocprof_summary.py traffic $RAW/pmc_FETCH_SIZE/bench_results.db $RAW/pmc_WRITE_SIZE/bench_results.db > $OUT/${TAG}_bench_traffic.json 2>&1
rm -rf $RAW
ls -la $OUT

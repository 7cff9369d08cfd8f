#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of one pass of bench.py's
# workload, summarised into gpurun_out/profiles_<tag>/ (copy the summaries into profiles/), and
# profiles/kernel_counters.json (the PMC constants bench.py reports, tied to the kernel sources' sha) refreshed
# into gpurun_out/profiles_<tag>/kernel_counters.json.
#   tools/collect_profiles.sh r03 [commit] [bench.py workload flags, e.g. --workload sprot-like | --kernel float | --max-gpu-mem 1G]
# PASSES="stats traffic valu lds clock" (default all) selects the rocprofv3 passes; the counters file needs traffic + valu.
# Every invocation ADDS its entries to profiles/kernel_counters.json (one entry per workload / kernel configuration /
# residency, each naming its own source file) as long as the kernel sources are the ones the file was started with.
set -u
TAG=${1:-r02}
COMMIT=${2:-unknown}
if [ $# -ge 2 ]; then shift 2; else shift $#; fi
EXTRA="$*"
NAME=$(echo "bench $EXTRA" | tr -s ' -' '__' | sed 's/_$//')
if [ "${CUDASW4_AMD_I32_NATIVE:-0}" = "1" ]; then NAME=${NAME}_i32native; fi
OUT=gpurun_out/profiles_$TAG
RAW=gpurun_out/prof_raw_$TAG
mkdir -p $OUT $RAW
export TMPDIR=/tmp
# rocprofv3 --pmc serialises kernels: a re-score service (a kernel that polls for a flag set BEHIND the bulk launch) would
# sit there until its spin bound expires (5 s per query).  The profiled passes run without it; the timed bench has it.
export CUDASW4_AMD_RESCORE_SERVICE=0
# ... and without the start handshake: under the profiler's kernel serialisation a stream that waits for a device-side signal
# (hipStreamWaitValue32) never got released (the Swiss-Prot-like passes hung until their timeout); with one kernel at a
# time there is nothing to run beside anyway
export CUDASW4_AMD_NO_HANDSHAKE=1
BENCH="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-secondary --kernel-table $EXTRA"
PASSES=${PASSES:-"stats traffic valu lds clock"}
has() { case " $PASSES " in *" $1 "*) return 0;; *) return 1;; esac; }
if has stats; then
    timeout 900 rocprofv3 --kernel-trace --stats -d $RAW/stats -o bench -- python3 $BENCH > $OUT/${NAME}_under_rocprof_stats.log 2>&1
    python3 tools/rocprof_summary.py stats $RAW/stats/bench_results.db > $OUT/${TAG}_${NAME}_kernel_stats.txt 2>&1
fi
: > $OUT/${TAG}_${NAME}_pmc.txt
pmc_pass() {  # counters of one pass (own run: gpurun wants --pmc alone with --kernel-trace)
    N=$(echo $1 | cut -d" " -f1)
    timeout 900 rocprofv3 --pmc $1 --kernel-trace -d $RAW/pmc_$N -o bench -- python3 $BENCH > $RAW/pmc_$N.log 2>&1
    python3 tools/rocprof_summary.py pmc $RAW/pmc_$N/bench_results.db "swk::sw_s" >> $OUT/${TAG}_${NAME}_pmc.txt 2>&1
}
if has traffic; then pmc_pass "FETCH_SIZE"; pmc_pass "WRITE_SIZE"; fi
if has valu; then pmc_pass "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; fi
if has lds; then pmc_pass "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; fi
if has clock; then pmc_pass "GRBM_GUI_ACTIVE"; fi
if has traffic && has valu; then
python3 tools/rocprof_summary.py counters $RAW/pmc_SQ_INSTS_VALU.log $RAW/pmc_FETCH_SIZE/bench_results.db $RAW/pmc_WRITE_SIZE/bench_results.db \
    $RAW/pmc_SQ_INSTS_VALU/bench_results.db "profiles/${TAG}_${NAME}_pmc.txt (rocprofv3 --pmc SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE)" $COMMIT \
    > $OUT/${TAG}_${NAME}_counters.json 2>&1
[ -n "${KERNEL_COUNTERS_OUT:-}" ] || cp profiles/kernel_counters.json $OUT/kernel_counters.json
fi
rm -rf $RAW
ls -la $OUT
# the counters entry of this configuration must exist when its passes were asked for
if has traffic && has valu; then grep -q '"key"' $OUT/${TAG}_${NAME}_counters.json || exit 1; fi

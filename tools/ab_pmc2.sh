#!/bin/bash
export TMPDIR=/tmp
QI=${1:-0}; L=${2:-128}; TAG=${3:-x}
O=gpurun_out/sq_$TAG; rm -rf $O; mkdir -p $O
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL SQ_INSTS_SENDMSG" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/raw_$i -o r -- python3 tools/one_query.py --query-index $QI --length $L --reps 2 > $O/log_$i.txt 2>&1
  python3 tools/rocprof_summary.py pmc $O/raw_$i/r_results.db "<0" >> $O/pmc.txt 2>&1
done
rm -rf $O/raw_*

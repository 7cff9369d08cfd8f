#!/bin/bash
# A/B counters for one query: tools/ab_pmc.sh <query-index> <L> ; variants via env (CUDASW4_AMD_LIB, CUDASW4_AMD_NO_STREAM)
export TMPDIR=/tmp
QI=${1:-0}; L=${2:-128}; TAG=${3:-x}
O=gpurun_out/sp_$TAG; rm -rf $O; mkdir -p $O
python3 tools/one_query.py --query-index $QI --length $L > $O/run.txt 2>&1
i=0
for C in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/raw_$i -o r -- python3 tools/one_query.py --query-index $QI --length $L --reps 2 > $O/log_$i.txt 2>&1
  python3 tools/rocprof_summary.py pmc $O/raw_$i/r_results.db "<0" >> $O/pmc.txt 2>&1
done
rm -rf $O/raw_*

export TMPDIR=/tmp
for V in r40 r36 p6 p3; do echo lib_$V; CUDASW4_AMD_LIB=$PWD/cudasw4_amd/lib_$V/libcudasw4_amd.so python tools/peak_sweep.py --lengths 512 --kernels half2 2>&1 | grep kernel; done > gpurun_out/ps_wvar.txt 2>&1
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  N=$(echo $C | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$N -o q -- python3 tools/one_query.py --query-index 19 --length 512 --reps 2 > /tmp/pmc_$N.log 2>&1
  python3 tools/rocprof_summary.py pmc /tmp/pmc_$N/q_results.db "swk::sw_s" >> gpurun_out/pmc_wide_q19.txt 2>&1
done

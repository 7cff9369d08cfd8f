#!/bin/bash
# multi-stripe queries: sw_scan_kernel (CUDASW4_AMD_STREAM=1) against the streamed kernels with rounds of different sizes
for K in half2 dpxs16; do
 for L in 128 256 512; do
  for qi in 9 19; do
   echo -n "$K L=$L q=$qi sw_scan_kernel: "
   CUDASW4_AMD_STREAM=1 timeout 300 python tools/one_query.py --query-index $qi --length $L --kernel $K --reps 5 2>&1 | tail -1
   for cols in 700 1536 4096; do
     echo -n "$K L=$L q=$qi stream cols<=$cols: "
     CUDASW4_AMD_STREAM_MULTI_MAX_SUBJECT=100000 CUDASW4_AMD_STREAM_MULTI_COLS=$cols timeout 300 python tools/one_query.py --query-index $qi --length $L --kernel $K --reps 5 2>&1 | tail -1
   done
  done
 done
done

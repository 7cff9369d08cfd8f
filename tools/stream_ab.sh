#!/bin/bash
# A/B of the streamed packed kernels (sw_stream_kernel.hpp) against sw_scan_kernel: single queries on the peak DB at several
# subject lengths (tools/one_query.py), CUDASW4_AMD_STREAM=1 (one batch at a time) against the default.
for L in ${LENGTHS:-128 256 512}; do
  for qi in ${QUERIES:-3 6 9 13 19}; do
    for st in 1 16; do
      echo -n "L=$L q=$qi STREAM=$st: "
      CUDASW4_AMD_STREAM=$st timeout 300 python tools/one_query.py --query-index $qi --length $L --kernel ${KERNEL:-half2} --reps ${REPS:-5} 2>&1 | tail -1
    done
  done
done

#!/bin/bash
# single-stripe queries on EQUAL-length subjects: one slot per round (CUDASW4_AMD_STREAM=1) against streamed rounds
mkdir -p gpurun_out/r6m
for K in dpxs16 half2; do
 for L in 128 256 384 512; do
  for qi in 3 4 5 7; do
   for slots in 1 4 16; do
     echo -n "$K L=$L q=$qi slots=$slots: "
     CUDASW4_AMD_STREAM=$slots timeout 300 python tools/one_query.py --query-index $qi --length $L --db-size 600000 --kernel $K --reps 10 2>&1 | tail -1
   done
  done
 done
done > gpurun_out/r6m/ab3.txt 2>&1
cat gpurun_out/r6m/ab3.txt

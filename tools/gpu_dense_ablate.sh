#!/bin/bash
# dense index (workload c: 10 000 spacers vs 2e8 residues): k_match's time with parts left out (make ablate_dense; wrong results, timing only)
mkdir -p gpurun_out
OUT=gpurun_out/dense_ablate.txt
: > $OUT
for lib in ${LIBS:-libspacepharer_amd.so libspacepharer_amd_ab3.so libspacepharer_amd_abd2.so libspacepharer_amd_abd3.so libspacepharer_amd_abd4.so libspacepharer_amd_abd5.so libspacepharer_amd_abd6.so}; do
  echo "== $lib" >> $OUT
  SP_LIB_NAME=$lib timeout 300 python tools/kmatch_time.py c 8 >> $OUT 2>&1
done
cat $OUT

#!/bin/bash
# A/B on the GPU box: HIP-event time of the matching kernel + crc of the hit list for several builds of the library (make variant NAME=...).
# usage: LIBS="libspacepharer_amd.so libspacepharer_amd_x.so" WL="c" REPS=8 tools/gpu_ab_libs.sh [out-name]
mkdir -p gpurun_out
OUT=gpurun_out/${1:-ab_libs}.txt
: > $OUT
for lib in ${LIBS:-libspacepharer_amd.so}; do
  echo "== $lib" >> $OUT
  SP_LIB_NAME=$lib timeout 300 python tools/kmatch_time.py ${WL:-c} ${REPS:-8} >> $OUT 2>&1
done
cat $OUT

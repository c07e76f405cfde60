#!/bin/bash
# k_match_dense (round 6) against k_match<512,false,false> on workload c (10 000 spacers vs 2e8 residues): time + crc of the hit list, then the dense parity tests
mkdir -p gpurun_out
OUT=gpurun_out/${1:-dense2}.txt
: > $OUT
for v in 0 1 0 1; do
  echo "== SP_MATCH_DENSE2=$v ${LIBNAME:-}" >> $OUT
  SP_MATCH_DENSE2=$v SP_LIB_NAME=${LIBNAME:-libspacepharer_amd.so} timeout 300 python tools/kmatch_time.py c ${REPS:-10} >> $OUT 2>&1
done
if [ -z "$NOTEST" ]; then timeout 900 python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -5 >> $OUT; fi
cat $OUT

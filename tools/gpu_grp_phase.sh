#!/bin/bash
# phase shares of k_match_grp / k_match (SP_PHASE_TIMERS build) on workload a
mkdir -p gpurun_out
OUT=gpurun_out/grp_phase.txt
: > $OUT
for g in 0 1; do
  echo "== SP_MATCH_GROUP=$g" >> $OUT
  SP_LIB_NAME=libspacepharer_amd_phase.so SP_MATCH_GROUP=$g timeout 300 python tools/kmatch_time.py a 6 2>&1 | tail -4 >> $OUT
done
cat $OUT

#!/bin/bash
# occupancy sensitivity of the matching kernels: blocks per CU 2 / 3 / 4 (SP_MATCH_PERCU), grouped or not
mkdir -p gpurun_out
OUT=gpurun_out/grp_occ.txt
: > $OUT
for g in 1 0; do for p in 2 3 4; do
  echo "== SP_MATCH_GROUP=$g SP_MATCH_PERCU=$p" >> $OUT
  SP_MATCH_GROUP=$g SP_MATCH_PERCU=$p SP_MATCH_WEIGH_EXACT=${EXACT:-0} timeout 300 python tools/kmatch_time.py a 10 2>&1 | tail -1 >> $OUT
done; done
cat $OUT

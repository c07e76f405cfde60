#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/grp_tail.txt
: > $OUT
for t in "0 0" "1 3" "2 4" "2 6" "4 8"; do set -- $t
  for p in 3 4; do
  echo "== TAIL1=$1 TAIL2=$2 PERCU=$p" >> $OUT
  SP_MATCH_TAIL1=$1 SP_MATCH_TAIL2=$2 SP_MATCH_PERCU=$p timeout 300 python tools/kmatch_time.py a 10 2>&1 | tail -1 | cut -c1-150 >> $OUT
  done
done
cat $OUT

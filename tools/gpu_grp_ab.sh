#!/bin/bash
# k_match_grp vs k_match on the GPU box: golden parity of the prefilter tests with the grouped kernel, then HIP-event times + crc of both.
mkdir -p gpurun_out
OUT=gpurun_out/grp_ab.txt
: > $OUT
timeout 600 python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu 2>&1 | tail -15 >> $OUT
for g in 0 1; do
  echo "== SP_MATCH_GROUP=$g" >> $OUT
  SP_MATCH_GROUP=$g timeout 300 python tools/kmatch_time.py a b 12 >> $OUT 2>&1
done
echo "== exact weights" >> $OUT
for g in 0 1; do
  SP_MATCH_GROUP=$g SP_MATCH_WEIGH_EXACT=1 timeout 300 python tools/kmatch_time.py a 12 >> $OUT 2>&1
done
cat $OUT

#!/bin/bash
# the bench's step against the matching kernel's blocks per CU while four contexts share the GPU (--match-blocks), grouped kernel on / off
mkdir -p gpurun_out
OUT=gpurun_out/blocks_sweep.txt
: > $OUT
for g in 1 0; do for b in 2 3 4; do
  echo "== SP_MATCH_GROUP=$g --match-blocks $b" >> $OUT
  SP_MATCH_GROUP=$g timeout 600 python bench.py --cpu-sample 0 --no-module-e2e --no-dense --match-blocks $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); sm = d['step_modes']
        print('value %.2f M  ms_per_step %.4f  regions %s  host-driven x4 %.2f M (k_match %.3f ms/launch), device-driven x2 %.2f M' % (d['value'] / 1e6, d['ms_per_step'], [round(x / 1e6, 2) for x in d['regions']['spacer_queries_per_s']], sm['host-driven x4']['spacer_queries_per_s'] / 1e6, sm['host-driven x4']['k_match_ms_per_launch'], sm['device-driven x2']['spacer_queries_per_s'] / 1e6))
" >> $OUT
done; done
cat $OUT

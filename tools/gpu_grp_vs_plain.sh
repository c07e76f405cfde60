#!/bin/bash
# four host-driven contexts: the grouped kernel (default) against k_match<256,false,true>, by blocks per CU
mkdir -p gpurun_out
for g in 1 0; do for mb in 0 2 3 -1; do
  echo "== SP_MATCH_GROUP=$g --match-blocks $mb"
  SP_MATCH_GROUP=$g timeout 600 python bench.py --cpu-sample 0 --no-module-e2e --no-dense --regions 3 --match-blocks $mb 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); sm = d['step_modes']
        print('value %.2f M  ms_per_step %.4f  regions %s  host x4 %.2f M  device x4 %.2f M' % (d['value'] / 1e6, d['ms_per_step'], [round(x / 1e6, 2) for x in d['regions']['spacer_queries_per_s']], sm['host-driven x4']['spacer_queries_per_s'] / 1e6, sm['device-driven x4']['spacer_queries_per_s'] / 1e6))
"
done; done 2>&1 | tee gpurun_out/grp_vs_plain.txt

#!/bin/bash
# host-driven step: contexts per GPU x matching-kernel blocks per CU (k_match_grp)
mkdir -p gpurun_out
OUT=gpurun_out/blocks_sweep2.txt
: > $OUT
for b in 1 2 3; do for p in 3 4 5 6; do
  echo -n "match-blocks $b, $p contexts: " >> $OUT
  timeout 600 python bench.py --cpu-sample 0 --no-module-e2e --no-dense --match-blocks $b --pipeline $p --host-driven --regions 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%.2f M  ms_per_step %.4f  regions %s' % (d['value'] / 1e6, d['ms_per_step'], [round(x / 1e6, 2) for x in d['regions']['spacer_queries_per_s']]))
" >> $OUT
done; done
cat $OUT

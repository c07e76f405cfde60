#!/bin/bash
# host-driven x4 step under tuning knobs of the matching kernel (k_match_grp, two blocks per CU)
mkdir -p gpurun_out
OUT=gpurun_out/tune_sweep.txt
: > $OUT
for t in "" "--tune match_weigh_exact=1" "--tune match_chunk=2" "--tune match_chunk=8" "--tune match_static=0" "--tune match_static=8" "--tune match_heavy_threshold=0" "--tune match_heavy_threshold=8000"; do
  echo -n "[$t] " >> $OUT
  timeout 600 python bench.py --cpu-sample 0 --no-module-e2e --no-dense --pipeline 4 --host-driven --regions 5 $t 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%.2f M  ms_per_step %.4f  regions %s' % (d['value'] / 1e6, d['ms_per_step'], [round(x / 1e6, 2) for x in d['regions']['spacer_queries_per_s']]))
" >> $OUT
done
cat $OUT

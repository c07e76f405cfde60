#!/bin/bash
# the bench's step with the grouped matching kernel on / off, match blocks per CU 2 / 3 / 4
mkdir -p gpurun_out
OUT=gpurun_out/grp_bench.txt
: > $OUT
for g in 1 0; do for b in ${BLOCKS:-2 3 4}; do
  echo "== SP_MATCH_GROUP=$g --match-blocks $b" >> $OUT
  SP_MATCH_GROUP=$g timeout 600 python bench.py --cpu-sample 0 --no-module-e2e --match-blocks $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.2f M  ms_per_step %.4f  regions %s  modes %s' % (d['value'] / 1e6, d['ms_per_step'], [round(r / 1e6, 2) for r in d.get('regions', [])] if isinstance(d.get('regions'), list) else d.get('regions'), json.dumps(d.get('step_modes'))[:300]))
" >> $OUT
done; done
cat $OUT

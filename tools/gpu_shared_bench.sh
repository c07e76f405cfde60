#!/bin/bash
# the bench's step with one shared staged target set (default) and with a copy per context (--unshared)
mkdir -p gpurun_out
OUT=gpurun_out/shared_bench.txt
: > $OUT
for rep in 1 2; do for u in "" "--unshared"; do
  echo "== bench.py $u (run $rep)" >> $OUT
  timeout 600 python bench.py --cpu-sample 0 --no-module-e2e $u 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('value %.2f M  ms_per_step %.4f  regions %s  k_match/launch %s frac %s' % (d['value'] / 1e6, d['ms_per_step'], [round(x / 1e6, 2) for x in d['regions']['spacer_queries_per_s']], r.get('kernel_ms'), r.get('frac')))
" >> $OUT
done; done
cat $OUT

# GPU box: the bench step, device-driven (sp_ctx_search_begin / _end) vs host-driven, k_match blocks per CU, contexts per GPU
for mode in "" "--host-driven"; do for mb in 2 3 4; do
  timeout 300 python bench.py --no-module-e2e --cpu-sample 0 --regions 3 --match-blocks $mb $mode > /tmp/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('mode=%-14s blocks=$mb  %.2f M/s  %.4f ms/step  frac %.3f' % ('$mode' or 'device-driven', d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))"
done; done
for p in 1 3; do for mode in "" "--host-driven"; do timeout 300 python bench.py --no-module-e2e --cpu-sample 0 --regions 3 --pipeline $p $mode > /tmp/b.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/b.json')); print('pipeline $p %-14s: %.2f M/s  %.4f ms/step  frac %.3f' % ('$mode' or 'device-driven', d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))"; done; done

for p in 2 3 4; do for f in 1 2; do timeout 300 python bench.py --no-module-e2e --cpu-sample 0 --regions 3 --pipeline $p --in-flight $f > /tmp/b.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/b.json')); print('pipeline $p in-flight $f device-driven: %.2f M/s  %.4f ms/step  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))"; done; done
for p in 3 4; do for mb in 2 3 4; do timeout 300 python bench.py --no-module-e2e --cpu-sample 0 --regions 3 --pipeline $p --host-driven --match-blocks $mb > /tmp/b.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/b.json')); print('pipeline $p host-driven blocks $mb: %.2f M/s  %.4f ms/step  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))"; done; done

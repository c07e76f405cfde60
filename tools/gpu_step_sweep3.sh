# GPU box: does the number of hardware queues the HIP runtime multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4) bound the
# device-driven step with three contexts (6 streams)?
cd $GRAFT_REPO_ROOT
for q in 4 8 16; do for p in 2 3; do for mode in "" "--host-driven"; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-module-e2e --cpu-sample 0 --regions 3 --pipeline $p $mode > /tmp/b.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/b.json')); print('hw queues $q pipeline $p ${mode:-device-driven}: %.2f M/s  %.4f ms/step  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))"
done; done; done

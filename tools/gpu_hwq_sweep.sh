#!/bin/bash
# GPU box: the headline step (driver's command line) and the dense step under GPU_MAX_HW_QUEUES (HIP's default is 4 hardware queues per process: nine streams -- four contexts x (search, copy) + the
# null stream -- share them, and a kernel queued behind another context's 1 ms copy waits for it) -> gpurun_out/hwq_sweep.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for q in ${QS:-4 8 12 16}; do
  export GPU_MAX_HW_QUEUES=$q
  timeout 300 python bench.py --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps 20 --warmup 5 > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('GPU_MAX_HW_QUEUES=%s headline %6.3f M/s  %.4f ms/step  modes %s' % (sys.argv[1], d['value']/1e6, d['ms_per_step'], {k: round(v.get('value', 0)/1e6, 2) if isinstance(v, dict) else v for k, v in (d.get('step_modes') or {}).items()}))" $q
  timeout 300 python bench.py --targets 1000000 --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps 12 --regions 3 --warmup 2 --pipeline 2 --host-driven > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('GPU_MAX_HW_QUEUES=%s dense    %6.3f M/s  %.4f ms/step' % (sys.argv[1], d['value']/1e6, d['ms_per_step']))" $q
done 2>&1 | tee gpurun_out/hwq_sweep.txt

#!/bin/bash
# GPU box: the headline step (10 000 spacers vs 1e7 residues, host-driven) at the driver's K = 20 / W = 5 under a few scheduling variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 300 python bench.py --cpu-sample 0 --no-module-e2e --no-dense --steps 20 --warmup 5 --regions 7 --host-driven "$@" > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('%-40s %6.2f M/s  %.4f ms/step  regions %s' % (sys.argv[1], d['value']/1e6, d['ms_per_step'], ' '.join('%.1f' % (x/1e6) for x in d['regions']['spacer_queries_per_s'])))" "$*"; }
for args in ${MODES:-"--pipeline=4" "--pipeline=4,--stagger-us=100" "--pipeline=4,--stagger-us=200" "--pipeline=4,--stagger-us=400" "--pipeline=3" "--pipeline=5" "--pipeline=4,--steps=100"}; do run ${args//,/ }; done 2>&1 | tee gpurun_out/headline_sweep.txt

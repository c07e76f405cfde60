#!/bin/bash
# GPU box: the dense step (10 000 spacers vs 2e8 residues) under contexts x matching-kernel blocks per CU  -> gpurun_out/dense_step.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 300 python bench.py --targets 1000000 --cpu-sample 0 --no-module-e2e --no-dense --steps 10 --regions 3 --warmup 2 "$@" > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('%-52s %6.3f M/s  %.3f ms/step  regions %s  kernels %s' % (sys.argv[1], d['value']/1e6, d['ms_per_step'], d.get('regions'), {k: v for k, v in d['kernels_ms_per_step'].items() if v > 0.03}))" "$*"; }
for args in ${MODES:-"--pipeline=2,--host-driven" "--pipeline=2,--host-driven,--match-blocks=2" "--pipeline=3,--host-driven,--match-blocks=2" "--pipeline=4,--host-driven,--match-blocks=2" "--pipeline=2" "--pipeline=2,--match-blocks=2" "--pipeline=3,--match-blocks=2"}; do run ${args//,/ }; done 2>&1 | tee gpurun_out/dense_step.txt

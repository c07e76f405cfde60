#!/bin/bash
# GPU box: the dense step (10 000 spacers vs 2e8 residues) with and without the co-scheduling of round 6 (SP_CO_SCHEDULE: the kernels behind the forward pass wait for the
# other context's matcher; matcher at two blocks per CU) -> gpurun_out/dense_cosched.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 300 python bench.py --targets 1000000 --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps 12 --regions 3 --warmup 2 "$@" > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; python -c "
import json,sys,os; d=json.load(open('/tmp/b.json')); print('SP_CO_SCHEDULE=%s %-44s %6.3f M/s  %.3f ms/step  kernels %s' % (os.environ.get('SP_CO_SCHEDULE','default'), sys.argv[1], d['value']/1e6, d['ms_per_step'], {k: v for k, v in d['kernels_ms_per_step'].items() if v > 0.03}))" "$*"; }
for cs in ${CS:-0 1}; do export SP_CO_SCHEDULE=$cs
for args in ${MODES:-"--pipeline=2,--host-driven" "--pipeline=3,--host-driven" "--pipeline=2"}; do run ${args//,/ }; done; done 2>&1 | tee gpurun_out/${1:-dense_cosched}.txt

#!/bin/bash
# GPU box: forward Smith-Waterman pass (and the rest of the step's kernels) on the dense step for several builds: LIBS="libspacepharer_amd.so libspacepharer_amd_x.so"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in ${LIBS:-libspacepharer_amd.so}; do
  for rep in 1 2; do
  SP_LIB_NAME=$lib timeout 300 python bench.py --targets 1000000 --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps 8 --regions 2 --warmup 2 --pipeline 2 --host-driven > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('%-34s dense %.3f ms/step  kernels %s' % (sys.argv[1], d['ms_per_step'], {k: v for k, v in d['kernels_ms_per_step'].items() if v > 0.03}))" $lib
  done
done 2>&1 | tee gpurun_out/${1:-sw_ab}.txt

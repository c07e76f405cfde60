# GPU box: the dense step (10 000 spacers vs 2e8 residues) in every step mode: contexts x host- / device-driven (MODES="flags;flags;..." for other sets) -> gpurun_out/${1:-dense_modes}.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
IFS=';' read -ra MODE_LIST <<< "${MODES:---pipeline 1 --host-driven;--pipeline 1 --in-flight 1;--pipeline 1 --in-flight 2;--pipeline 2 --host-driven;--pipeline 2 --in-flight 1;--pipeline 2 --in-flight 2}"
for mode in "${MODE_LIST[@]}"; do
  timeout 300 python bench.py --targets 1000000 --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps 8 --regions 3 --warmup 2 $mode ${BENCH_FLAGS} > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('%-34s dense %.3f ms/step  kernels %s' % (sys.argv[1], d['ms_per_step'], {k: round(v, 3) for k, v in d['kernels_ms_per_step'].items() if v > 0.03}))" "$mode"
done 2>&1 | tee gpurun_out/${1:-dense_modes}.txt

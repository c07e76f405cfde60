# GPU box: the headline step (10 000 spacers vs 1e7 residues) by step mode and number of contexts -> gpurun_out/${1:-step_modes2}.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
IFS=';' read -ra MODE_LIST <<< "${MODES:---pipeline 4 --host-driven;--pipeline 2 --in-flight 2;--pipeline 3 --in-flight 2;--pipeline 4 --in-flight 2;--pipeline 4 --in-flight 1;--pipeline 6 --in-flight 1}"
for mode in "${MODE_LIST[@]}"; do
  timeout 300 python bench.py --cpu-sample 0 --no-module-e2e --no-dense --no-probe-proxy --steps ${STEPS:-60} --regions 3 --warmup 5 $mode ${BENCH_FLAGS} > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('%-34s %.2f M/s  %.4f ms/step' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step']))" "$mode"
done 2>&1 | tee gpurun_out/${1:-step_modes2}.txt

# GPU box: the bench step under its launch modes -- device-driven / host-driven x 2, 3, 4 contexts -- and host-driven x4 against the number of hardware
# queues the HIP runtime multiplexes the streams onto (GPU_MAX_HW_QUEUES, default 4).  -> profiles/r03_step_modes.txt
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-module-e2e --cpu-sample 0 --regions 3 "$@" > /tmp/b.json 2>/dev/null; python -c "
import json,sys; d=json.load(open('/tmp/b.json')); print('%-44s %6.2f M/s  %.4f ms/step  k_match %.3f ms per launch in the region' % (sys.argv[1], d['value']/1e6, d['ms_per_step'], d['roofline']['algorithmic_bytes_per_launch']/d['roofline']['achieved']/1e6))" "$*"; }
for p in 1 2 3 4; do run --pipeline $p; done
for p in 1 2 3 4 6; do run --pipeline $p --host-driven; done
for q in 2 3 4 6 8; do echo -n "GPU_MAX_HW_QUEUES=$q  "; GPU_MAX_HW_QUEUES=$q run --pipeline 4 --host-driven; done
echo -n "exact weighing           "; run --pipeline 4 --host-driven --tune match_weigh_exact=1
echo -n "no heavy-query list      "; run --pipeline 4 --host-driven --tune match_heavy_threshold=0
echo "default command line (both modes, the better one is the value):"
timeout 600 python bench.py --no-module-e2e --cpu-sample 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['step_modes'])); print('value %.2f M/s  %.4f ms/step' % (d['value']/1e6, d['ms_per_step']))"

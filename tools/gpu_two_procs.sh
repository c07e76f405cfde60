# GPU box experiment: two independent bench processes on ONE GPU at the same time (do the latency-bound matcher of one and
# the VALU-bound aligner of the other share the machine?).  Prints the two values and their sum.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for percu in 3 2; do
  export SP_MATCH_PERCU=$percu
  timeout 600 python bench.py --steps 300 --warmup 5 --cpu-sample 0 > gpurun_out/two_a.json 2>/dev/null &
  PA=$!
  timeout 600 python bench.py --steps 300 --warmup 5 --cpu-sample 0 > gpurun_out/two_b.json 2>/dev/null &
  PB=$!
  wait $PA; wait $PB
  python - <<PY
import json
a=json.loads([l for l in open('gpurun_out/two_a.json') if l.startswith('{')][-1]); b=json.loads([l for l in open('gpurun_out/two_b.json') if l.startswith('{')][-1])
print("blocks/CU $percu: two processes %.0f + %.0f = %.0f spacer-queries/s (%.3f / %.3f ms per step)" % (a['value'], b['value'], a['value']+b['value'], a['ms_per_step'], b['ms_per_step']))
PY
done
unset SP_MATCH_PERCU
timeout 600 python bench.py --steps 300 --warmup 5 --cpu-sample 0 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one process: %.0f spacer-queries/s (%.3f ms per step)' % (d['value'], d['ms_per_step']))"

# GPU box: the size sweep quoted in DESIGN.md §7 (one line per workload)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/sizes.txt
one() {
  timeout 600 python bench.py --steps 40 --warmup 3 --regions 3 --cpu-sample 0 --no-module-e2e "$@" 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', '|', round(d['value']), d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/sizes.txt
}
one --spacers 20000
one --spacers 40000
one --spacers 100000
one --stage prefilter
one --targets 1000000 --target-len 200 --pipeline 2      # (four contexts of a dense index: 1.4 M against 2.2 M with two)

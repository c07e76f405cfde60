# GPU box: A/B of two builds of the library (SP_LIB_NAME) on the bench step.  usage: gpu_ab.sh <libA> <libB> [bench flags...]
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for rep in 1 2; do for lib in $A $B; do for mode in "--pipeline 2" "--pipeline 3 --host-driven" "--pipeline 1"; do
  SP_LIB_NAME=$lib timeout 300 python bench.py --no-module-e2e --cpu-sample 0 --regions 3 $mode "$@" > /tmp/b.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/b.json')); print('%-28s %-26s: %.2f M/s  %.4f ms/step  frac %.3f  k_match %.4f ms' % ('$lib', '$mode', d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['kernels_ms_per_step']['match']))"
done; done; done

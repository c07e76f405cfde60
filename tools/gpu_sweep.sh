# GPU box: A/B of k_match scheduling knobs on one box
cd $GRAFT_REPO_ROOT
for cfg in "4000 4 4" "4000 2 4" "4000 0 4" "4000 0 2" "3000 2 4" "3000 0 4" "2500 1 4" "6000 2 4" "0 4 4"; do
  set -- $cfg
  export SP_MATCH_HEAVY=$1 SP_MATCH_STATIC=$2 SP_MATCH_DCHUNK=$3
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --stage prefilter 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('heavy>=$1 static=$2 dchunk=$3', '|', round(d['value']), d['ms_per_step'], d['kernels_ms_per_step']['match'], d['kernels_ms_per_step']['prep'])"
done

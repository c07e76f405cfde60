# GPU box: A/B of k_match scheduling knobs on one box (SP_MATCH_HEAVY / SP_MATCH_STATIC / SP_MATCH_DCHUNK)
cd $GRAFT_REPO_ROOT
for cfg in "4000 4 4" "3000 4 4" "2000 4 4" "6000 4 4" "3000 6 4" "3000 5 4" "3000 4 8" "2500 6 8" "4000 4 4"; do
  set -- $cfg
  export SP_MATCH_HEAVY=$1 SP_MATCH_STATIC=$2 SP_MATCH_DCHUNK=$3
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --stage prefilter 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('heavy>=$1 static=$2/8 dchunk=$3', '|', round(d['value']), d['ms_per_step'], d['roofline']['achieved'] and round(1182.99/d['roofline']['achieved'],4))"
done

# GPU box: A/B of pipeline depth x k_match blocks per CU on one box (SP_MATCH_PERCU overrides the bench's choice)
cd $GRAFT_REPO_ROOT
for cfg in "3 1" "4 1" "3 2" "2 2" "2 1"; do
  set -- $cfg
  export SP_MATCH_PERCU=$2
  timeout 300 python bench.py --steps 120 --warmup 3 --cpu-sample 0 --pipeline $1 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipeline=$1 blocks/CU=$2', '|', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
done

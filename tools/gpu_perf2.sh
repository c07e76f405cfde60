cd $GRAFT_REPO_ROOT
echo "== 100k spacers vs 1e7"; timeout 600 python bench.py --spacers 100000 --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['config']['pairs_aligned_per_step'], d['config']['index_build_s'])"
echo "== 10k spacers vs 2e8"; timeout 900 python bench.py --targets 1000000 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -3 | cut -c1-900

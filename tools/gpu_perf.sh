cd $GRAFT_REPO_ROOT
make -s -C oracle liboracle.so
timeout 900 python -m pytest tests -x -q -m gpu --timeout 400 2>&1 | tail -4
for v in "SP_ABLATE=0"; do echo "== $v"; env $v timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['frac'])"; done

# GPU box: all -m gpu tests, the smoke entry, one short bench line
cd $GRAFT_REPO_ROOT
make -s -C oracle liboracle.so
timeout 900 python -m pytest tests -x -q -m gpu --timeout 400 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 60 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['frac'])"

cd $GRAFT_REPO_ROOT
nproc; free -g | head -2
timeout 900 python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step']); print(d['cpu_baseline']); print(d['roofline'])"

cd $GRAFT_REPO_ROOT
make -s -C oracle liboracle.so
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q --timeout 900 2>&1 | tail -5
timeout 600 python bench.py --shard targets --cpu-sample 0 --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"

cd $GRAFT_REPO_ROOT
make -s -C oracle liboracle.so
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q --timeout 900 2>&1 | tail -25
timeout 600 python bench.py --shard targets --cpu-sample 0 2>&1 | tail -1 | cut -c1-1500

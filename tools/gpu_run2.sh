cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout 400 python -m pytest tests/test_gpu_align.py -x -q -m gpu --timeout 120 2>&1 | tail -30

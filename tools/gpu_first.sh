cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout 600 python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu --timeout 300 2>&1 | tail -40

cd $GRAFT_REPO_ROOT
make -s -C oracle liboracle.so
timeout 900 python -m pytest tests/test_gpu_prefilter.py -x -q --timeout 400 2>&1 | tail -15

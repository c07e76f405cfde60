cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout 900 python -m pytest tests -x -q -m gpu --timeout 400 2>&1 | tail -6

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout 600 python -m pytest tests -x -q -m gpu --timeout 200 2>&1 | tail -30
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3

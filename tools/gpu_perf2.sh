cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q --timeout 400 2>&1 | tail -3
timeout 600 python tools/module_e2e.py 10000 2>&1 | tail -2
timeout 600 python tools/module_e2e.py 100000 2>&1 | tail -2

cd $GRAFT_REPO_ROOT
timeout 300 python tools/exp_order.py 2>&1 | tail -6

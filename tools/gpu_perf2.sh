cd $GRAFT_REPO_ROOT
timeout 300 python tools/host_prof.py 2>&1 | tail -6

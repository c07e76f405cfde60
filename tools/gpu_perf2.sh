cd $GRAFT_REPO_ROOT
SP_DEBUG=1 timeout 300 python tools/host_prof.py 2>&1 | grep 'align_hits:\|^prefilter' | tail -4

cd $GRAFT_REPO_ROOT
SP_DEBUG=1 timeout 300 python tools/host_prof.py > /tmp/hp.log 2>&1; grep "align_hits:" /tmp/hp.log | tail -2; grep "^prefilter" /tmp/hp.log | tail -3

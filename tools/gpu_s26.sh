cd $GRAFT_REPO_ROOT
timeout 900 python tools/soak.py 1000000 2>&1 | tail -12

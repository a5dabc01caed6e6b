cd $GRAFT_REPO_ROOT
python tools/host_split.py 20 2>&1 | tail -4
python tools/host_split.py 4 2>&1 | tail -4

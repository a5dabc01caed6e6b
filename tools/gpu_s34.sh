cd $GRAFT_REPO_ROOT
timeout 600 python tools/replay_bench.py 2>&1 | tail -4

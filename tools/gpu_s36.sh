cd $GRAFT_REPO_ROOT
timeout 600 python tools/closed_loop_bench.py 2>&1 | tail -9

cd $GRAFT_REPO_ROOT
timeout 900 python tools/fused_stress.py 4000 2>&1 | tail -8

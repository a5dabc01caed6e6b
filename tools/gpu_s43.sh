cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "bench" 2>&1 | tail -3

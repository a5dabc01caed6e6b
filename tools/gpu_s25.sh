cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_polygon.py -x -q -k box_fast 2>&1 | tail -15

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "unchanged_reference_loop" 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_task.py -x -q 2>&1 | tail -5

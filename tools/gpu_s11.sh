O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== polygon tests"; timeout 600 python -m pytest tests/test_gpu_polygon.py -x -q 2>&1 | tail -30 | tee $O/s11_polygon.log
echo "== motion"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "motion" 2>&1 | tail -30 | tee $O/s11_motion.log
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/s11_gpu_tests.log

O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== targeted"; timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "bench_launches or serialised or launch_modes or reserving or replayed or feedforward or form_selection" 2>&1 | tail -30 | tee $O/r04f_targeted.txt
echo "== full"; timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fused.py::test_bench_launches_its_own_ranks 2>&1 | tail -40 > $O/r04f_gpu_tests.txt; cat $O/r04f_gpu_tests.txt

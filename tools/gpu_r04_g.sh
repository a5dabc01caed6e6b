O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== full tests"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r04g_gpu_tests.txt; cat $O/r04g_gpu_tests.txt
echo "== stress"; timeout 1500 python tools/fused_stress.py 20000 2>&1 | grep -v amdgpu.ids | tee $O/r04g_fused_stress.txt
echo "== driver line"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04g_driver_bench_line.json 2> /dev/null; cut -c1-400 $O/r04g_driver_bench_line.json

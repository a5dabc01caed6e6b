O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== probe"; timeout 900 python tools/r04_probe.py 20 300 > $O/r04b_probe.txt 2>&1; grep -v amdgpu.ids $O/r04b_probe.txt | tail -25
echo "== tests"; timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/r04b_gpu_tests.txt; cat $O/r04b_gpu_tests.txt

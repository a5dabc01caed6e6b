O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in drop keep; do
echo "== slow mode probe ($mode)"; timeout 600 python tools/slow_mode_probe.py 16 $mode 2>&1 | grep -v amdgpu.ids | tee $O/r04e_slow_$mode.txt
done
echo "== probe"; timeout 900 python tools/r04_probe.py 20 300 > $O/r04e_probe.txt 2>&1; grep -v amdgpu.ids $O/r04e_probe.txt | tail -25
echo "== tests"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/r04e_gpu_tests.txt; cat $O/r04e_gpu_tests.txt

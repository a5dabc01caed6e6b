O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in drop keep; do
echo "== slow mode probe ($mode)"; timeout 600 python tools/slow_mode_probe.py 24 $mode 2>&1 | grep -v amdgpu.ids | tee $O/r04c_slow_$mode.txt
done
echo "== RIAB_GATE=always"; RIAB_GATE=always timeout 600 python tools/slow_mode_probe.py 16 drop 2>&1 | grep -v amdgpu.ids | tee $O/r04c_slow_always.txt

cd $GRAFT_REPO_ROOT
echo "== fused tests"; timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -4
echo "== whole suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== standalone"; timeout 120 python tools/traj_probe.py 2>&1 | grep "T="
RIAB_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/libt4_31.so timeout 120 python tools/traj_probe.py 2>&1 | grep "T="
timeout 120 python tools/traj_probe.py maze 2>&1 | grep "T="
echo "== traj profile"; RIAB_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/libt4_prof.so timeout 120 python tools/traj_profile.py 2>&1 | tail -4
echo "== host split"; timeout 300 python tools/host_split.py 20 2>&1 | grep -v Warn | tail -4
echo "== probe"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0"

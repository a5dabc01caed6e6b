cd $GRAFT_REPO_ROOT
echo "== fused tests"; timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -5
echo "== standalone trajectory kernel"
timeout 120 python tools/traj_probe.py 2>&1 | grep "T="
RIAB_TRAJ2=1 timeout 120 python tools/traj_probe.py 2>&1 | grep "T="
RIAB_NO_PC=1 timeout 120 python tools/traj_probe.py 2>&1 | grep "T="
for b in 1 2 4 8 16 3 31; do RIAB_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/libt4_$b.so timeout 120 python tools/traj_probe.py 2>&1 | grep "T="; done
echo "== maze"; timeout 120 python tools/traj_probe.py maze 2>&1 | grep "T="
RIAB_TRAJ2=1 timeout 120 python tools/traj_probe.py maze 2>&1 | grep "T="

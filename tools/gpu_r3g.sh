cd $GRAFT_REPO_ROOT
echo "== prelude"; timeout 300 python tools/prelude_probe.py 2>&1 | grep " us"
echo "== host split"; timeout 300 python tools/host_split.py 20 2>&1 | grep -v Warn | tail -5
echo "== traj profile"; RIAB_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/libt4_prof.so timeout 120 python tools/traj_profile.py 2>&1 | tail -5
echo "== probe"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0"

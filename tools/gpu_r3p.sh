cd $GRAFT_REPO_ROOT
echo "== probe default"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=False\|spin=0 timing=True"
echo "== probe launcher"; RIAB_LAUNCHER=1 timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=False\|spin=0 timing=True"
echo "== fused tests with the launcher"; RIAB_LAUNCHER=1 timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
echo "== probe default again"; timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=False"
echo "== probe launcher again"; RIAB_LAUNCHER=1 timeout 300 python tools/k20_probe.py 20 200 2>&1 | grep "spin=0 timing=False"

cd $GRAFT_REPO_ROOT
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== example"; timeout 600 python examples/simple_example.py 2>&1 | tail -8

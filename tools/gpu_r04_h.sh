O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== randomised sweeps, 60 worlds each"; RIAB_TEST_WORLDS=60 timeout 2400 python -m pytest tests -m gpu -q -k "randomised" 2>&1 | tail -6 | tee $O/r04h_randomised.txt

cd $GRAFT_REPO_ROOT
python tools/host_split.py 20 2>&1 | tail -4
python -c "
import cProfile, pstats, sys, torch
sys.path.insert(0,'.')
import bench, ratinabox_amd as riab
env, ag, pops = bench.build_world(riab, bench.CONFIGS['cfg2'], 0, 64)
ag.preallocate_history(20*3000)
for _ in range(5): ag.simulate(20)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): ag.simulate(20)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
" 2>&1 | tail -32

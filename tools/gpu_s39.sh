cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -45
import cProfile, pstats, sys, os
import numpy as np, torch
sys.path.insert(0, ".")
import ratinabox_amd as riab
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, get_goal_vector
B, n = 4096, 1024
np.random.seed(0)
env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=2),
                             episode_terminate_delay=0.05, teleport_on_reset=True)
ag = riab.Agent(env, {"n_agents": B, "dt": 0.01})
pcs = riab.PlaceCells(ag, {"n": n, "save_spikes": False})
env.add_agents(ag); env.reset(); ag.preallocate_history(1200)
def step():
    v = get_goal_vector(ag)
    obs, reward, terminal, truncated, info = env.step(11 * ag.speed_mean * v / v.norm(dim=1, keepdim=True).clamp_min(1e-9))
    env.reset(mask=terminal)
    pcs.update()
for _ in range(40): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
PY

"""Long production-mode runs of the motion kernel: every state value stays finite, agents stay inside the box, the
bounce loop never saturates.  python tools/soak.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab  # noqa: E402

MAZE = [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
cases = {
    "open dt=10ms": ({}, {"dt": 0.01}, None),
    "maze dt=10ms": ({"walls": MAZE}, {"dt": 0.01}, None),
    "maze dt=50ms fast": ({"walls": MAZE}, {"dt": 0.05, "speed_mean": 0.3, "speed_std": 0.2}, None),
    "periodic + wall": ({"boundary_conditions": "periodic", "walls": [[[0.5, 0.2], [0.5, 0.8]]]}, {"dt": 0.02}, None),
    "2x1 box, drift": ({"aspect": 2, "walls": [[[1.0, 0.0], [1.0, 0.7]]]}, {"dt": 0.02}, [0.1, -0.05]),
    "no repulsion, thigmotaxis 1": ({"walls": MAZE}, {"dt": 0.02, "wall_repel_strength": 0.0, "thigmotaxis": 1.0}, None),
}
for name, (envp, agp, drift) in cases.items():
    np.random.seed(1)
    env = riab.Environment(envp)
    ag = riab.Agent(env, dict(agp, n_agents=4096, save_history=False, seed=99))
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        n = min(8192, steps - done)
        ag.simulate(n, chunk=1024, neurons=[], drift_velocity=drift, drift_to_random_strength_ratio=0.5)
        done += n
    st = ag.state_tensor[:, :4096]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    e = env.extent
    ok_finite = bool(torch.isfinite(st[:11]).all().item())
    inside = bool(((st[0] >= e[0]) & (st[0] <= e[1]) & (st[1] >= e[2]) & (st[1] <= e[3])).all().item())
    hd_norm = torch.sqrt(st[8] ** 2 + st[9] ** 2)
    d = ag.diagnostics
    print(f"{name:30s} {steps} steps x 4096 agents in {el:5.1f} s: finite={ok_finite} inside={inside} "
          f"|hd|-1 max {float((hd_norm - 1).abs().max()):.1e} speed mean {float(torch.sqrt(st[2] ** 2 + st[3] ** 2).mean()):.4f} {d}",
          flush=True)
    assert ok_finite and inside and d["bounce_saturations"] == 0

# ---- closed loop inside a TaskEnvironment (scripted policy, auto-reset): rewards finite, caches within bounds
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment  # noqa: E402

np.random.seed(3)
env = SpatialGoalEnvironment(possible_goal_positions="random_12",
                             goalcachekws=dict(reset_n_goals=3), teleport_on_reset=True, episode_terminate_delay=0.05,
                             episode_log_capacity=1 << 22)
ag = riab.Agent(env, {"n_agents": 4096, "dt": 0.01, "save_history": False})
env.add_agents(ag)
plan = env.make_step_plan(neurons=[], auto_reset=True, scripted_speed=11 * ag.speed_mean)
n_task = min(steps, 200_000)
t0 = time.perf_counter()
for _ in range(n_task // 100):
    plan.step(100)
torch.cuda.synchronize()
r = env.get_reward()
d = env.diagnostics
print(f"task loop: {n_task} steps x 4096 lanes in {time.perf_counter() - t0:.1f} s: rewards finite={bool(torch.isfinite(r).all())} "
      f"max active rewards {int(ag.reward.active()[2].max())} episodes {int(env._ep_count.item())} {d}", flush=True)
assert torch.isfinite(r).all() and d["reward_overflow"] == 0

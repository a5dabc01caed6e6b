"""Host-side cost of the reference-style task loop `a = policy(obs); obs, r, term, _, _ = env.step(a); env.reset(mask=term);
PCs.update()` at the cfg 2 shape (the eager API: four small kernels and their Python per step), next to the same loop as
a native plan (`env.make_step_plan(auto_reset=True, scripted_speed=...)`: one kernel per step)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment

np.random.seed(0)
env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=2),
                             teleport_on_reset=True, episode_terminate_delay=0.05, seed=1)
ag = riab.Agent(env, {"n_agents": 4096, "dt": 0.01, "seed": 1})
pcs = riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean", "save_spikes": False})
env.add_agents(ag)
speed = 11 * ag.speed_mean
N = 400
ag.preallocate_history(3 * N + 64)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        a = env._goal_vector(speed)
        obs, r, term, trunc, info = env.step(a)
        env.reset(mask=term)
        pcs.update()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"eager loop: {(t1 - t0) / N * 1e6:6.1f} us per step issued, {(time.perf_counter() - t0) / N * 1e6:6.1f} until synchronised")
plan = env.make_step_plan(capacity=N, auto_reset=True, scripted_speed=speed)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N // 2):
        plan.step(1)
    torch.cuda.synchronize()
    print(f"plan      : {(time.perf_counter() - t0) / (N // 2) * 1e6:6.1f} us per step")

"""Microseconds per closed-loop step of a native plan (cfg 2 shape by default): `python tools/step1_time.py [B n steps]`."""
import sys
import time

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ratinabox_amd as riab

TASK = "--task" in sys.argv
if TASK:
    sys.argv.remove("--task")
B, n, steps = (int(x) for x in (sys.argv[1:4] + ["4096", "1024", "256"][len(sys.argv) - 1:]))
np.random.seed(0)
if TASK:  # (bench.py --task's world)
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=2),
                                 teleport_on_reset=True, episode_terminate_delay=0.05, seed=1234)
else:
    env = riab.Environment()
ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 1234})
pcs = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": False})
if TASK:
    env.add_agents(ag)


def make_plan():
    if TASK:
        return env.make_step_plan(capacity=steps, auto_reset=True, scripted_speed=11 * ag.speed_mean)
    return ag.make_step_plan(capacity=steps)


out = []
for rep in range(6):
    ag.reset_history(); pcs.reset_history()
    plan = make_plan()
    plan.step(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.step(steps - 8)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / (steps - 8) * 1e6)
info = plan.info()
print(f"B={B} n={n}: {min(out):.2f} us/step best, {sorted(out)[len(out)//2]:.2f} median  ({B / min(out):.1f} M agent-steps/s)  {info}")

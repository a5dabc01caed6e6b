"""Where a one-launch step (csrc/riab_step1.hip) spends its time, on the device's constant clock.

Needs a library built with -DRIAB_STEP1_PROFILE (tools/build_variants.sh step1prof) loaded through RIAB_HIP_LIB:
three workgroups of the grid stamp s_memrealtime at the phase boundaries of every step (the last step's stamps stay).
Prints, per workgroup, microseconds since the first writer's entry."""
import sys

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ratinabox_amd as riab
from ratinabox_amd import _lib as L

TASK = "--task" in sys.argv
if TASK:
    sys.argv.remove("--task")
B, n, steps = (int(x) for x in (sys.argv[1:4] + ["4096", "1024", "64"][len(sys.argv) - 1:]))
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


plan = make_plan()
names = ["entry", "state+tables in", "motion done", "rates issued", "rates acked", "writer: all arrived", "writer: state acked",
         "task: verdict posted / read", "rewards updated", "goal pass 1", "pad + pass 2", "pass 3 + totals", "rates corrected / helper: next action stored", "new goals",
         "helper: draws done", "lane done"]
rows = []
for rep in range(5):
    plan.step(steps)
    torch.cuda.synchronize()
    w = plan._sync_words[L.step1_sync_words(ag._Bp):].cpu().numpy().view(np.uint64).reshape(-1, 16)[:3]
    rows.append(w.astype(np.float64))
    ag.reset_history(); pcs.reset_history()
    plan = make_plan()
r = np.median(np.stack(rows), 0)
t0 = r[0, 0]
print(f"B={B} n={n}: microseconds after the first writer's entry (100 MHz constant clock assumed), median of 5 runs' last step")
for slot, label in enumerate(("writer (0,0)", "workgroup (0,1)", "last workgroup")):
    print(f"  {label:18s}", "  ".join(f"{names[k]}: {(r[slot, k] - t0) * 0.01:6.2f}" for k in range(16) if r[slot, k] > 0))

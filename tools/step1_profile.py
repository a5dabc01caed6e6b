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

B, n, steps = (int(x) for x in (sys.argv[1:4] + ["4096", "1024", "64"][len(sys.argv) - 1:]))
np.random.seed(0)
env = riab.Environment()
ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 1234})
pcs = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": False})
plan = ag.make_step_plan(capacity=steps)
names = ["entry", "state+tables in", "motion done", "rates issued", "rates acked", "writer: all arrived", "writer: state acked"]
rows = []
for rep in range(5):
    plan.step(steps)
    torch.cuda.synchronize()
    w = plan._sync_words[L.step1_sync_words(ag._Bp):].cpu().numpy().view(np.uint64).reshape(-1, 8)[:3]
    rows.append(w.astype(np.float64))
    ag.reset_history(); pcs.reset_history()
    plan = ag.make_step_plan(capacity=steps)
r = np.median(np.stack(rows), 0)
t0 = r[0, 0]
print(f"B={B} n={n}: microseconds after the first writer's entry (100 MHz constant clock assumed), median of 5 runs' last step")
for slot, label in enumerate(("writer (0,0)", "workgroup (0,1)", "last workgroup")):
    print(f"  {label:18s}", "  ".join(f"{names[k]}: {(r[slot, k] - t0) * 0.01:6.2f}" for k in range(7) if r[slot, k] > 0))

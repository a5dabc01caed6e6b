"""Microseconds per closed-loop step of a native plan (cfg 2 shape by default): `python tools/step1_time.py [B n steps]`."""
import sys
import time

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ratinabox_amd as riab

B, n, steps = (int(x) for x in (sys.argv[1:4] + ["4096", "1024", "256"][len(sys.argv) - 1:]))
np.random.seed(0)
env = riab.Environment()
ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 1234})
pcs = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": False})
out = []
for rep in range(6):
    ag.reset_history(); pcs.reset_history()
    plan = ag.make_step_plan(capacity=steps)
    plan.step(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.step(steps - 8)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / (steps - 8) * 1e6)
info = plan.info()
print(f"B={B} n={n}: {min(out):.2f} us/step best, {sorted(out)[len(out)//2]:.2f} median  ({B / min(out):.1f} M agent-steps/s)  {info}")

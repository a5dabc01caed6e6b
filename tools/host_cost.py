"""Host cost of one `plan.step(1)` (cfg 2 world): total per step, the native call's share, and the same loop with the
native call stubbed out (pure Python bookkeeping)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab

B, n, K = 4096, 1024, 2048
np.random.seed(0)
env = riab.Environment()
ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 1})
pcs = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": False})
for mode in ("plain", "timed-call", "stub"):
    ag.reset_history(); pcs.reset_history()
    plan = ag.make_step_plan(capacity=K)
    plan.step(8)
    torch.cuda.synchronize()
    inner = [0.0]
    real = plan._step_fn
    if mode == "timed-call":
        def fn(h, k, s, real=real):
            t = time.perf_counter()
            rc = real(h, k, s)
            inner[0] += time.perf_counter() - t
            return rc
        plan._step_fn = fn
    elif mode == "stub":
        plan._step_fn = lambda h, k, s: 0
    t0 = time.perf_counter()
    for _ in range(K - 8):
        plan.step(1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode:10s}: {(t1 - t0) / (K - 8) * 1e6:6.2f} us per step issued, {(t2 - t0) / (K - 8) * 1e6:6.2f} us until synchronised, "
          f"native call {inner[0] / (K - 8) * 1e6:5.2f} us")
    if mode == "stub":
        plan._step_fn = real

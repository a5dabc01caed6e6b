"""Host cost of one closed-loop step (cfg 2 world).  `plan.step(1)`: total per step, the native call's share, and the
same loop with the native call stubbed out (pure Python bookkeeping); then the unchanged `Ag.update(); PCs.update()`
loop (plan.AutoStepper: value-keyed checks of the motion parameters and the tuning arrays on every call) the same way."""
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

# ---- the unchanged per-step loop `Ag.update(); PCs.update()` (plan.AutoStepper): the same three ways
import ratinabox_amd._lib as L
for mode in ("stub", "plain"):
    ag.reset_history(); pcs.reset_history()
    for _ in range(40):         # (the stepper records itself after a few eager steps)
        ag.update(); pcs.update()
    torch.cuda.synchronize()
    st = ag._plan
    assert st is not None and st.__class__.__name__ == "AutoStepper", st
    lib = L.lib
    real_a, real_p = lib.riab_plan_step_agent, lib.riab_plan_step_population
    if mode == "stub":
        lib.riab_plan_step_agent = lambda h, s: 0
        lib.riab_plan_step_population = lambda h, i, s: 0
    n = 1500 if mode == "plain" else 50    # (the stubbed loop must stay inside the open history chunk)
    t0 = time.perf_counter()
    for _ in range(n):
        ag.update(); pcs.update()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    lib.riab_plan_step_agent, lib.riab_plan_step_population = real_a, real_p
    print(f"per-step loop, {mode:6s}: {(t1 - t0) / n * 1e6:6.2f} us per step issued, {(t2 - t0) / n * 1e6:6.2f} us until synchronised")
    if mode == "stub":   # (the native cursors did not move: drop this stepper and its rows)
        st._a_pending, st._a_times = 0, []
        st._p_pending, st._p_times = [0] * len(st.neurons), [[] for _ in st.neurons]
        st.close()

"""Per-step replay of an imported trajectory (Ag.update(); PCs.update()), eager against the automatic step plan."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab

def run(auto, B=4096, n=1024, T=600):
    os.environ["RIAB_NO_AUTO_PLAN"] = "0" if auto else "1"
    np.random.seed(0)
    env = riab.Environment()
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01})
    pcs = riab.PlaceCells(ag, {"n": n})
    tt = np.linspace(0, 20, 401)
    ag.import_trajectory(times=tt, positions=np.stack((0.5 + 0.4 * np.cos(tt), 0.5 + 0.4 * np.sin(0.7 * tt)), axis=-1))
    ag.preallocate_history(T + 50)
    for _ in range(40):
        ag.update(); pcs.update()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(T):
        ag.update(); pcs.update()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return B * T / el, el / T * 1e6

for auto in (False, True, False, True):
    v, us = run(auto)
    print("auto plan %-5s  %.1f M agent-steps/s  %.1f us per step" % (auto, v / 1e6, us))

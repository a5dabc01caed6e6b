"""The reference's closed loops, unchanged, eager against the automatic step plan (cfg 2 shape):
  (a) Ag.update(drift_velocity=policy(pos)); PCs.update()
  (b) obs, r, term, trunc, info = env.step(actions); env.reset(mask=term); PCs.update()   (TaskEnvironment.py:1597-1614)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, get_goal_vector

B, n, T = 4096, 1024, 500

def loop_a(auto):
    os.environ["RIAB_NO_AUTO_PLAN"] = "0" if auto else "1"
    np.random.seed(0)
    env = riab.Environment()
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01})
    pcs = riab.PlaceCells(ag, {"n": n, "save_spikes": False})
    ag.preallocate_history(T + 60)
    target = torch.tensor([0.7, 0.3], dtype=torch.float64, device="cuda")
    def step():
        pos = ag.state_tensor[:2, :B].t()
        d = target - pos
        ag.update(drift_velocity=0.2 * d / d.norm(dim=1, keepdim=True).clamp_min(1e-9))
        pcs.update()
    for _ in range(40):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(T):
        step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return B * T / el / 1e6, el / T * 1e6

def loop_b(auto):
    os.environ["RIAB_NO_AUTO_PLAN"] = "0" if auto else "1"
    np.random.seed(0)
    env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=2),
                                 episode_terminate_delay=0.05, teleport_on_reset=True)
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01})
    pcs = riab.PlaceCells(ag, {"n": n, "save_spikes": False})
    env.add_agents(ag)
    env.reset()
    ag.preallocate_history(T + 60)
    def step():
        v = get_goal_vector(ag)
        obs, reward, terminal, truncated, info = env.step(11 * ag.speed_mean * v / v.norm(dim=1, keepdim=True).clamp_min(1e-9))
        env.reset(mask=terminal)
        pcs.update()
    for _ in range(40):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(T):
        step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return B * T / el / 1e6, el / T * 1e6

import gc
for name, f in (("update(drift_velocity=policy); PCs.update()", loop_a), ("env.step(); env.reset(mask); PCs.update()", loop_b)):
    for auto in (False, True, False, True):
        v, us = f(auto)
        gc.collect(); torch.cuda.empty_cache()
        print("%-46s auto plan %-5s %.1f M agent-steps/s  %.1f us per step" % (name, auto, v, us))

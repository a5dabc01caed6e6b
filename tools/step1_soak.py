"""Soak of the one-launch steps (plain and task) at the cfg 2 shape, alone and next to a foreign load on another stream
(large matrix products that take compute units away while the step kernels' workgroups wait for each other): no
workgroup may ever give up waiting (the plans' timeout counter stays 0) and the results must not depend on the company."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
from ratinabox_amd import _lib as L
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
B, n = 4096, 1024


def world(task):
    np.random.seed(0)
    if task:
        env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=2),
                                     teleport_on_reset=True, episode_terminate_delay=0.05, seed=1)
    else:
        env = riab.Environment()
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 1, "save_history": False})
    pcs = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": False, "save_history": False})
    if task:
        env.add_agents(ag)
        plan = env.make_step_plan(capacity=256, auto_reset=True, scripted_speed=11 * ag.speed_mean)
    else:
        plan = ag.make_step_plan(capacity=256)
    return env, ag, pcs, plan


def run(task, load):
    env, ag, pcs, plan = world(task)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.perf_counter()
    done = 0
    while done < STEPS:
        if load:
            with torch.cuda.stream(side):
                for _ in range(4):
                    a = torch.tanh(a @ a * 1e-3)
        plan.step(256)
        done += 256
        if done % (256 * 64) == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    info = plan.info()
    w = plan._sync_words
    timeouts = int(w[L.step1_sync_tail(ag._Bp) + L.STEP1_SYNC_TIMEOUTS].item())
    out = dict(state=ag.state_tensor.cpu().numpy(), rates=np.array(pcs.firingrate))
    if task:
        out["ts"] = env.task_state.cpu().numpy()
        out["episodes"] = int(env._ep_count.item())
    print(f"{'task ' if task else 'plain'} step, {'with' if load else 'no  '} foreign load: {done} steps in {dt:6.2f} s "
          f"({dt / done * 1e6:5.2f} us per step), one-launch steps {info['fused_steps']}, workgroups that gave up waiting: {timeouts}"
          + (f", episodes ended {out['episodes']}" if task else ""))
    assert timeouts == 0 and info["fused_steps"] == done
    plan.close()
    return out


for task in (False, True):
    ref = run(task, False)
    got = run(task, True)
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    print("   identical results with and without the foreign load")

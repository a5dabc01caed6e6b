"""Soak of the one-world task's step plan (`lanes="agents"`; three launches per step, the second stage of the world's step run
by whichever workgroup finishes last) at the cfg 2 shape, alone and next to a foreign load on another stream (large matrix
products that take compute units away, so the workgroups of a launch finish in a different order): the ticket and work-list
words must be back at zero, and the results must not depend on the company — agent state, rates, reward caches, the shared
state, the episode table."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
B, n = 4096, 1024


def world():
    np.random.seed(0)
    env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=4),
                                 goalkws={"goal_radius": 0.005}, teleport_on_reset=True, episode_terminate_delay=0.05, seed=1,
                                 lanes="agents")
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 1, "save_history": False})
    pcs = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": False, "save_history": False})
    env.add_agents(ag)
    plan = env.make_step_plan(capacity=256, auto_reset=True, scripted_speed=11 * ag.speed_mean)
    return env, ag, pcs, plan


def run(load):
    env, ag, pcs, plan = world()
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
    out = dict(state=ag.state_tensor.cpu().numpy(), rates=np.array(pcs.firingrate), ts=env.task_state.cpu().numpy(),
               world=env._world.cpu().numpy(), episodes=int(env._ep_count.item()), reward=env.get_reward().cpu().numpy())
    d = env.diagnostics
    print(f"one-world step plan, {'with' if load else 'no  '} foreign load: {done} steps in {dt:6.2f} s ({dt / done * 1e6:5.2f} us per "
          f"step), episodes ended {out['episodes']}, resets {d['resets']}, reward-cache overflows {d['reward_overflow']}, "
          f"ticket / work-list words {env._ticket.cpu().tolist()}")
    assert not env._ticket.any().item() and d["episode_log_overflow"] == 0
    plan.close()
    return out


ref = run(False)
got = run(True)
for k in ref:
    assert np.array_equal(ref[k], got[k]), k
print("   identical results with and without the foreign load")

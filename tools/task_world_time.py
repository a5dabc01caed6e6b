"""What a step of the one-world task (`lanes="agents"`, csrc/riab_task_world.hip) costs at the cfg 2 batch: the world step
kernel alone (HIP events around back-to-back launches at fixed positions — the decay and the "stands inside" masks of every
lane, the last workgroup's walk of the shared list) for a quiet step and for a step in which goals are consumed, and the eager
closed loop `a = goal vector; env.step(a); PCs.update()` with the reset when the world's episode ends, and the same loop as a
step plan (`env.make_step_plan(auto_reset=True, scripted_speed=...)`)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment

L = riab._lib
for B in (4096, 65536):
    np.random.seed(0)
    env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_12", goalcachekws=dict(reset_n_goals=8),
                                 goalkws={"goal_radius": 0.02}, teleport_on_reset=True, episode_terminate_delay=0.05, seed=1,
                                 lanes="agents")
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 1})
    pcs = riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean", "save_spikes": False}) if B == 4096 else None
    env.add_agents(ag)
    env_s, walls = env.device_tables(ag.state_tensor.device)
    task = env._task_struct()
    st = ag.state_tensor

    def launch(t_env):
        rc = L.lib.riab_task_world_step(env_s, task, L.ptr(env.task_state), L.ptr(env._world), L.ptr(st[0]), L.ptr(st[1]), B,
                                        float(t_env), L.ptr(env._reward), L.ptr(env._terminal), L.ptr(env._met),
                                        L.ptr(env._cand), L.ptr(env._ticket), L.ptr(env._diag), L.current_stream())
        assert rc == 0

    # the first launch consumes what the agents stand in; the following ones are quiet steps
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    left0 = len(env.goal_cache)
    e0.record()
    launch(0.01)
    e1.record()
    N = 200
    for i in range(N):
        launch(0.01)
    e2.record()
    torch.cuda.synchronize()
    print(f"B={B}: world step kernel, first (goals {left0} -> {len(env.goal_cache)}): {e0.elapsed_time(e1) * 1e3:7.1f} us; "
          f"quiet steps back to back: {e1.elapsed_time(e2) / N * 1e3:6.2f} us each")
    if pcs is None:
        continue
    env.reset()
    speed = 11 * ag.speed_mean
    N = 400
    ag.preallocate_history(5 * N + 64)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        resets = 0
        for k in range(N):
            a = env._goal_vector(speed)
            obs, r, term, trunc, info = env.step(a)
            pcs.update()
            if k % 16 == 15 and bool(term[0].item()):   # (one flag for the world; looked at every 16 steps)
                env.reset()
                resets += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B}: eager loop (goal vector + motion + world step + 1024 place cells): {dt / N * 1e6:6.1f} us per step "
              f"= {B * N / dt / 1e6:6.1f} M agent-steps/s, {resets} resets")
    plan = env.make_step_plan(capacity=N, auto_reset=True, scripted_speed=speed)
    for rep in range(3):
        torch.cuda.synchronize()
        t0, l0 = time.perf_counter(), plan.info()['launches']
        for _ in range(N // 2):
            plan.step(1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B}: the same as a step plan (one native call per step, the reset decided on the device; "
              f"{(plan.info()['launches'] - l0) // (N // 2)} launches per step): {dt / (N // 2) * 1e6:6.1f} us per step "
              f"= {B * (N // 2) / dt / 1e6:6.1f} M agent-steps/s")

"""Soak of three more task plans at 4096 agents, alone and beside a foreign load on another stream (as tools/task_world_soak.py): the
one-world plan of one population, the replica plan and the one-world plan of three store-bound populations in one launch per step —
the results must not depend on the company.  python tools/task_plans_soak.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import ratinabox_amd as riab
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
STEPS = 300_000
def world(lanes, multi):
    np.random.seed(0)
    env = SpatialGoalEnvironment(params={}, possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=4 if lanes == "agents" else 2),
                                 goalkws={"goal_radius": 0.005} if lanes == "agents" else {}, teleport_on_reset=True, episode_terminate_delay=0.05, seed=1, lanes=lanes)
    ag = riab.Agent(env, {"n_agents": 4096, "dt": 0.01, "seed": 1, "save_history": False})
    pops = [riab.PlaceCells(ag, {"n": 512, "wall_geometry": "euclidean", "save_spikes": True, "max_fr": 20, "save_history": False})]
    if multi:
        pops += [riab.GridCells(ag, {"n": 128, "save_spikes": True, "max_fr": 20, "save_history": False}),
                 riab.HeadDirectionCells(ag, {"n": 32, "save_spikes": False, "save_history": False})]
    env.add_agents(ag)
    plan = env.make_step_plan(neurons=pops, capacity=256, auto_reset=True, scripted_speed=11 * ag.speed_mean)
    return env, ag, pops, plan
def run(lanes, multi, load):
    env, ag, pops, plan = world(lanes, multi)
    side = torch.cuda.Stream(); a = torch.randn(4096, 4096, device="cuda"); done = 0; t0 = time.perf_counter()
    while done < STEPS:
        if load:
            with torch.cuda.stream(side):
                for _ in range(4): a = torch.tanh(a @ a * 1e-3)
        plan.step(256); done += 256
        if done % (256 * 64) == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    info = plan.info()
    out = dict(state=ag.state_tensor.cpu().numpy(), ts=env.task_state.cpu().numpy(), episodes=int(env._ep_count.item()))
    for i, p in enumerate(pops):
        out[f"fr{i}"] = np.array(p.firingrate)
    print(f"{lanes} multi={multi} load={load}: {dt/done*1e6:.2f} us/step fused {info['fused_steps']}/{done} launches {info['launches']} episodes {out['episodes']} timeouts {ag.diagnostics.get('step1_timeouts_recovered')}")
    plan.close()
    return out
for lanes, multi in (("agents", False), ("replicas", True), ("agents", True)):
    ref = run(lanes, multi, False); got = run(lanes, multi, True)
    bad = [k for k in ref if not np.array_equal(ref[k], got[k])]
    print("   identical" if not bad else f"   DIFFERENT: {bad}")

"""The reference's simple example (README.md:43, demos/simple_example.ipynb), unchanged except for the import,
then the same world batched, fused and in a closed loop.

    python examples/simple_example.py          # needs an MI355X (the package has no CPU path)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ratinabox_amd import Agent, Environment, PlaceCells, GridCells, BoundaryVectorCells  # noqa: E402

# ---- 1. the reference script: one agent, a per-step Python loop ------------------------------------------------
np.random.seed(0)
Env = Environment()
Ag = Agent(Env, params={"dt": 0.01})
PCs = PlaceCells(Ag, params={"n": 100})
for _ in range(int(2 / Ag.dt)):
    Ag.update()
    PCs.update()
print("1 agent:", np.asarray(Ag.history["pos"]).shape, np.asarray(PCs.history["firingrate"]).shape,
      "mean rate %.3f" % np.mean(PCs.history["firingrate"]))

# ---- 2. the batched extension: 4096 independent agents in a maze, three populations, the same loop ------------
Env = Environment({"walls": [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]]]})
Ag = Agent(Env, params={"dt": 0.01, "n_agents": 4096, "seed": 7})
pops = [PlaceCells(Ag, {"n": 256}), GridCells(Ag, {"n": 128}), BoundaryVectorCells(Ag, {"n": 64})]
for _ in range(50):
    Ag.update()
    for N in pops:
        N.update()
print("4096 agents:", Ag.pos.shape, [N.firingrate.shape for N in pops])

# ---- 3. the fused path: no return to Python between steps; histories stay in HBM ------------------------------
torch.cuda.synchronize()
t0 = time.perf_counter()
Ag.simulate(1000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fr, spikes = pops[0].get_history_tensors()           # device tensors [T, n, B]
print("simulate(1000): %.1f M agent-steps/s, PlaceCells history %s on %s" % (4096 * 1000 / dt / 1e6, tuple(fr.shape), fr.device))

# ---- 4. a closed loop: a policy producing drift_velocity every step, one native call per step ------------------
Ag2 = Agent(Environment(), params={"dt": 0.01, "n_agents": 4096})
PCs2 = PlaceCells(Ag2, {"n": 256})
plan = Ag2.make_step_plan(capacity=200)
target = torch.tensor([[0.8], [0.2]], dtype=torch.float64, device="cuda")
for _ in range(200):
    pos = Ag2.state_tensor[0:2]                        # device rows x, y (no host round trip)
    plan.step(drift_velocity=1.5 * (target - pos), drift_to_random_strength_ratio=10.0)
print("closed loop: mean distance to the target %.3f m" % float(torch.linalg.vector_norm(Ag2.state_tensor[0:2] - target, dim=0).mean()))

# ---- 5. the batched TaskEnvironment: goals, rewards, episodes on the device -----------------------------------
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment  # noqa: E402

env = SpatialGoalEnvironment(possible_goal_positions="random_5", goalcachekws=dict(reset_n_goals=2), teleport_on_reset=True)
Ag3 = Agent(env, params={"dt": 0.01, "n_agents": 1024})
env.add_agents(Ag3)
env.reset(seed=3)
total = torch.zeros(1024, dtype=torch.float64, device="cuda")
for _ in range(300):
    action = env._goal_vector(0.5)                     # a scripted policy: head for a goal at 0.5 m/s
    obs, reward, terminal, truncated, info = env.step(action)
    total += reward
    env.reset(mask=terminal)
print("task: %d episodes finished, mean return per lane %.2f" % (len(env.episodes["episode"]), float(total.mean())))

# ---- 6. ... and the same 1024 agents in ONE world (the reference's multi-agent task, agentmode="interact") -----
# one episode, one shared goal list: a goal is consumed for everybody by the first agent found inside it
world = SpatialGoalEnvironment(possible_goal_positions="random_5", goalcachekws=dict(reset_n_goals=3), goalkws={"goal_radius": 0.01},
                               teleport_on_reset=True, lanes="agents")
Ag4 = Agent(world, params={"dt": 0.01, "n_agents": 1024})
world.add_agents(Ag4)
plan = world.make_step_plan(capacity=300, auto_reset=True, scripted_speed=0.5)   # three launches per step, one native call
total = torch.zeros(1024, dtype=torch.float64, device="cuda")
for _ in range(300):
    plan.step(1)
    total += world.get_reward()
print("one world: %d episodes finished, %d of 1024 agents were rewarded" % (len(world.episodes["episode"]), int((total > 0).sum())))

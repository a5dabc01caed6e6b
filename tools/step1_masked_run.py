"""One process's view of the one-launch step on whatever part of the chip it is given: tests/test_gpu_step1_device.py runs
this under HSA_CU_MASK (and with the residency rule / the spin limit switched by RIAB_STEP1_RESIDENCY / RIAB_STEP1_SPIN)
and compares the digests with an unmasked run's.  Prints one JSON line."""
import hashlib
import json
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ratinabox_amd as riab  # noqa: E402
from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment  # noqa: E402


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:24]


def task_world(B, n, steps):
    np.random.seed(5)
    env = SpatialGoalEnvironment(params={"walls": [[[0.5, 0.0], [0.5, 0.4]]]}, possible_goal_positions="random_5",
                                 goalcachekws=dict(reset_n_goals=2, goalorder="nonsequential"), goalkws={"goal_radius": 0.2},
                                 episode_terminate_delay=0.0, teleport_on_reset=True, seed=13)
    ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 21})
    np.random.seed(6)
    pops = [riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": True, "max_fr": 20}),
            riab.GridCells(ag, {"n": max(8, n // 4), "save_spikes": False})]
    env.add_agents(ag)
    plan = env.make_step_plan(neurons=pops, capacity=steps, auto_reset=True, scripted_speed=0.9)
    plan.step()                      # (+ the launch that works out the first scripted action)
    l0 = plan.info()["launches"]
    for _ in range(steps - 1):
        plan.step()
    info = plan.info()
    info["launches"] = (info["launches"] - l0) * steps / (steps - 1)
    torch.cuda.synchronize()
    arrays = [ag.state_tensor.cpu().numpy(), ag.get_history_tensor().cpu().numpy(), env.task_state.cpu().numpy(),
              env.get_reward().cpu().numpy(), np.asarray(pops[0].firingrate), np.asarray(pops[1].firingrate)]
    for p in pops:
        fr, sp = p.get_history_tensors()
        arrays.append(fr.cpu().numpy())
        if p.save_spikes:
            arrays.append(sp.cpu().numpy())
    d = ag.diagnostics
    plan.close()
    return {"digest": digest(arrays), "launches_per_step": info["launches"] / steps, "fused_steps": info["fused_steps"],
            "compute_units": info["compute_units"], "give_ups_recovered": d["step1_timeouts_recovered"],
            "steps_recovered": d["step1_recovered_steps"], "resets": int(env.diagnostics.get("resets", 0)),
            "episodes": len(env.episodes["episode"])}


def plain_world(B, n, steps):
    np.random.seed(7)
    env = riab.Environment()
    ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 3})
    np.random.seed(8)
    pops = [riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean"}), riab.HeadDirectionCells(ag, {"n": 20, "save_spikes": True, "max_fr": 20})]
    plan = ag.make_step_plan(capacity=steps)
    for _ in range(steps):
        plan.step()
    info = plan.info()
    torch.cuda.synchronize()
    arrays = [ag.state_tensor.cpu().numpy(), ag.get_history_tensor().cpu().numpy()]
    for p in pops:
        fr, sp = p.get_history_tensors()
        arrays += [fr.cpu().numpy(), sp.cpu().numpy()]
    d = ag.diagnostics
    plan.close()
    return {"digest": digest(arrays), "launches_per_step": info["launches"] / steps, "fused_steps": info["fused_steps"],
            "compute_units": info["compute_units"], "give_ups_recovered": d["step1_timeouts_recovered"],
            "steps_recovered": d["step1_recovered_steps"]}


if __name__ == "__main__":
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = {"counted_compute_units": riab._lib.compute_units(0),
               "runtime_compute_units": torch.cuda.get_device_properties(0).multi_processor_count,
               "task4096": task_world(4096, 256, 40), "task8192": task_world(8192, 64, 30),
               "task32768": task_world(32768, 16, 12), "plain2048": plain_world(2048, 300, 30)}
        out["recovery_warnings"] = sum("one-launch step" in str(w.message) for w in caught)
    print("RESULT " + json.dumps(out))

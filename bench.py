#!/usr/bin/env python
"""bench.py — agent-steps/s of the batched hot path on MI355X.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "cfg2"): per GPU 4096 batched agents x 1024 gaussian
PlaceCells, open 1x1 m box, dt = 10 ms, default motion parameters, in-kernel Philox
noise (seed 1234), synthetic random-walk trajectories.  One STEP = one Agent.update() +
one PlaceCells.update() for all 4096 agents of the rank (positions/velocities advanced,
trajectory history row written, 1024 firing rates per agent written to HBM).  Agents are
independent: ranks own disjoint agent ranges (weak scaling, no collective on the step
path; the only communication is the barrier / max-reduce around the timed region).

Rank 0 prints ONE JSON line: the contract fields plus
  roofline      the dominant kernel (PlaceCells rate kernel): algorithmic bytes per launch /
                its average duration, measured with HIP events on the stream it runs on
  cpu_baseline  the float64 NumPy oracle of the same path timed on the host (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured-achievable copy

CONFIGS = {
    # name: agents per GPU, cells, walls, spikes
    "cfg2": dict(agents=4096, place=1024, grid=0, bvc=0, hdc=0, walls=[], spikes=False,
                 desc="4096 agents x 1024 gaussian PlaceCells, open box, dt=10ms (BASELINE configs[1])"),
    "cfg4": dict(agents=4096, place=4096, grid=0, bvc=0, hdc=0, walls=[], spikes=False,
                 desc="4096 agents/GPU x 4096 gaussian PlaceCells, open box (BASELINE configs[3] per-GPU shard)"),
    "cfg3": dict(agents=4096, place=0, grid=1024, bvc=256, hdc=0, spikes=False,
                 walls=[[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]],
                        [[.3, .5], [.7, .5]]],
                 desc="4096 agents x (1024 GridCells + 256 BVCs), 5 interior walls (BASELINE configs[2])"),
    "cfg5": dict(agents=8192, place=1024, grid=512, bvc=256, hdc=256, walls=[], spikes=True,
                 desc="8192 agents/GPU x 2048 mixed cells with Poisson spikes (BASELINE configs[4] per-GPU shard)"),
}


def bytes_per_agent_step(cfg):
    """4*n_cells (+1*n_cells if spiking) + 112 B of state/history."""
    n = cfg["place"] + cfg["grid"] + cfg["bvc"] + cfg["hdc"]
    return 4 * n + (n if cfg["spikes"] else 0) + 112


def build_world(riab, cfg, rank, precision, seed=1234, task=False):
    np.random.seed(1000 + rank)
    if task:  # the same world inside a goal-directed task (closed loop: contribs/TaskEnvironment.py)
        from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
        env = SpatialGoalEnvironment(params={"walls": cfg["walls"]}, possible_goal_positions="random_8",
                                     goalcachekws=dict(reset_n_goals=2), teleport_on_reset=True,
                                     episode_terminate_delay=0.05, seed=seed)
    else:
        env = riab.Environment({"walls": cfg["walls"]})
    B = cfg["agents"]
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": seed, "agent_id0": rank * B, "precision": precision})
    pops = []
    np.random.seed(0)  # identical cell tables on every rank
    common = {"save_spikes": cfg["spikes"]}
    if cfg["place"]:
        pops.append(riab.PlaceCells(ag, dict(common, n=cfg["place"], wall_geometry="euclidean")))
    if cfg["grid"]:
        pops.append(riab.GridCells(ag, dict(common, n=cfg["grid"])))
    if cfg["bvc"]:
        pops.append(riab.BoundaryVectorCells(ag, dict(common, n=cfg["bvc"])))
    if cfg["hdc"]:
        pops.append(riab.HeadDirectionCells(ag, dict(common, n=cfg["hdc"])))
    if task:
        env.add_agents(ag)
    return env, ag, pops


def cpu_baseline(cfg, budget_s=12.0):
    """The float64 NumPy oracle (oracle/riab_oracle.py) of the same path, one host core,
    on a bounded sample of the workload: 256 agents, as many steps as fit the budget."""
    from oracle import riab_oracle as orc
    rs = np.random.RandomState(0)
    B = 256
    env = orc.EnvSpec(walls=np.asarray(cfg["walls"], dtype=float).reshape(-1, 2, 2))
    st = orc.init_state(env, B, 0.08, rs)
    n = cfg["place"]
    side = int(np.sqrt(max(n, 1)))
    gx = (np.arange(side) + 0.5) / side
    centres = np.stack(np.meshgrid(gx, gx), -1).reshape(-1, 2)
    steps = 0
    t0 = time.perf_counter()
    while True:
        z = rs.standard_normal((2, B))
        st = orc.agent_step(env, st, 0.01, z[0], z[1])
        if n:
            fr = orc.place_cells(env, st["pos"], centres, 0.2)
            if cfg["spikes"]:
                orc.spikes_ref(fr, rs.random_sample(fr.shape), 0.01)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {"value": B * steps / el, "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": f"{B} agents x {steps} steps of Agent.update + PlaceCells({n}).update, float64 NumPy oracle, "
                      f"{el:.1f} s on 1 of {os.cpu_count()} host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--chunk", type=int, default=128, help="steps per kernel launch in the fused path")
    ap.add_argument("--precision", type=int, default=64, choices=(32, 64), help="motion-kernel arithmetic")
    ap.add_argument("--per-step", action="store_true", help="time the drop-in per-step API instead of simulate()")
    ap.add_argument("--plan", action="store_true", help="time the closed-loop path through a native step plan")
    ap.add_argument("--plan-batch", type=int, default=1, help="steps per riab_plan_step call (1 = closed loop)")
    ap.add_argument("--task", action="store_true",
                    help="closed loop through the batched TaskEnvironment: goal-seeking actions, rewards, goal checks "
                         "and per-lane auto-reset every step (implies --plan)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-history", action="store_true", help="ring buffers instead of a full T-long history")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run", file=sys.stderr)
    share = os.environ.get("RIAB_BENCH_SHARE_GPU") == "1"  # test hook: all ranks on cuda:0, gloo control plane
    if share:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import ratinabox_amd as riab
    cfg = CONFIGS[args.config]
    args.plan = args.plan or args.task
    env, ag, pops = build_world(riab, cfg, rank, args.precision, task=args.task)
    # the full rate history of K steps must fit in HBM next to the warmup's; otherwise stream
    # through ring buffers (every byte is still written, the oldest rows are overwritten)
    n_cells = sum(int(p.n) for p in pops)
    need = (args.steps + args.warmup) * cfg["agents"] * (n_cells * (5 if cfg["spikes"] else 4) + 32)
    free_b, _total_b = torch.cuda.mem_get_info()
    if need > 0.6 * free_b:
        args.no_history = True
    if args.no_history:
        ag.save_history = False
        for p in pops:
            p.save_history = False
    B = cfg["agents"]
    K, W = args.steps, args.warmup
    # short runs: about four chunks in flight so that the two pipeline stages still overlap (below ~64 steps
    # a single launch of each stage is faster than several short ones) [MI355X: K=200 853 -> 961 M/s]
    if K < 4 * args.chunk:
        args.chunk = K if K <= 64 else max(32, (K // 4 + 3) // 4 * 4)

    plan = {"p": None}

    def run(n_steps):
        if args.plan:
            if plan["p"] is None or ag._plan is not plan["p"]:
                if args.task:
                    plan["p"] = env.make_step_plan(capacity=max(K, W), auto_reset=True, scripted_speed=11 * ag.speed_mean)
                else:
                    plan["p"] = ag.make_step_plan(capacity=max(K, W))
            nb = args.plan_batch
            for _ in range(n_steps // nb):
                plan["p"].step(nb)
            if n_steps % nb:
                plan["p"].step(n_steps % nb)
        elif args.per_step:
            for _ in range(n_steps):
                ag.update()
                for p in pops:
                    p.update()
        else:
            ag.simulate(n_steps, chunk=args.chunk)

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- warmup (untimed); its history is dropped so the timed run owns fresh HBM
    run(W)
    torch.cuda.synchronize()
    ag.reset_history()
    for p in pops:
        p.reset_history()

    # ---- HIP-event timing of the dominant kernel's launches, on the stream it runs on
    spans = []

    # Every event record is a packet on the rate stream, which runs back to back: bracketing all eight
    # launches costs ~3 % of the whole-path value [MI355X], so every second launch is bracketed.
    seen = {"n": 0}

    def hook(pop, what, tc):
        if pop is not pops[0]:
            return
        if what == "begin":
            seen["n"] += 1
        if seen["n"] % 2 == 1:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if what == "begin":
            spans.append([ev, None, tc])
        else:
            spans[-1][1] = ev

    if not (args.per_step or args.plan):
        ag._profile_hook = hook
    ag.preallocate_history(K)  # output buffers are allocated outside the timed region
    torch.cuda.synchronize()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_units = world * B * K
    value = total_units / elapsed
    bpu = bytes_per_agent_step(cfg)

    roofline = None
    if spans:
        ms = [a.elapsed_time(b) for a, b, _ in spans]
        units = [B * tc for _, _, tc in spans]
        n0 = pops[0].n
        # the dominant kernel moves, per agent-step, 4*n0 B of rates (+n0 B spikes) and reads 8 B of position;
        # the contract's `achieved` uses SURVEY §8(d)'s per-unit figure restricted to this population
        unit_bytes = 4 * n0 + (n0 if cfg["spikes"] else 0) + 112
        avg_ms = float(np.mean(ms))
        avg_units = float(np.mean(units))
        achieved = unit_bytes * avg_units / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.config}.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = round(json.load(f).get("hbm_bytes_per_unit") * avg_units)  # PMC bytes/unit x units/launch
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": f"rate_kernel<{type(pops[0]).__name__}>", "launches": len(ms),
                    "launches_in_timed_region": seen["n"],
                    "avg_launch_ms": round(avg_ms, 4), "units_per_launch": int(avg_units),
                    "bytes_per_unit": unit_bytes, "kernel_own_bytes_per_unit": 4 * n0 + 8,
                    # the averages above are over ALL timed launches (what rocprofv3 --stats averages too)
                    "full_chunk_launch_ms": round(float(np.mean([m for m, u in zip(ms, units) if u == max(units)])), 4),
                    "frac_of_measured_copy_bw_6290": round(achieved / 6290.0, 4)}

    # the chip's measured store ceiling in this same process (riab_fill: one float4 per thread,
    # address-ordered), for context next to the spec peak
    store_ceiling = None
    if rank == 0 and roofline is not None:
        L = riab._lib
        nbytes = 1 << 30
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            L.lib.riab_fill(L.ptr(buf), nbytes, 1.0, L.current_stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.lib.riab_fill(L.ptr(buf), nbytes, 1.0, L.current_stream())
        e1.record()
        torch.cuda.synchronize()
        store_ceiling = 5 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
        roofline["measured_store_ceiling_GBps"] = round(store_ceiling, 1)
        roofline["frac_of_measured_store_ceiling"] = round(roofline["achieved"] / store_ceiling, 4)
        del buf

    if rank == 0:
        out = {
            "metric": "agent-steps/sec (whole node) at 4096 agents x 1024 PlaceCells",
            "value": round(value, 1), "unit": "agent-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 rates / f%d motion" % args.precision, "data": "synthetic",
            "config": {"workload": args.config + ": " + cfg["desc"], "agents_per_gpu": B,
                       "cells": {k: cfg[k] for k in ("place", "grid", "bvc", "hdc")},
                       "parallelism": f"agent-sharded x{world}, no step-path collective",
                       "api": ("TaskEnvironment step plan: action, Agent.update, rewards/goals, auto-reset, rates; one "
                               "native call per step" if args.task else
                               "step plan (one native call per step)" if args.plan else "per-step update()"
                               if args.per_step else f"simulate(), {args.chunk} steps/launch"),
                       "history": "ring" if args.no_history else "full", "spikes": cfg["spikes"],
                       "bytes_per_agent_step": bpu},
            "hbm_GBps_whole_path": round(value / world * bpu / 1e9, 1),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        diag = ag.diagnostics
        if args.task:
            diag = dict(diag, **env.diagnostics, episodes_finished=len(env.episodes["episode"]))
        out["diagnostics"] = diag
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

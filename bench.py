#!/usr/bin/env python
"""bench.py — agent-steps/s of the batched hot path on MI355X.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python bench.py --gpus N ...      # launches the N ranks itself (torch.distributed.run on 127.0.0.1), or is
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # launched as N ranks by the driver

Workload (BASELINE.json configs[1], "cfg2"): per GPU 4096 batched agents x 1024 gaussian
PlaceCells, open 1x1 m box, dt = 10 ms, default motion parameters, in-kernel Philox
noise (seed 1234), synthetic random-walk trajectories.  One STEP = one Agent.update() +
one PlaceCells.update() for all 4096 agents of the rank (positions/velocities advanced,
trajectory history row written, 1024 firing rates per agent written to HBM).  Agents are
independent: ranks own disjoint agent ranges (weak scaling, no collective on the step
path; the only communication is the barrier / max-reduce around the timed region).

Timing: W untimed warm-up steps, then the K-step region — bracketed by barrier +
torch.cuda.synchronize() on both sides, maximum over ranks — is run `repeats` times, every
repeat doing the full work into freshly reserved history rows; `value` comes from the MEDIAN
repeat, the spread is reported (`timed_region_ms`).  A 20-step region is ~80 us: one sample
of it is mostly host jitter.

Rank 0 prints ONE JSON line: the contract fields plus
  roofline      the dominant kernel (the PlaceCells rate kernel): algorithmic bytes per launch /
                its average duration, measured with HIP events on the stream it runs on
  cpu_baseline  the float64 NumPy oracle of the same path timed on the host cores (rank 0, N=1)
"""
import argparse
import json
import os

# (before anything initialises the HIP runtime: see ratinabox_amd/__init__.py — under torch.distributed.run the process
# also holds an RCCL communicator, and with the default 4 hardware queues the two streams of simulate() shared one)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_ALL_CPUS = None       # the process's affinity mask before bind_rank_to_gpu_numa narrowed it
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured-achievable copy
# BoundaryVectorCells: one v_exp_f32 per (cell, direction, position) term is irreducible.  Issue rates of single
# instructions measured on MI355X at 8 waves per SIMD (tools/exp_bench.hip, profiles/r03_exp_bench.txt), in cycles per
# wave-instruction per SIMD at 2.4 GHz: v_exp_f32 9.44 (CDNA4's transcendental rate: not the 16 of a quarter-rate
# unit), v_pk_fma_f32 5.17, v_pk_add_f32 4.98.
VALU_CYCLES = {"v_exp_f32": 9.44, "v_pk_fma_f32": 5.17, "v_pk_add_f32": 4.98}
VALU_SIMDS, VALU_CLOCK_GHZ = 1024, 2.4
# peak = the v_exp_f32 issue ceiling alone: 1024 SIMDs x 64 lanes x 2.4 GHz / 9.44 cycles = 16.7 T exponentials/s
VALU_EXP_PEAK_TTERMS = VALU_SIMDS * 64 * VALU_CLOCK_GHZ / VALU_CYCLES["v_exp_f32"] / 1e3
# the kernel's own instruction mix per term (1 v_exp + 1 v_pk_fma + 1/2 v_pk_add: two terms per packed instruction):
# 17.1 cycles if nothing overlapped = 9.2 T terms/s; measured with that mix in a register-only loop: 15.7 cycles = 10.0
VALU_MIX_SERIAL_TTERMS = VALU_SIMDS * 64 * VALU_CLOCK_GHZ / (9.44 + 5.17 + 0.5 * 4.98) / 1e3
VALU_MIX_MEASURED_TTERMS = 10.0

# steps of the cfg3 / cfg4 / cfg5 runs in the `secondary` block of the cfg2 line: what `--config cfgN` runs by default.
# (More than 256: every run then takes the chunk form of the rate stage, so that `rate_kernel_gated` in a rocprofv3 summary of the driver's command is the
# headline configuration's kernel alone — cfg4's PlaceCells would otherwise share its name at 150 times its size.)
SECONDARY_STEPS = 1024

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
    # cfg 3 in a wall-heavy room: a comb maze of 60 interior segments (RIAB_MAX_WALLS = 64 with the room's four): VERDICT
    # r5 #5 — what the float64 wall loop of the motion step and stage A of the boundary vector cells cost at N_w = 64
    "cfg3_64w": dict(agents=4096, place=0, grid=1024, bvc=256, hdc=0, spikes=False, walls="comb60",
                     desc="4096 agents x (1024 GridCells + 256 BVCs), 60 interior walls (a comb maze: 64 wall segments)"),
    "cfg5": dict(agents=8192, place=1024, grid=512, bvc=256, hdc=256, walls=[], spikes=True,
                 desc="8192 agents/GPU x 2048 mixed cells with Poisson spikes (BASELINE configs[4] per-GPU shard)"),
}


def metric_name(cfg):
    cells = " + ".join(f"{cfg[k]} {name}" for k, name in (("place", "PlaceCells"), ("grid", "GridCells"),
                                                          ("bvc", "BoundaryVectorCells"), ("hdc", "HeadDirectionCells"))
                       if cfg[k])
    return f"agent-steps/sec (whole node) at {cfg['agents']} agents x {cells}"


def bytes_per_agent_step(cfg):
    """4*n_cells (+1*n_cells if spiking) + 112 B of state/history."""
    n = cfg["place"] + cfg["grid"] + cfg["bvc"] + cfg["hdc"]
    return 4 * n + (n if cfg["spikes"] else 0) + 112


def comb_walls(n=60):
    """n interior wall segments: teeth of two interleaved combs (from the floor and from the ceiling), every tooth in two
    collinear pieces with a doorway — a room whose every position has several walls within the repel distance."""
    walls, teeth = [], n // 2
    for i in range(teeth):
        x = (i + 1) / (teeth + 1)
        if i % 2 == 0:   # from the floor: [0, 0.3] and [0.4, 0.7]
            walls += [[[x, 0.0], [x, 0.3]], [[x, 0.4], [x, 0.7]]]
        else:            # from the ceiling: [1, 0.7] and [0.6, 0.3]
            walls += [[[x, 1.0], [x, 0.7]], [[x, 0.6], [x, 0.3]]]
    return walls[:n]


def resolve(cfg):
    return dict(cfg, walls=comb_walls(60)) if cfg["walls"] == "comb60" else cfg


def build_world(riab, cfg, rank, seed=1234, task=False, one_world=False):
    import numpy as np
    np.random.seed(1000 + rank)
    cfg = resolve(cfg)
    if task:  # the same world inside a goal-directed task (closed loop: contribs/TaskEnvironment.py)
        from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
        # (one_world: the rank's agents share ONE task — the reference's multi-agent environment, agentmode="interact":
        # small goals, or 4096 agents empty the list on the spot)
        kw = dict(lanes="agents", goalkws={"goal_radius": 0.005}) if one_world else {}
        env = SpatialGoalEnvironment(params={"walls": cfg["walls"]}, possible_goal_positions="random_8",
                                     goalcachekws=dict(reset_n_goals=4 if one_world else 2), teleport_on_reset=True,
                                     episode_terminate_delay=0.05, seed=seed, **kw)
    else:
        env = riab.Environment({"walls": cfg["walls"]})
    B = cfg["agents"]
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": seed, "agent_id0": rank * B})
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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """(cores this process may actually use, what limits them): the affinity mask and the cgroup's CPU quota — a
    container with `cpu.max = 1000000 100000` runs ten cores' worth however many the host shows."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    info["affinity"] = aff
    quota = None
    for path, v1 in (("/sys/fs/cgroup/cpu.max", False), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", True)):
        try:
            with open(path) as f:
                txt = f.read().split()
            if v1:
                q = int(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read().split()[0])
                quota = q / per if q > 0 else None
                info["cgroup_cpu"] = f"cfs_quota_us {q} / cfs_period_us {per}"
            else:
                info["cgroup_cpu_max"] = " ".join(txt)
                quota = int(txt[0]) / int(txt[1]) if txt[0] != "max" else None
            break
        except (OSError, ValueError, IndexError):
            continue
    try:
        info["loadavg_1min"] = os.getloadavg()[0]
    except OSError:
        pass
    n = aff if quota is None else max(1, min(aff, int(quota)))
    info["cgroup_quota_cores"] = quota
    info["usable"] = n
    return n, info


def cpu_baseline(cfg, seconds=8.0):
    """The float64 NumPy oracle (oracle/riab_oracle.py) of the same path on the host cores, SURVEY.md §8(d)(ii):
    one single-threaded worker process (`oracle/cpu_bench.py`) per USABLE core — the affinity mask capped by the
    cgroup's CPU quota —, each stepping its share of the GPU batch (agents / workers, at least 16) for `seconds`
    seconds; the rates add up.  Every worker reports its CPU seconds next to its wall seconds: on a shared host the sum
    is what the box gave this job, and `workers` says how evenly.  The one-core figure on a 256-agent batch (round 1's
    number) is kept beside it."""
    all_cpus = _ALL_CPUS or os.sched_getaffinity(0)   # (the workers run on every core, not on the rank's NUMA share)
    usable, host = usable_cores()
    quota = host["cgroup_quota_cores"]
    cores = max(1, min(len(all_cpus), int(quota))) if quota else len(all_cpus)
    host["workers_started"] = cores
    B = cfg["agents"]
    per = max(16, B // cores)
    n = cfg["place"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    base = [sys.executable, "-m", "oracle.cpu_bench", "--cells", str(n), "--walls-json", json.dumps(cfg["walls"])]
    if cfg["spikes"]:
        base.append("--spikes")

    def launch(agents, secs, seed):
        p = subprocess.Popen(base + ["--agents", str(agents), "--seconds", str(secs), "--seed", str(seed)],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT, env=env, text=True)
        try:   # (the child inherited this rank's NUMA share: give it every core back — from here, not in a preexec_fn)
            os.sched_setaffinity(p.pid, all_cpus)
        except OSError:
            pass
        return p

    def collect(procs):
        rates, longest, cpu_frac = [], 0.0, []
        for p in procs:
            out, _ = p.communicate()
            try:
                parts = out.split()
                steps, el = float(parts[0]), float(parts[1])
                rates.append(steps / el)
                if len(parts) > 2:
                    cpu_frac.append(float(parts[2]) / el)
                longest = max(longest, el)
            except (ValueError, IndexError):
                pass
        return rates, longest, cpu_frac

    t0 = time.perf_counter()
    one_rates, one_el, _ = collect([launch(256, min(seconds, 4.0), 0)])
    one_per, _e, _ = collect([launch(per, min(seconds, 3.0), 0)])     # one worker of the fleet's size, alone on the box
    rates, all_el, cpu_frac = collect([launch(per, seconds, 1 + i) for i in range(cores)])
    wall = time.perf_counter() - t0
    ok = len(rates)
    rs, cf = sorted(rates), sorted(cpu_frac)
    med = lambda v: v[len(v) // 2] if v else None   # noqa: E731
    alone = one_per[0] if one_per else None
    total = sum(rates)
    return {"value": total, "unit": "agent-steps/s", "cores": ok, "kind": "port",
            "cpu_model": cpu_model(),
            "sample": f"{ok} single-threaded worker processes (one per usable host core) x {per} agents x ~{all_el:.1f} s of "
                      f"Agent.update + PlaceCells({n}).update, float64 NumPy oracle; {wall:.1f} s wall in total",
            "host": host,
            "workers": {"rate_min": rs[0] if rs else None, "rate_median": med(rs), "rate_max": rs[-1] if rs else None,
                        "one_such_worker_alone": alone,
                        "cores_worth": round(total / alone, 1) if alone else None,
                        "cpu_seconds_per_wall_second_median": round(med(cf), 3) if cf else None,
                        "cpu_seconds_per_wall_second_sum": round(sum(cf), 1) if cf else None,
                        "note": "cores_worth = the fleet's rate / one such worker alone on the box: how many cores the host "
                                "actually delivered (shared hosts, memory bandwidth, SMT siblings); "
                                "cpu_seconds_per_wall_second = what the scheduler gave a worker"},
            "one_core": {"value": one_rates[0] if one_rates else None, "agents": 256, "seconds": round(one_el, 2)}}


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def _format_cpulist(cpus):
    cpus, out, i = sorted(cpus), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def gpu_numa_nodes(sysfs="/sys"):
    """[(numa node, [cpus of that node], pci address)] of the GPUs in the order the HIP runtime numbers them, from sysfs
    only (no HIP call): the KFD topology lists the GPU nodes in the runtime's order and names each one's DRM render
    node, whose PCI device carries `numa_node` / `local_cpulist`; ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES (integer lists)
    are applied the way the runtime applies them."""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    gpus = []
    for node in sorted((d for d in os.listdir(base) if d.isdigit()), key=int):
        props = {}
        with open(os.path.join(base, node, "properties")) as f:
            for line in f:
                k, _, v = line.strip().partition(" ")
                props[k] = v
        if int(props.get("simd_count", "0")) <= 0:
            continue  # a CPU node
        dev = os.path.join(sysfs, f"class/drm/renderD{int(props['drm_render_minor'])}/device")
        with open(os.path.join(dev, "numa_node")) as f:
            numa = int(f.read().strip())
        with open(os.path.join(dev, "local_cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        gpus.append((numa, cpus, os.path.basename(os.path.realpath(dev))))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        sel = os.environ.get(var)
        if sel is None or (var == "CUDA_VISIBLE_DEVICES" and "HIP_VISIBLE_DEVICES" in os.environ):
            continue
        idx = [int(x) for x in sel.split(",") if x.strip() != ""]   # (UUID lists raise: the caller reports "unbound")
        gpus = [gpus[i] for i in idx]
    return gpus


def bind_rank_to_gpu_numa(local, n_local, sysfs="/sys", apply=True):
    """Pin this process — before its first HIP call, so that the runtime's own threads inherit the mask — to cores of
    the NUMA node its GPU hangs off.  Every rank spins on the host for a fifth of a 90 us region and `value` takes the
    slowest rank of every repeat: a rank scheduled on the far socket, or two ranks sharing a core, would set the
    8-GPU number.  Ranks whose GPUs share a node get disjoint, equal sets of whole cores (all hardware threads of a
    core go to the same rank).  Returns what was done, for the bench line's `config`."""
    try:
        gpus = gpu_numa_nodes(sysfs)
        if local >= len(gpus):
            return {"binding": "none", "why": f"{len(gpus)} GPUs found in sysfs, local rank {local}"}
        numa, cpus, pci = gpus[local]
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus if c in allowed]
        if numa < 0 or not cpus:
            return {"binding": "none", "why": f"GPU {pci}: numa_node {numa}, {len(cpus)} usable cpus", "gpu_pci": pci}
        mates = [r for r in range(min(n_local, len(gpus))) if gpus[r][0] == numa]
        cores = {}
        for c in cpus:  # hardware threads grouped into cores
            try:
                with open(os.path.join(sysfs, f"devices/system/cpu/cpu{c}/topology/thread_siblings_list")) as f:
                    key = tuple(x for x in _parse_cpulist(f.read()) if x in allowed)
            except OSError:
                key = (c,)
            cores.setdefault(key or (c,), None)
        cores = sorted(cores)
        per = len(cores) // len(mates)
        if per >= 1:
            k = mates.index(local)
            mine = sorted(c for core in cores[k * per:(k + 1) * per] for c in core)
        else:
            mine = cpus
        if apply:
            os.sched_setaffinity(0, mine)
        return {"binding": "numa", "numa_node": numa, "gpu_pci": pci, "cpus": _format_cpulist(mine), "n_cpus": len(mine),
                "ranks_on_this_node": len(mates)}
    except Exception as e:  # noqa: BLE001  (sysfs layout, permissions, UUID device lists: run unbound rather than not at all)
        return {"binding": "none", "why": f"{type(e).__name__}: {e}"}


def relaunch_as_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def measure(args, config, K, W, repeats, rank, world, local, dist, ctrl_on_cpu, control_plane, store_ceiling=True):
    """One configuration: build the world, W warm-up steps, `repeats` timed K-step regions; the contract's fields
    (rank 0; None elsewhere) and the configuration."""
    import numpy as np
    import torch
    import ratinabox_amd as riab
    cfg = resolve(dict(CONFIGS[config]))
    if args.strong and world > 1:   # (a fixed batch over more GPUs is launch-bound; labelled "strong")
        per = cfg["agents"] // world // 256 * 256
        if per <= 0:
            print(f"[bench] --strong: {cfg['agents']} agents do not split into whole 256-agent groups over {world} ranks",
                  file=sys.stderr)
            sys.exit(2)
        cfg["agents"] = per
    args.plan = args.plan or args.task
    one_world = bool(getattr(args, "task_world", False))
    env, ag, pops = build_world(riab, cfg, rank, task=args.task, one_world=one_world)
    # the full rate history of K steps must fit in HBM next to the warmup's; otherwise stream
    # through ring buffers (every byte is still written, the oldest rows are overwritten)
    n_cells = sum(int(p.n) for p in pops)
    need = (K + W) * cfg["agents"] * (n_cells * (5 if cfg["spikes"] else 4) + 32)
    free_b, _total_b = torch.cuda.mem_get_info()
    if need > 0.6 * free_b:
        args.no_history = True
    if args.no_history:
        ag.save_history = False
        for p in pops:
            p.save_history = False
    if getattr(args, "strict", False):
        ag.pipeline_mode(strict=True)
    B = cfg["agents"]
    R = repeats or max(3, min(20, 4096 // max(K, 1)))
    # the chunked two-stream path (several populations): about four chunks in flight for short runs so that the
    # two pipeline stages still overlap (below ~64 steps a single launch of each stage is faster)
    chunk = args.chunk
    if K < 4 * chunk:
        chunk = K if K <= 64 else max(32, (K // 4 + 3) // 4 * 4)

    plan = {"p": None}
    native = not (args.per_step or args.plan) and os.environ.get("RIAB_NO_NATIVE") != "1"
    # one store-bound population: the one-kernel form of riab_simulate's rate stage (its waves wait for their rows)
    fused_mode = (native and len(pops) == 1 and pops[0]._stream_kind is not None and pops[0].noise_std == 0
                  and ag._Bp % 256 == 0 and os.environ.get("RIAB_NO_FUSED") != "1")
    # anything else: the chunk form (every chunk of rows behind a gate, each population's ordinary kernel; one native
    # call per timed region); the events then bracket every launch of the DOMINANT population's kernel
    native_mode = native and not fused_mode
    # the one-kernel form is timed by the device's own clock (first wave's start / last wave's end, no host cost);
    # --event-timing attaches HIP start / stop events to the launch instead (~7 us of host time per region)
    ag._time_rate_kernel = ("events" if args.event_timing else True) if (fused_mode or native_mode) else False

    def prepare():
        # (recording the plan is set-up, like building the world: outside the timed region — the history reset before every
        # repeat closes the previous plan)
        if args.plan and (plan["p"] is None or ag._plan is not plan["p"]):
            if args.task:
                plan["p"] = env.make_step_plan(capacity=max(K, W), auto_reset=True, scripted_speed=11 * ag.speed_mean)
            else:
                plan["p"] = ag.make_step_plan(capacity=max(K, W))

    def run(n_steps):
        if args.plan:
            prepare()
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
            ag.simulate(n_steps, chunk=chunk)

    def barrier():
        if dist is not None:
            dist.barrier()

    def fresh_history(n):
        ag.reset_history()
        for p in pops:
            p.reset_history()
        ag.preallocate_history(n)  # output buffers are allocated outside the timed region

    # the dominant kernel: BoundaryVectorCells where there are any (65-70 % of the kernel time of cfg 3 / cfg 5,
    # profiles/r03_cfg3_kernel_stats.csv), else the first population
    dominant = next((p for p in pops if type(p).__name__ == "BoundaryVectorCells"), pops[0])
    ag._timed_population = dominant
    # ---- warmup (untimed); its history is dropped so the timed runs own fresh HBM
    if W > 0:
        run(W)
    torch.cuda.synchronize()
    warm_ms = ag.last_rate_kernel_ms() if (fused_mode and W > 0) else None

    # ---- HIP-event timing of the dominant kernel's launches, on the stream it runs on.
    # flag-coupled path: the native call records the events around its rate kernel (Agent.last_rate_kernel_ms).
    # chunked path: torch events around the first population's launches (every launch when there are few; every
    # second one otherwise: each record is a packet on a stream that runs back to back, ~3 % of the value).
    spans = []
    seen = {"n": 0}
    n_launches = (K + chunk - 1) // max(chunk, 1)


    def hook(pop, what, tc):
        if pop is not dominant:
            return
        if what == "begin":
            seen["n"] += 1
        if n_launches > 4 and seen["n"] % 2 == 1:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if what == "begin":
            spans.append([ev, None, tc])
        else:
            spans[-1][1] = ev

    if not (args.per_step or args.plan or fused_mode or native_mode):
        ag._profile_hook = hook

    elapsed, kernel_ms, kernel_units = [], [], []
    host_split = []   # per repeat (us): Python in front of the native call, inside it, from its return to the synchronised end
    for _r in range(R):
        fresh_history(K)
        prepare()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        ag._host_clock = [] if fused_mode else None
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        t1 = time.perf_counter()  # this rank's K steps are done; the MAX over ranks is taken per repeat below,
        barrier()                 # so the closing barrier (an RCCL all-reduce, tens of us) stays outside the interval
        elapsed.append(t1 - t0)
        hc, ag._host_clock = ag._host_clock, None
        if hc and len(hc) == 1:   # (the short road of simulate(): one native call per region)
            host_split.append(((hc[0][0] - t0) * 1e6, (hc[0][1] - hc[0][0]) * 1e6, (t1 - hc[0][1]) * 1e6))
        if fused_mode or native_mode:
            kernel_ms.append(ag.last_rate_kernel_ms())
            kernel_units.append(getattr(ag, "_last_fused_units", B * K))  # (rings: the last ring-length piece of the run)
    # the closed-loop plans once more with ALL K steps in one native call per region (`plan.step(K)`: no per-step action
    # from the host, the step kernels back to back at the device's pace) — not part of `value`: how far the one-call-per-
    # step figure above is from the kernels themselves
    one_call_us = None
    if args.plan and not args.per_step and rank == 0 and K > 1:
        oc = []
        for _r in range(5):
            fresh_history(K)
            prepare()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan["p"].step(K)
            torch.cuda.synchronize()
            oc.append((time.perf_counter() - t0) / K * 1e6)
        one_call_us = round(sorted(oc)[len(oc) // 2], 3)
    # cross-check of the device-clock timing of the one-kernel rate stage: a few more regions — not part of `value` —
    # with HIP start / stop events attached to the kernel's launch (what rocprofv3 would report for it)
    event_ms = []
    one_kernel = fused_mode and ag.last_rate_stage_form() == "one-kernel"   # (what the library chose: riab_streamer_last_form)
    if one_kernel and ag._time_rate_kernel is True:
        ag._time_rate_kernel = "events"
        for _r in range(10):
            fresh_history(K)
            torch.cuda.synchronize()
            run(K)
            torch.cuda.synchronize()
            m = ag.last_rate_kernel_ms()
            if m is not None:
                event_ms.append(m)
        ag._time_rate_kernel = True
    el = torch.tensor(elapsed, dtype=torch.float64)
    per_rank = None
    med3 = None
    if host_split:
        cols = list(zip(*host_split))
        med3 = [round(sorted(c)[len(c) // 2], 2) for c in cols]
    my_diag = dict(ag.diagnostics)
    if dist is not None:
        el = el.to("cpu" if ctrl_on_cpu else "cuda")
        # every rank's own regions (min / median / max): `value` is set by the slowest rank of every repeat, so a slow
        # HOST among the ranks must be visible in the line
        parts = [torch.zeros_like(el) for _ in range(world)]
        dist.all_gather(parts, el)
        per_rank = []
        for r, x in enumerate(parts):
            xs = sorted(x.cpu().tolist())
            per_rank.append({"rank": r, "min": round(xs[0] * 1e3, 5), "median": round(xs[len(xs) // 2] * 1e3, 5),
                             "max": round(xs[-1] * 1e3, 5)})
        # ... and where each rank's HOST spent its region (a slow rank is then attributable: a late Python, a slow
        # launch path, or a late wake-up after the kernels) + its own pipeline diagnostics
        extra = [None] * world
        dist.all_gather_object(extra, {"host_us": med3, "diagnostics": my_diag})
        for r, x in enumerate(extra):
            if x["host_us"]:
                per_rank[r]["host_us"] = dict(zip(("python_before_native_call", "in_native_call", "call_return_to_synchronised"),
                                                  x["host_us"]))
            per_rank[r]["pipeline_serialised"] = x["diagnostics"].get("pipeline_serialised")
            per_rank[r]["pipeline_timeouts"] = x["diagnostics"].get("pipeline_timeouts")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)  # per repeat: the slowest rank
        el = el.cpu()
    el_sorted = sorted(el.tolist())
    med = el_sorted[len(el_sorted) // 2] if len(el_sorted) % 2 else 0.5 * (el_sorted[len(el_sorted) // 2 - 1] +
                                                                           el_sorted[len(el_sorted) // 2])

    total_units = world * B * K
    value = total_units / med
    bpu = bytes_per_agent_step(cfg)

    roofline = None
    n0 = int(dominant.n)
    # the contract's `achieved` uses SURVEY §8(d)'s per-unit figure (rates written + spikes + the 112 B of state /
    # history that the trajectory kernel moves while the rate kernel runs), restricted to the dominant population;
    # the rate kernel ALONE moves 4*n0 (+n0) + 8 B per unit: `achieved_kernel_own_bytes`
    unit_bytes = 4 * n0 + (n0 if cfg["spikes"] else 0) + 112
    own_bytes = 4 * n0 + (n0 if cfg["spikes"] else 0) + 8
    ms, units = [], []
    if fused_mode or native_mode:
        units = [u for m, u in zip(kernel_ms, kernel_units) if m is not None]
        ms = [m for m in kernel_ms if m is not None]
    elif spans:
        ms = [a.elapsed_time(b) for a, b, _ in spans]
        units = [B * tc for _, _, tc in spans]
    # The one-kernel rate stage: `avg_launch_ms` / `frac` come from the HIP events (what the contract asks for and what a
    # profiler's begin / end of the dispatch agrees with); the device-clock figure of the timed regions — first wave's start
    # to last store acknowledged: no dispatch ramp, no write-back at the end of the kernel, 5-10 % shorter — goes beside it.
    clock_ms = None
    if one_kernel and event_ms and ms and ag._time_rate_kernel is True:
        clock_ms, ms = ms, list(event_ms)
        units = [units[0]] * len(ms)
    if ms:
        avg_ms = float(np.mean(ms))
        avg_units = float(np.mean(units))
        achieved = unit_bytes * avg_units / (avg_ms * 1e-3) / 1e9
        # HBM bytes from the PMC counters: NOT collected in this run — the per-unit figure of the kernel that was timed,
        # from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, keyed by kernel), times the units
        traffic = traffic_from = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pmc_kernel = None
        if type(dominant).__name__ == "PlaceCells" and not cfg["spikes"]:
            pmc_kernel = ("rate_kernel_gated" if K <= 256 else "rate_kernel_gated_long") if one_kernel else "rate_kernel_wide"
        if pmc_kernel and os.path.exists(tpath):
            with open(tpath) as f:
                entry = json.load(f).get("kernels", {}).get(pmc_kernel)
            if entry:
                traffic = round(entry["hbm_bytes_per_unit"] * avg_units)
                traffic_from = (f"profiles/pmc_traffic.json[{pmc_kernel}]: {entry['hbm_bytes_per_unit']:.1f} B per agent-step "
                                f"(rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes, {entry['source'].split(': ')[-1]}) x units "
                                "of this run's launches; counters are not collected inside bench.py")
        kname = (("rate_kernel_gated" if one_kernel else
                  ("rate stage = rate_kernel_gated (256 rows) + rate_kernel_wide (512 rows per launch behind progress gates)"
                   if ag.last_rate_stage_form() == "head+pieces" else
                   "rate stage = rate_kernel_wide per chunk behind progress gates"))
                 if fused_mode else "rate_kernel_wide") + f"<{type(dominant).__name__}>"
        if native_mode:
            kname = f"every launch of {type(dominant).__name__}'s kernel in one riab_simulate call (chunks of rows behind gates)"
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_from": traffic_from,
                    "kernel": kname, "launches": len(ms),
                    "avg_launch_ms": round(avg_ms, 5), "min_launch_ms": round(float(np.min(ms)), 5),
                    "max_launch_ms": round(float(np.max(ms)), 5), "units_per_launch": int(avg_units),
                    "bytes_per_unit": unit_bytes, "bytes_per_unit_is": "SURVEY.md 8(d): 4*n (+n spikes) + 112",
                    "achieved_kernel_own_bytes": round(own_bytes * avg_units / (avg_ms * 1e-3) / 1e9, 1),
                    "kernel_own_bytes_per_unit": own_bytes,
                    "frac_of_measured_copy_bw_6290": round(achieved / 6290.0, 4)}
        if type(dominant).__name__ == "BoundaryVectorCells":
            # BVC is bound by transcendental issue, not by HBM: n*K exponentials per position, minus
            # the directions the cells' windows leave out.  achieved = terms ISSUED per second; peak = the v_exp_f32
            # issue ceiling derived from the instruction's measured issue rate (independent of this kernel).
            terms = float(n0) * float(dominant.n_test_angles)
            issued = float(getattr(dominant, "_window_stats", {}).get("issued", 1.0)) if hasattr(dominant, "_window_stats") else 1.0
            t_full = terms * avg_units / (avg_ms * 1e-3) / 1e12
            t_ach = t_full * issued
            roofline = {"bound": "valu", "achieved": round(t_ach, 3), "peak": round(VALU_EXP_PEAK_TTERMS, 2), "unit": "Tterm/s",
                        "frac": round(t_ach / VALU_EXP_PEAK_TTERMS, 4), "traffic": None,
                        "kernel": "bvc_kernel", "launches": len(ms), "avg_launch_ms": round(avg_ms, 5),
                        "units_per_launch": int(avg_units), "terms_per_unit": int(terms),
                        "issued_fraction_of_terms": round(issued, 4), "full_sum_terms_per_s_T": round(t_full, 3),
                        "peak_is": "v_exp_f32 issue ceiling: 1024 SIMDs x 64 lanes x 2.4 GHz / 9.44 cycles per "
                                   "wave-instruction (measured for the single instruction, tools/exp_bench.hip); one term = "
                                   "one fused exponent exp2(-(a d - a mu)^2 + T[c][k])",
                        "instruction_mix_ceiling_Tterm_s": {"serial_sum_of_measured_issue_cycles": round(VALU_MIX_SERIAL_TTERMS, 2),
                                                            "measured_register_only_loop": VALU_MIX_MEASURED_TTERMS,
                                                            "mix": "per term 1 v_exp_f32 + 1 v_pk_fma_f32 + 1/2 v_pk_add_f32"},
                        "frac_of_measured_mix_ceiling": round(t_ach / VALU_MIX_MEASURED_TTERMS, 4),
                        "hbm_GBps_of_this_kernel": round(achieved, 1), "hbm_frac_of_this_kernel": round(achieved / HBM_PEAK_GBS, 4)}
        if native_mode:
            roofline["note"] = ("`avg_launch_ms` is the SUM of the kernel's launches of one timed region (HIP events around each "
                                "launch on the stream it runs on, riab_streamer_last_rate_ms), `units_per_launch` the agent-"
                                "steps of the region; rocprofv3's per-launch average x launches per region is the same sum")
        if one_kernel:
            if ag._time_rate_kernel == "events":
                roofline["timed_by"] = "HIP start / stop events attached to the kernel's launch (hipExtLaunchKernel) in the timed regions"
            elif clock_ms is not None:
                roofline["timed_by"] = (f"HIP start / stop events attached to the kernel's launch (hipExtLaunchKernel) in {len(ms)} "
                                        "regions run right after the timed ones (same call, same sizes, fresh rows): inside "
                                        "the timed regions the two events would cost 7 us of host time in front of the "
                                        "dispatch.  `avg_launch_ms_device_clock`: the kernel in the TIMED regions, from the "
                                        "device's constant clock (s_memrealtime) read by its first-row waves at their start "
                                        "and its last-row waves once their stores are acknowledged")
                cm = float(np.mean(clock_ms))
                roofline["avg_launch_ms_device_clock"] = round(cm, 5)
                roofline["min_launch_ms_device_clock"] = round(float(np.min(clock_ms)), 5)
                roofline["max_launch_ms_device_clock"] = round(float(np.max(clock_ms)), 5)
                roofline["frac_device_clock"] = round(unit_bytes * avg_units / (cm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            else:
                roofline["timed_by"] = ("the device's constant clock (s_memrealtime) read by the kernel's first-row waves at "
                                        "their start and its last-row waves once their stores are acknowledged")
        if fused_mode:
            roofline["note"] = ("the rate stage runs concurrently with the trajectory kernel whose rows it consumes (coupled "
                                "by flags in device memory, one native call per timed region): its duration includes "
                                "waiting for rows; for one store-bound population it is ONE kernel (every wave waits for "
                                "its rows), otherwise every population's kernel per chunk of rows behind a one-wave gate")
            roofline["rate_stage_form"] = ag.last_rate_stage_form()
            if warm_ms is not None:  # (rocprofv3 --stats averages over ALL launches of the process, warm-up included)
                roofline["warmup_launch_ms"] = round(warm_ms, 5)

    plan_info = None
    if args.plan and plan["p"] is not None:
        plan_info = plan["p"].info()
    elif args.per_step and ag._plan is not None and hasattr(ag._plan, "info"):
        plan_info = ag._plan.info()
    if roofline is None and (args.plan or args.per_step):
        # the closed-loop paths: kernels follow each other on one stream, so a step's time on the host clock IS the
        # time of its kernel(s) + the gap between them; no per-kernel events (they would sit between the launches)
        step_s = med / K
        ach = unit_bytes * B / step_s / 1e9
        one = bool(plan_info and plan_info["fused_steps"] > 0)
        traffic = traffic_from = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if one and type(dominant).__name__ == "PlaceCells" and not cfg["spikes"] and os.path.exists(tpath):
            tkey = "step1_task_kernel" if args.task else "step1_kernel"
            with open(tpath) as f:
                entry = json.load(f).get("kernels", {}).get(tkey)
            if entry:   # (counters are not collected in this run: the committed PMC passes' per-unit figure x this run's units)
                traffic = round(entry["hbm_bytes_per_unit"] * B)
                traffic_from = (f"profiles/pmc_traffic.json[{tkey}]: {entry['hbm_bytes_per_unit']:.1f} B per agent-step "
                                f"(rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of `bench.py --{'task' if args.task else 'plan'}`) x "
                                "the units of one launch")
        roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_from": traffic_from,
                    "kernel": ("step1_task_kernel<%s> (Agent.update + the rest of TaskEnvironment.step + auto-reset + next action + "
                               "Neurons.update in one launch)" % type(dominant).__name__) if one and args.task
                    else ("step1_kernel<%s> (Agent.update + Neurons.update in one launch)" % type(dominant).__name__) if one
                    else ("motion_world_kernel (Agent.update + the shared world's TaskEnvironment.step) + task_world_reset_kernel "
                          "(device-decided reset + next action) + rate_kernel_wide<%s> per step" % type(dominant).__name__) if one_world
                    else ("motion_task_kernel + rate_kernel_wide<%s> per step" % type(dominant).__name__) if args.task
                    else "agent_step_kernel + rate_kernel_wide<%s> per step" % type(dominant).__name__,
                    "launches": None, "avg_launch_ms": round(step_s * 1e3, 6), "units_per_launch": B,
                    "bytes_per_unit": unit_bytes, "bytes_per_unit_is": "SURVEY.md 8(d): 4*n (+n spikes) + 112",
                    "timed_by": "timed region / steps (host clock around K back-to-back steps): the step's kernel(s) plus the "
                                "inter-kernel gap; rocprofv3's per-kernel average is in profiles/"}

    # the chip's measured store ceiling in this same process (riab_fill: one float4 per thread,
    # address-ordered), for context next to the spec peak
    if rank == 0 and roofline is not None and store_ceiling:
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
        ceiling = 5 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
        roofline["measured_store_ceiling_GBps"] = round(ceiling, 1)
        if roofline["bound"] == "hbm":
            roofline["frac_of_measured_store_ceiling"] = round(roofline["achieved"] / ceiling, 4)
        del buf

    out = None
    if rank == 0:
        api = ("TaskEnvironment(lanes='agents') step plan — the rank's agents in ONE world, agentmode='interact': Agent.update, "
               "shared goal list / rewards, device-decided reset, next action, rates; one native call per step" if one_world else
               "TaskEnvironment step plan: action, Agent.update, rewards/goals, auto-reset, rates; one native call per step"
               if args.task else "step plan (one native call per step)" if args.plan else "per-step update()"
               if args.per_step else "simulate(): trajectory kernel + rate stage running concurrently, coupled by flags in device memory, one native call"
               if fused_mode else "simulate(): trajectory kernel + every population's kernels per chunk of rows behind gates, "
               "one native call (riab_simulate)" if native_mode
               else f"simulate(): chunked two-stream pipeline, {chunk} steps/launch")
        if getattr(args, "strict", False) and native:
            api += " — STRICT mode (riab_hip.h 'Two modes': nothing allocated / synchronised / queried / process-wide; the started gate always)"
        out = {
            "metric": metric_name(cfg),
            "value": round(value, 1), "unit": "agent-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(med / K * 1e3, 6), "higher_is_better": True,
            "scaling": "strong" if (args.strong and world > 1) else "weak",
            "vs_baseline": None, "dtype": "f32 rates / f64 motion", "data": "synthetic",
            "config": {"workload": config + ": " + cfg["desc"], "agents_per_gpu": B,
                       "cells": {k: cfg[k] for k in ("place", "grid", "bvc", "hdc")},
                       "parallelism": f"agent-sharded x{world}, no step-path collective", "control_plane": control_plane,
                       "host_binding": getattr(args, "host_binding", None),
                       "api": api, "history": "ring" if args.no_history else "full", "spikes": cfg["spikes"],
                       "bytes_per_agent_step": bpu},
            "repeats": R,
            "timed_region_ms": {"median": round(med * 1e3, 5), "min": round(el_sorted[0] * 1e3, 5),
                                "max": round(el_sorted[-1] * 1e3, 5), "first": round(float(el[0]) * 1e3, 5),
                                "note": "every repeat runs the full K steps into fresh history rows; value = total "
                                        "agent-steps of one repeat / the median repeat (max over ranks per repeat)"},
            "timed_region_ms_per_rank": per_rank,
            "timed_region_host_us": (dict(zip(("python_before_native_call", "in_native_call", "call_return_to_synchronised"), med3))
                                     if med3 else None),
            "value_best_repeat": round(total_units / el_sorted[0], 1),
            "hbm_GBps_whole_path": round(value / world * bpu / 1e9, 1),
            "frac_whole_path": round(value / world * bpu / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roofline,
        }
        diag = ag.diagnostics
        out["pipeline"] = ag.pipeline_info()
        if plan_info is not None:
            # (a plan lives for one region: the history reset in front of every repeat closes it — these are the counters of
            # the LAST region's plan, which served K steps)
            out["plan"] = dict(plan_info, launches_per_step=round(plan_info["launches"] / max(1, K), 3) if args.plan else None,
                               us_per_step_all_steps_in_one_native_call=one_call_us)
        if args.task:
            diag = dict(diag, **env.diagnostics, episodes_finished=len(env.episodes["episode"]))
        out["diagnostics"] = diag
    del ag, pops, env
    import gc
    gc.collect()                # (Agent <-> Neurons <-> history views are reference cycles: tens of GB of rows per configuration)
    torch.cuda.empty_cache()
    return out, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--chunk", type=int, default=128, help="steps per kernel launch in the chunked two-stream path")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the timed K-step region (0 = by K)")
    ap.add_argument("--per-step", action="store_true", help="time the drop-in per-step API instead of simulate()")
    ap.add_argument("--plan", action="store_true", help="time the closed-loop path through a native step plan")
    ap.add_argument("--plan-batch", type=int, default=1, help="steps per riab_plan_step call (1 = closed loop)")
    ap.add_argument("--task", action="store_true",
                    help="closed loop through the batched TaskEnvironment: goal-seeking actions, rewards, goal checks "
                         "and per-lane auto-reset every step (implies --plan)")
    ap.add_argument("--task-world", action="store_true",
                    help="as --task, but each rank's agents share ONE task world (TaskEnvironment(lanes='agents'): the "
                         "reference's multi-agent environment with agentmode='interact')")
    ap.add_argument("--strict", action="store_true",
                    help="simulate() in the mode of riab_simulate that meets SURVEY 8(b2) to the letter (include/riab_hip.h 'Two modes')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the config's agents are the TOTAL, split over the ranks (default: weak scaling, "
                         "the config's agents per GPU)")
    ap.add_argument("--no-history", action="store_true", help="ring buffers instead of a full T-long history")
    ap.add_argument("--force-process-group", action="store_true",
                    help="bring the process group (RCCL control plane: barriers + the max-reduce of the timings) up even "
                         "for one rank, when launched by torch.distributed.run (a one-GPU test of the multi-GPU launch form)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short cfg3 / cfg4 / cfg5 runs reported in the `secondary` block of the cfg2 line")
    ap.add_argument("--secondary-timeout", type=float, default=240.0,
                    help="seconds after which the secondary block is abandoned and the headline line printed without it")
    ap.add_argument("--no-bind", action="store_true",
                    help="do not pin the rank to cores of its GPU's NUMA node (default: pinned before the first HIP call)")
    ap.add_argument("--event-timing", action="store_true",
                    help="time the one-kernel rate stage with HIP start / stop events on its launch instead of the "
                         "device clock stamps")
    args = ap.parse_args()
    args.task = args.task or args.task_world

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch_as_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line for the wrong rank count",
              file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    share = os.environ.get("RIAB_BENCH_SHARE_GPU") == "1"  # test hook: all ranks on cuda:0, gloo control plane
    # ---- host placement, before the first HIP call: cores of the GPU's NUMA node, disjoint between ranks -----------------
    global _ALL_CPUS
    _ALL_CPUS = os.sched_getaffinity(0)
    n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if args.no_bind:
        args.host_binding = {"binding": "none", "why": "--no-bind"}
    elif share:   # (every rank on GPU 0: its node's cores, split between the ranks as if each had a GPU of that node)
        args.host_binding = bind_rank_to_gpu_numa(0, 1)
        if args.host_binding.get("binding") == "numa" and n_local > 1:
            cpus = _parse_cpulist(args.host_binding["cpus"])
            per = max(1, len(cpus) // n_local)
            mine = cpus[local * per:(local + 1) * per] or cpus
            try:
                os.sched_setaffinity(0, mine)
                args.host_binding.update(cpus=_format_cpulist(mine), n_cpus=len(mine), ranks_on_this_node=n_local)
            except OSError as e:
                args.host_binding = {"binding": "none", "why": f"{type(e).__name__}: {e}"}
    else:
        args.host_binding = bind_rank_to_gpu_numa(local, n_local)
    if share:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    control_plane = "none"
    if world > 1 or (args.force_process_group and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ctrl_on_cpu = share
        if share:
            dist.init_process_group("gloo")
            control_plane = "gloo"
        else:
            # The step path has no collective: the process group only carries the barriers and the max-reduce of the
            # timings.  RCCL first; if it cannot be brought up on this node, the same control plane over gloo (every
            # rank fails the same way, so every rank falls back) rather than no line at all.
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                control_plane = "nccl"
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: RCCL control plane failed ({type(e).__name__}: {e}); using gloo", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                dist.init_process_group("gloo")
                ctrl_on_cpu = True
                control_plane = "gloo"

    out, cfg = measure(args, args.config, args.steps, args.warmup, args.repeats, rank, world, local, dist, ctrl_on_cpu if dist
                       is not None else False, control_plane)
    if dist is not None:   # every rank's placement in rank 0's line
        allb = [None] * world
        dist.all_gather_object(allb, args.host_binding)
        if out is not None:
            out["config"]["host_binding_per_rank"] = allb
    # ---- the other BASELINE configurations under the same clock: one block in the same line -----------------------------
    # cfg2 again at the length SURVEY 8(d) quotes it on (1000 steps after 100 warm-up: here 1024 / 128), cfg3 (a one-GPU
    # configuration: only at N = 1), and the per-GPU shards of the two 8-GPU configurations cfg4 / cfg5 — at every N,
    # so that a multi-rank line carries BASELINE configs[3] and [4] as they are defined
    secondary = {}
    # A secondary run must never cost the headline line: every rank arms a watchdog that — should the block not finish
    # (one rank failing alone leaves the others in a barrier) — lets rank 0 print the line with what it has and ends the
    # process with the launcher's success code.
    import threading
    state = {"running": None, "done": False}
    line_lock = threading.Lock()   # (the line is printed exactly once: by the watchdog or by the main thread)

    def bail():
        with line_lock:
            if state["done"]:
                return
            state["done"] = True
            if rank == 0 and out is not None:
                if secondary:
                    out["secondary"] = dict(secondary)
                out["secondary_error"] = f"watchdog: the secondary run '{state['running']}' did not finish in {args.secondary_timeout} s"
                print(json.dumps(out), flush=True)
        # (the headline line is out; a rank stuck in a collective cannot be torn down cleanly.  Exit code 0 keeps the
        # launcher from discarding that line; the line itself carries `secondary_error`)
        os._exit(0)

    dog = threading.Timer(args.secondary_timeout, bail)
    dog.daemon = True
    sec_steps = 256 if share else SECONDARY_STEPS   # (the test hook: N ranks' histories on ONE GPU)
    if args.config == "cfg2" and not (args.no_secondary or args.per_step or args.plan or args.task or args.strong):
        dog.start()
        saved = args.no_history
        # ... and the CLOSED-LOOP forms of cfg2 (the reference's per-step API, Agent.py:160-242 + Neurons.py:145-171, and
        # its TaskEnvironment.step, contribs/TaskEnvironment.py:361-453): an explicit step plan (one native call per
        # step), the unchanged `Ag.update(); PCs.update()` loop, a task plan (every agent its own replica of the task; all
        # agents of the rank in ONE task world) — 256 steps after 32, five repeats each
        # ... the closed loops of cfg 3 / cfg 5 as well (the loop of the reference's tests/test_advanced.py:159-176: every
        # population updated after every agent step; the store-bound populations ride in the agent step's launch, the
        # boundary vector cells follow: two kernels per step), the headline workload in STRICT mode (what meeting SURVEY
        # 8(b2) to the letter costs: at the driver's length and at 1024 steps), and cfg 3 in a room of 64 walls
        runs = [("cfg2_T1024", "cfg2", 128, None)] + ([("cfg3", "cfg3", 32, None)] if world == 1 else []) + \
            [("cfg4", "cfg4", 32, None), ("cfg5", "cfg5", 32, None),
             ("cfg2_closed_loop_plan", "cfg2", 32, "plan"), ("cfg2_closed_loop_per_step", "cfg2", 32, "per_step"),
             ("cfg2_closed_loop_task", "cfg2", 32, "task"), ("cfg2_closed_loop_task_world", "cfg2", 32, "task_world")] + \
            ([("cfg3_closed_loop_plan", "cfg3", 32, "plan"), ("cfg5_closed_loop_plan", "cfg5", 32, "plan"),
              ("cfg2_strict", "cfg2", args.warmup, "strict"), ("cfg2_strict_T1024", "cfg2", 128, "strict1024"),
              ("cfg3_64w", "cfg3_64w", 32, None), ("cfg3_64w_closed_loop_plan", "cfg3_64w", 32, "plan")] if world == 1 else [])
        for key, name, warm, mode in runs:
            if key == "cfg2_T1024" and args.steps == SECONDARY_STEPS:
                continue   # (the headline run IS that run)
            args.no_history = False
            args.plan, args.per_step = mode in ("plan", "task", "task_world"), mode == "per_step"
            args.task, args.task_world = mode in ("task", "task_world"), mode == "task_world"
            args.strict = mode in ("strict", "strict1024")
            state["running"] = key
            t0 = time.perf_counter()
            try:
                steps_here = args.steps if mode == "strict" else 256 if mode in ("plan", "per_step", "task", "task_world") else sec_steps
                o, c = measure(args, name, steps_here, warm, 0 if mode == "strict" else 5, rank, world, local, dist,
                               ctrl_on_cpu if dist is not None else False, control_plane, store_ceiling=False)
            except Exception as e:  # noqa: BLE001  (the headline line must not be lost to a secondary run)
                secondary[key] = {"error": f"{type(e).__name__}: {e}"}
                continue
            finally:
                args.plan = args.per_step = args.task = args.task_world = args.strict = False
            torch.cuda.empty_cache()   # (tens of GB of history per configuration: give them back before the next one)
            if o is None:
                continue
            r = o["roofline"] or {}
            secondary[key] = {"workload": o["config"]["workload"], "api": o["config"]["api"], "plan": o.get("plan"),
                              "value": o["value"], "unit": o["unit"], "n_gpus": world,
                              "steps": o["steps"], "warmup": warm, "repeats": o["repeats"], "ms_per_step": o["ms_per_step"],
                              "timed_region_ms": o["timed_region_ms"]["median"],
                              "timed_region_ms_per_rank": o["timed_region_ms_per_rank"],
                              "bytes_per_agent_step": o["config"]["bytes_per_agent_step"],
                              "hbm_GBps_whole_path": o["hbm_GBps_whole_path"], "frac_whole_path": o["frac_whole_path"],
                              "roofline": {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel",
                                                                 "launches", "avg_launch_ms", "units_per_launch",
                                                                 "avg_launch_ms_device_clock", "frac_device_clock",
                                                                 "rate_stage_form")},
                              "diagnostics": o.get("diagnostics"),
                              "wall_s": round(time.perf_counter() - t0, 2)}
        args.no_history = saved
    with line_lock:
        if state["done"]:      # (the watchdog has printed the line and is ending the process)
            return
        state["done"] = True
    dog.cancel()
    if rank == 0:
        if secondary:
            out["secondary"] = secondary
            t1024 = secondary.get("cfg2_T1024")
            if t1024 and "value" in t1024:
                # the same configuration at SURVEY 8(d)'s length, where 36 us of host time per region are 1 % instead of
                # 45 %: the figure the >= 7x-at-8-GPUs clause is ALSO read from
                out["value_T1024"] = t1024["value"]
                out["frac_whole_path_T1024"] = t1024.get("frac_whole_path")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Kernel-level microbenchmarks on one MI355X (HIP events on the launch stream).
    python tools/microbench.py [agent] [fill] [rates]
"""
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab  # noqa: E402

L = riab._lib


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


MAZE = [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]


def agent():
    T = 256
    for B in (4096, 32768):
        for walls, wname in (([], "open"), (MAZE, "maze")):
            for prec in (64,):
                for mode in ("philox", "z_in"):
                    np.random.seed(0)
                    env = riab.Environment({"walls": walls})
                    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "save_history": True})
                    z = torch.randn((T, 2, B), dtype=torch.float64, device="cuda") if mode == "z_in" else None
                    hist = torch.empty((T, 8, B), dtype=torch.float32, device="cuda")

                    def run():
                        kw = {} if z is None else {"noise": z}
                        ag._advance(T, None, None, 1, kw, hist_view=hist)
                    ms = timeit(run)
                    print(f"agent_step B={B} {wname} f{prec} {mode}: {ms / T * 1e3:.2f} us/step  "
                          f"{B * T / ms / 1e6:.1f} M agent-steps/s", flush=True)


def agent_t():
    """Launch duration of the motion kernel vs steps per launch (B = 4096, open box, f64, Philox)."""
    B = 4096
    np.random.seed(0)
    ag = riab.Agent(riab.Environment(), {"n_agents": B, "dt": 0.01, "save_history": True})
    hist = torch.empty((256, 8, B), dtype=torch.float32, device="cuda")
    for T in (1, 4, 15, 16, 32, 64, 128, 256):
        ms = timeit(lambda: ag._advance(T, None, None, 1, {}, hist_view=hist[:T]))
        print(f"agent_step T={T}: {ms * 1e3:.1f} us/launch  {ms / T * 1e3:.2f} us/step", flush=True)


def hostio():
    """The host-buffer form of the boundary: get_state(pos=host array) -> host float64 array (upload of the
    positions, kernel, download + widening of the (n, P) rates) against the device-resident form."""
    np.random.seed(0)
    ag = riab.Agent(riab.Environment(), {"n_agents": 4, "dt": 0.01})
    pcs = riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean"})
    P = 4096
    pos = np.random.rand(P, 2)
    for _ in range(3):
        out = pcs.get_state(evaluate_at=None, pos=pos)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = pcs.get_state(evaluate_at=None, pos=pos)
    dt = (time.perf_counter() - t0) / reps
    print(f"get_state host->host, {P} positions x 1024 PlaceCells: {dt * 1e3:.2f} ms  {P / dt / 1e6:.2f} M positions/s "
          f"({out.nbytes / 2 / dt / 1e9:.2f} GB/s of fp32 rates over PCIe incl. fp64 widening on the host)", flush=True)


def fill():
    for gb in (0.25, 1, 4):
        n = int(gb * (1 << 30))
        buf = torch.empty(n, dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: L.check(L.lib.riab_fill(L.ptr(buf), n, 1.0, L.current_stream()), "fill"))
        print(f"riab_fill {gb} GiB: {ms:.3f} ms  {n / ms / 1e6:.0f} GB/s", flush=True)
        ms = timeit(lambda: buf.fill_(1))
        print(f"torch fill_ {gb} GiB: {ms:.3f} ms  {n / ms / 1e6:.0f} GB/s", flush=True)
    a = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
    b = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
    ms = timeit(lambda: b.copy_(a))
    print(f"torch copy 4 GiB -> 4 GiB: {ms:.3f} ms  {2 * 4 * (1 << 30) / ms / 1e6:.0f} GB/s (read+write)", flush=True)


def rates():
    B, T = 4096, 128
    np.random.seed(0)
    env = riab.Environment({"walls": MAZE})
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01})
    traj = ag.simulate(T)
    torch.cuda.synchronize()
    s = L.current_stream
    for name, pop, spikes in (
            ("PlaceCells 1024", riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean"}), False),
            ("PlaceCells 1024 +spikes", riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean"}), True),
            ("PlaceCells 4096", riab.PlaceCells(ag, {"n": 4096, "wall_geometry": "euclidean"}), False),
            ("PlaceCells 1024 line_of_sight", riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "line_of_sight"}), False),
            ("GridCells 1024", riab.GridCells(ag, {"n": 1024}), False),
            ("GridCells 1024 +spikes", riab.GridCells(ag, {"n": 1024}), True),
            ("HeadDirectionCells 256", riab.HeadDirectionCells(ag, {"n": 256}), False),
            ("BVC 256", riab.BoundaryVectorCells(ag, {"n": 256}), False),
            ("BVC 8 (ray stage dominated)", riab.BoundaryVectorCells(ag, {"n": 8}), False),
            ("BVC 1024", riab.BoundaryVectorCells(ag, {"n": 1024}), False),
            ("BVC 256 egocentric", riab.BoundaryVectorCells(ag, {"n": 256, "reference_frame": "egocentric"}), False)):
        n = pop.n
        Tt = T if "BVC" not in name else 16
        fr = torch.empty((Tt, n, B), dtype=torch.float32, device="cuda")
        sp = torch.empty((Tt, n, B), dtype=torch.uint8, device="cuda") if spikes else None
        out = dict(fr=fr, sp=sp, ring=None)
        ms = timeit(lambda: pop._rates_from_trajectory(traj[:Tt], out, 0, Tt, 0, 0.01, stream=s()))
        byt = Tt * B * n * (5 if spikes else 4)
        print(f"{name}: {ms:.3f} ms for {Tt}x{B} positions  {Tt * B / ms / 1e3:.1f} M pos/s  "
              f"{byt / ms / 1e6:.0f} GB/s written", flush=True)


def ff():
    """FeedForwardLayer GEMM on the fp32 matrix cores: PlaceCells(1024) -> n_out, 128 x 4096 positions."""
    B, T = 4096, 128
    np.random.seed(0)
    ag = riab.Agent(riab.Environment(), {"n_agents": B, "dt": 0.01})
    pcs = riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean", "save_spikes": False})
    ag.simulate(T)
    torch.cuda.synchronize()
    x = pcs.get_history_tensors()[0]  # [T, 1024, B]
    for n_out in (32, 128, 512):
        F = riab.FeedForwardLayer(ag, {"n": n_out, "input_layers": [pcs], "name": f"F{n_out}",
                                       "activation_function": {"activation": "relu"}})
        out = torch.empty((T, n_out, B), dtype=torch.float32, device="cuda")
        ms = timeit(lambda: F._gemm([x], T, B, out, None, L.current_stream()))
        fl = 2.0 * 1024 * n_out * T * B
        print(f"feedforward 1024 -> {n_out}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  "
              f"(reads {x.numel() * 4 / ms / 1e6:.0f} GB/s of rates)", flush=True)


def task():
    """Closed-loop TaskEnvironment steps/s: 4096 lanes, 1024 PlaceCells as observation, scripted
    goal-seeking policy, auto-reset; eager Python loop vs one native plan call per step."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    B, K = 4096, 400
    for mode in ("eager", "plan", "plan x16"):
        np.random.seed(0)
        env = SpatialGoalEnvironment(possible_goal_positions="random_8", goalcachekws=dict(reset_n_goals=2),
                                     teleport_on_reset=True, episode_terminate_delay=0.05)
        ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "save_history": False})
        pcs = riab.PlaceCells(ag, {"n": 1024, "save_history": False, "save_spikes": False})
        env.add_agents(ag)
        speed = 11 * ag.speed_mean
        if mode == "eager":
            def run(n):
                for _ in range(n):
                    a = env._goal_vector(speed)
                    obs, rew, term, _, _ = env.step(a)
                    env.reset(mask=term)
                    pcs.update()
        else:
            plan = env.make_step_plan(auto_reset=True, scripted_speed=speed)
            chunk = 16 if "x16" in mode else 1

            def run(n):
                for _ in range(n // chunk):
                    plan.step(chunk)
        run(64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"task loop [{mode}]: {dt / K * 1e6:.1f} us/step  {B * K / dt / 1e6:.1f} M agent-steps/s  "
              f"episodes finished {len(env.episodes['episode'])}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["agent", "fill", "rates"]
    for w in which:
        globals()[w]()

"""GPU tests of the flag-coupled pipeline (`riab_simulate`: the trajectory kernel publishing its rows to a
persistent firing-rate kernel that runs concurrently, csrc/riab_simulate.hip).

The pipeline runs the SAME arithmetic as the chunked two-stream path (`RIAB_NO_FUSED=1`), which the parity
tests pin against the reference goldens and the oracle; so the requirement here is bit-identity with that path —
every history row, every rate, every spike — plus a direct oracle check of the rates on the trajectory rows the
pipeline produced, and that no wait of the pipeline ever gave up (`diagnostics["pipeline_timeouts"] == 0`).
A stale read (a rate evaluated on a row that was not published yet) shows as a mismatch of whole 4-float groups."""
import os

import numpy as np
import pytest
import torch

from oracle import riab_oracle as orc

pytestmark = pytest.mark.gpu

MAZE = [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]


@pytest.fixture(scope="module")
def riab():
    assert torch.cuda.is_available(), "these tests need the GPU"
    import ratinabox_amd
    return ratinabox_amd


def _world(riab, B, make_pop, env_params=None, seed=7, agent_params=None):
    np.random.seed(seed)
    env = riab.Environment(env_params or {})
    ag = riab.Agent(env, dict({"n_agents": B, "dt": 0.01, "seed": 99}, **(agent_params or {})))
    np.random.seed(seed + 1)
    pop = make_pop(riab, ag)
    return env, ag, pop


def _run(riab, fused, B, make_pop, schedule, env_params=None, drift=None, agent_params=None):
    """schedule: list of ("sim", T) / ("step", k) entries.  Returns (trajectory, rates, spikes, agent)."""
    os.environ["RIAB_NO_FUSED"] = "0" if fused else "1"
    os.environ["RIAB_NO_NATIVE"] = "0" if fused else "1"   # (the reference is the Python-driven chunked pipeline)
    try:
        env, ag, pop = _world(riab, B, make_pop, env_params, agent_params=agent_params)
        for what, n in schedule:
            if what == "sim":
                ag.simulate(n, drift_velocity=drift)
            else:
                for _ in range(n):
                    ag.update(drift_velocity=drift)
                    pop.update()
        torch.cuda.synchronize()
        traj = ag.get_history_tensor().cpu().numpy()
        fr, sp = pop.get_history_tensors()
        return traj, fr.cpu().numpy(), sp.cpu().numpy(), ag
    finally:
        os.environ.pop("RIAB_NO_FUSED", None)
        os.environ.pop("RIAB_NO_NATIVE", None)


def _pc(n, **kw):
    def make(riab, ag):
        return riab.PlaceCells(ag, dict({"n": n, "wall_geometry": "euclidean"}, **kw))
    return make


def _gc(n, **kw):
    def make(riab, ag):
        return riab.GridCells(ag, dict({"n": n}, **kw))
    return make


def _hdc(n, **kw):
    def make(riab, ag):
        return riab.HeadDirectionCells(ag, dict({"n": n}, **kw))
    return make


CASES = [
    # name, B, population, schedule, env, drift
    ("pc_cfg2_20", 4096, _pc(1024, save_spikes=False), [("sim", 20)], None, None),
    ("pc_cfg2_133", 4096, _pc(1024, save_spikes=False), [("sim", 133)], None, None),
    ("pc_short_runs", 1024, _pc(256, save_spikes=False), [("sim", 1), ("sim", 3), ("sim", 4), ("sim", 9)], None, None),
    ("pc_ragged_cells_spikes", 512, _pc(203), [("sim", 37)], None, None),
    ("pc_tiny", 256, _pc(5), [("sim", 50)], None, None),
    ("pc_mixed_with_eager", 1024, _pc(64), [("step", 3), ("sim", 21), ("step", 2), ("sim", 8)], None, None),
    ("pc_threshold_los_maze", 1024, _pc(100, description="gaussian_threshold", wall_geometry="line_of_sight"),
     [("sim", 60)], {"walls": MAZE}, None),
    ("pc_dog_periodic", 512, _pc(77, description="diff_of_gaussians", widths=0.1), [("sim", 45)],
     {"boundary_conditions": "periodic"}, None),
    ("pc_top_hat_drift", 512, _pc(64, description="top_hat"), [("sim", 30)], None, np.array([0.3, -0.1])),
    ("gc_rectified_spikes", 1024, _gc(300), [("sim", 40)], None, None),
    ("gc_shifted", 512, _gc(64, description="shifted_cosines", save_spikes=False), [("sim", 25)], {"walls": MAZE}, None),
    ("hdc_spikes", 1024, _hdc(50), [("sim", 64)], None, None),
    ("hdc_wide", 2048, _hdc(200, save_spikes=False), [("sim", 17)], None, None),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fused_equals_chunked_pipeline(riab, case):
    _name, B, make_pop, schedule, env_params, drift = case
    t_a, fr_a, sp_a, ag_a = _run(riab, True, B, make_pop, schedule, env_params, drift)
    assert ag_a._streamer is not None, "the flag-coupled pipeline did not run"
    assert ag_a.diagnostics["pipeline_timeouts"] == 0
    t_b, fr_b, sp_b, ag_b = _run(riab, False, B, make_pop, schedule, env_params, drift)
    assert ag_b._streamer is None
    assert t_a.shape == t_b.shape and fr_a.shape == fr_b.shape and sp_a.shape == sp_b.shape
    np.testing.assert_array_equal(t_a, t_b)
    np.testing.assert_array_equal(fr_a, fr_b)
    np.testing.assert_array_equal(sp_a, sp_b)
    np.testing.assert_array_equal(ag_a.state_tensor.cpu().numpy(), ag_b.state_tensor.cpu().numpy())
    assert ag_a.t == pytest.approx(ag_b.t) and ag_a._step_index == ag_b._step_index


def test_fused_rates_match_oracle_on_its_own_rows(riab):
    """Independent of the chunked path: the rates the pipeline wrote are the oracle's rates at the positions the
    pipeline's own trajectory rows hold (1e-5 relative, north_star) — for every time row of a 20-step run at the
    bench shape, so a row evaluated before it was published cannot hide."""
    os.environ.pop("RIAB_NO_FUSED", None)
    env, ag, pcs = _world(riab, 4096, _pc(1024, save_spikes=False))
    ag.simulate(20)
    torch.cuda.synchronize()
    assert ag._streamer is not None and ag.diagnostics["pipeline_timeouts"] == 0
    traj = ag.get_history_tensor().cpu().numpy()
    fr = pcs.get_history_tensors()[0].cpu().numpy()
    oenv = orc.EnvSpec(walls=np.zeros((0, 2, 2)))
    for t in range(20):
        pos = np.stack((traj[t, 0], traj[t, 1]), -1).astype(np.float64)
        ref = orc.place_cells(oenv, pos, pcs.place_cell_centres, pcs.place_cell_widths)
        np.testing.assert_allclose(fr[t], ref, rtol=1e-5, atol=1e-37)


def test_cfg2_full_length_run_in_every_form(riab):
    """BASELINE configs[1] at its full size and length — 4096 agents x 1024 PlaceCells, 1000 steps after 100 — through
    the row-following kernel, the chunk form (RIAB_NO_FUSED=1) and the Python-driven pipeline (RIAB_NO_NATIVE=1): the
    trajectories are identical and so is every time row's checksum of rates (float64 sum and sum of squares per row,
    reduced on the device: 18 GB of rates per run are not downloaded); the last row against the oracle."""
    sums = {}
    for mode, envs in (("one-kernel", {}), ("chunks", {"RIAB_NO_FUSED": "1"}), ("python", {"RIAB_NO_NATIVE": "1"})):
        os.environ.update(envs)
        try:
            env, ag, pcs = _world(riab, 4096, _pc(1024, save_spikes=False))
            ag.simulate(100)
            ag.simulate(1000)
            torch.cuda.synchronize()
            if mode != "python":
                assert ag.last_rate_stage_form() == mode and ag.diagnostics["pipeline_timeouts"] == 0
            fr = pcs.get_history_tensors()[0]
            assert fr.shape == (1100, 1024, 4096)
            f64 = [fr[i:i + 50].to(torch.float64) for i in range(0, 1100, 50)]
            sums[mode] = (ag.get_history_tensor().cpu(), torch.cat([x.sum((1, 2)) for x in f64]).cpu(),
                          torch.cat([(x * x).sum((1, 2)) for x in f64]).cpu())
            if mode == "one-kernel":
                row = ag.get_history_tensor()[-1].cpu().numpy()
                sel = np.arange(0, 4096, 129)
                pos = np.stack((row[0, sel], row[1, sel]), -1).astype(np.float64)
                ref = orc.place_cells(orc.EnvSpec(walls=np.zeros((0, 2, 2))), pos, pcs.place_cell_centres, pcs.place_cell_widths)
                np.testing.assert_allclose(fr[-1][:, sel].cpu().numpy(), ref, rtol=1e-5, atol=1e-37)
            del fr, f64, ag, pcs
            torch.cuda.empty_cache()
        finally:
            for k in envs:
                os.environ.pop(k, None)
    for other in ("chunks", "python"):
        for x, y in zip(sums["one-kernel"], sums[other]):
            assert torch.equal(x, y), other


def test_fused_many_calls_and_long_run(riab):
    """Progress words are absolute step counts and are never reset: many back-to-back calls (no synchronisation
    in between), then one long call, all equal to the chunked path."""
    sched = [("sim", 5)] * 12 + [("sim", 700)]
    t_a, fr_a, _sp, ag_a = _run(riab, True, 1024, _pc(128, save_spikes=False), sched)
    assert ag_a.diagnostics["pipeline_timeouts"] == 0
    t_b, fr_b, _sp, _ag = _run(riab, False, 1024, _pc(128, save_spikes=False), sched)
    np.testing.assert_array_equal(t_a, t_b)
    np.testing.assert_array_equal(fr_a, fr_b)


def test_fused_without_history_keeps_last_rows(riab):
    """save_history=False: the rates stream through a ring; the newest row is what `firingrate` shows."""
    res = []
    for fused in (True, False):
        os.environ["RIAB_NO_FUSED"] = "0" if fused else "1"
        try:
            env, ag, pcs = _world(riab, 512, _pc(96, save_history=False), agent_params={"save_history": False})
            ag.simulate(300, chunk=32)
            torch.cuda.synchronize()
            res.append((np.asarray(ag.pos), pcs.firingrate))
            if fused:
                assert ag._streamer is not None and ag.diagnostics["pipeline_timeouts"] == 0
        finally:
            os.environ.pop("RIAB_NO_FUSED", None)
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_one_engine_for_simulate(riab, monkeypatch):
    """Agent.simulate() is the native call for every run it accepts (AgentVectorCells and recurrent layers advance
    through update() only; the Python-driven chunk pipeline is left as the RIAB_NO_NATIVE=1 comparator): explicit `noise=`
    normals, per-call motion kwargs, `resample_positions=`, an imported trajectory, a batch that is neither whole waves
    nor whole 256-agent groups, an agent without populations — each makes riab_simulate calls (counted) and gives,
    bit for bit, the rows of the Python-driven chunk pipeline (RIAB_NO_NATIVE=1)."""
    L = riab._lib
    calls = {"n": 0}
    real = L.lib

    class Counting:
        def __getattr__(self, k):
            return getattr(real, k)

        def riab_simulate(self, *a):
            calls["n"] += 1
            return real.riab_simulate(*a)

    def world(B, pops=True, env_params=None):
        np.random.seed(11)
        env = riab.Environment(env_params or {"walls": [[[0.5, 0.0], [0.5, 0.4]]]})
        ag = riab.Agent(env, {"n_agents": B, "dt": 0.02, "seed": 5})
        np.random.seed(12)
        ps = [riab.PlaceCells(ag, {"n": 24}), riab.BoundaryVectorCells(ag, {"n": 6})] if pops else []
        return env, ag, ps

    rs = np.random.RandomState(0)
    z100 = rs.normal(size=(40, 2, 100))
    traj_t = np.linspace(0, 3, 60)
    traj_p = np.stack((0.5 + 0.3 * np.cos(traj_t), 0.5 + 0.3 * np.sin(traj_t)), -1)
    hole_env = {"holes": [[[0.4, 0.4], [0.6, 0.4], [0.6, 0.6], [0.4, 0.6]]]}
    rs_pos = np.stack((np.full((40, 100), 0.2), np.full((40, 100), 0.8)), -1)

    def cases():
        yield "noise", 100, True, None, lambda ag: ag.simulate(40, noise=z100)
        yield "kwargs", 100, True, None, lambda ag: ag.simulate(40, speed_mean=0.3, thigmotaxis=0.9)
        yield "ragged", 300, True, None, lambda ag: ag.simulate(70)
        yield "no populations", 128, False, None, lambda ag: ag.simulate(40, noise=np.zeros((40, 2, 128)))
        yield "resample", 100, True, hole_env, lambda ag: ag.simulate(40, resample_positions=rs_pos)

        def imported(ag):
            ag.import_trajectory(times=traj_t, positions=traj_p)
            ag.simulate(50)
        yield "imported", 4, True, None, imported

    for name, B, pops, envp, run in cases():
        got = []
        for native in (True, False):
            monkeypatch.setenv("RIAB_NO_NATIVE", "0" if native else "1")
            monkeypatch.setattr(L, "lib", Counting())
            calls["n"] = 0
            env, ag, ps = world(B, pops, envp)
            run(ag)
            torch.cuda.synchronize()
            monkeypatch.setattr(L, "lib", real)
            assert (calls["n"] >= 1) == native, (name, native, calls["n"])
            assert ag.engine_runs == {"native": int(native), "plan": 0, "chunks": int(not native)}, (name, ag.engine_runs)
            got.append([ag.get_history_tensor().cpu(), ag.state_tensor.cpu()] + [t.cpu() for N in ps for t in N.get_history_tensors()])
            assert ag.diagnostics.get("pipeline_timeouts", 0) == 0
        for x, y in zip(*got):
            assert torch.equal(x, y), name
    monkeypatch.delenv("RIAB_NO_NATIVE", raising=False)
    # any number of populations (there is no table of fixed size behind the call)
    got = []
    for native in (True, False):
        monkeypatch.setenv("RIAB_NO_NATIVE", "0" if native else "1")
        env, ag, ps = world(64, pops=False)
        np.random.seed(13)
        many = [riab.PlaceCells(ag, {"n": 3 + k % 4, "save_spikes": bool(k % 2)}) if k % 3 else riab.HeadDirectionCells(ag, {"n": 4})
                for k in range(40)]
        ag.simulate(30)
        torch.cuda.synchronize()
        assert ag.engine_runs["native"] == int(native)
        got.append([t.cpu() for N in many for t in N.get_history_tensors() if t is not None])
    assert len(got[0]) == len(got[1]) and all(torch.equal(x, y) for x, y in zip(*got))
    monkeypatch.delenv("RIAB_NO_NATIVE", raising=False)
    # populations that read the float64 state -> a native step plan; populations that follow ANOTHER Agent object
    # have no open-loop run at all (both agents advance through update())
    env, ag, ps = world(8)
    riab.VelocityCells(ag)
    ag.simulate(12)
    assert ag.engine_runs == {"native": 0, "plan": 1, "chunks": 0}
    env, ag, ps = world(8)
    other = riab.Agent(env, {"n_agents": 8, "dt": 0.02, "seed": 6})
    riab.AgentVectorCells(ag, other, {"n": 4})
    with pytest.raises(NotImplementedError):
        ag.simulate(12)


def test_repeated_simulate_calls_see_every_edit(riab, monkeypatch):
    """simulate() after a plain native simulate() takes a short road that reuses what the first call prepared
    (Agent._simulate_repeat) — after re-examining all of it.  A script that edits motion parameters, a tuning array in
    place, the firing-rate range, the geometry, the state, a seed, the history switches, steps with update() and
    changes dt between calls gives the same rows with the short road and without it."""
    def script(ag, pcs, gcs, env):
        ag.simulate(20)
        ag.simulate(20)                                   # (the short road, when it exists)
        ag.speed_mean = 0.2                               # reference tests/test_advanced.py:47-48
        ag.simulate(12)
        pcs.place_cell_centres[-1] = [0.9, 0.9]           # in-place edit of a tuning array (:59)
        ag.simulate(12)
        gcs.max_fr = 3.0
        ag.simulate(9)
        env.add_wall([[0.2, 0.6], [0.8, 0.6]])
        ag.simulate(16)
        ag.pos = np.full((ag.n_agents, 2), 0.25)
        ag.simulate(8)
        ag.update(); pcs.update(); gcs.update()
        ag.simulate(8)
        ag.simulate(8, dt=0.02)
        ag.simulate(8)
        gcs.save_history = False
        ag.simulate(8)
        gcs.save_history = True
        ag.simulate(8, drift_velocity=[0.05, 0.0])
        ag.simulate(8)
        ag.simulate(300)
        ag.simulate(8)
        torch.cuda.synchronize()
        return (ag.get_history_tensor().cpu(), pcs.get_history_tensors()[0].cpu(), gcs.get_history_tensors()[0].cpu(),
                gcs.get_history_tensors()[1].cpu(), ag.state_tensor.cpu(), list(ag.history["t"]), ag.diagnostics)

    res, used = [], []
    for short in (True, False):
        np.random.seed(3)
        env = riab.Environment({})
        ag = riab.Agent(env, {"n_agents": 512, "dt": 0.01, "seed": 17})
        np.random.seed(4)
        pcs = riab.PlaceCells(ag, {"n": 32, "save_spikes": False})
        gcs = riab.GridCells(ag, {"n": 16})
        count = {"n": 0}
        real = type(ag)._simulate_repeat

        def repeat(self, n, _real=real, _short=short, _count=count):
            if not _short:
                return None
            out = _real(self, n)
            _count["n"] += out is not None
            return out
        monkeypatch.setattr(type(ag), "_simulate_repeat", repeat)
        res.append(script(ag, pcs, gcs, env))
        used.append(count["n"])
        monkeypatch.undo()
    assert used[0] >= 4 and used[1] == 0, used
    for x, y in zip(res[0][:5], res[1][:5]):
        assert torch.equal(x, y)
    assert res[0][5] == res[1][5] and res[0][6] == res[1][6] and res[0][6]["pipeline_timeouts"] == 0


@pytest.mark.parametrize("gate,launches", [("reserved", 2), ("always", 3), ("when_busy", 2)])
def test_fused_launch_modes(riab, gate, launches):
    """Residency of the two kernels (include/riab_hip.h "Residency"): the row-following rate kernel in its RESERVING shape
    (twelve-wave workgroups: two per compute unit, one wave slot per SIMD always free for a trajectory workgroup — the
    default: two launches per call), behind the one-wave started gate (three launches), and straight behind the
    trajectory kernel with neither (an option for callers that own the device) give the same, correct rows; a call
    whose stream is busy takes the gate in every mode."""
    os.environ["RIAB_GATE"] = gate
    try:
        t_a, fr_a, _sp, ag_a = _run(riab, True, 1024, _pc(256, save_spikes=False), [("sim", 48), ("sim", 16)])
        assert ag_a.diagnostics["pipeline_timeouts"] == 0 and ag_a.diagnostics["pipeline_serialised"] <= 1   # (a lone count: a device hiccup)
        L = riab._lib
        torch.cuda.synchronize()
        ag_a.simulate(16)                       # the stream is idle: the mode's own number of launches
        assert L.lib.riab_streamer_info(ag_a._streamer, 3) == launches, (gate, L.lib.riab_streamer_info(ag_a._streamer, 3))
        x = torch.zeros(1 << 26, device="cuda")
        for _ in range(8):
            x.add_(1.0)                         # the stream is busy: event + started gate, whatever the mode
        ag_a.simulate(16)
        assert L.lib.riab_streamer_info(ag_a._streamer, 3) == 3
        torch.cuda.synchronize()
        assert ag_a.diagnostics["pipeline_timeouts"] == 0
        tail = ag_a.get_history_tensor()[-32:].cpu().numpy()
    finally:
        os.environ.pop("RIAB_GATE", None)
    sched = [("sim", 48), ("sim", 16), ("sim", 16), ("sim", 16)]
    t_b, fr_b, _sp, _ag = _run(riab, False, 1024, _pc(256, save_spikes=False), sched)
    np.testing.assert_array_equal(t_a, t_b[:64])
    np.testing.assert_array_equal(fr_a, fr_b[:64])
    np.testing.assert_array_equal(tail, t_b[64:])


def test_reserving_shape_at_the_bench_size_and_with_ragged_cell_groups(riab):
    """The twelve-wave shape of the row-following kernel covers three cell groups per workgroup: cell counts that are
    not a multiple of 24 (place cells: 8 per group), a population with spikes, grid and head-direction cells, and the
    bench shape itself — all bit-identical to the gated four-wave shape and to the Python-driven pipeline."""
    for B, make, T in ((4096, _pc(1024, save_spikes=False), 20), (512, _pc(203), 37), (256, _pc(5), 50), (1024, _gc(300), 40),
                       (1024, _hdc(50), 33), (2048, _pc(1000, save_spikes=False), 300)):
        res = {}
        for gate in ("reserved", "always"):
            os.environ["RIAB_GATE"] = gate
            try:
                res[gate] = _run(riab, True, B, make, [("sim", T), ("sim", 9)])
                ag = res[gate][3]
                assert ag.diagnostics["pipeline_timeouts"] == 0
                torch.cuda.synchronize()
                ag.simulate(8)
                assert riab._lib.lib.riab_streamer_info(ag._streamer, 3) == (2 if gate == "reserved" else 3)
                torch.cuda.synchronize()
            finally:
                os.environ.pop("RIAB_GATE", None)
        ref = _run(riab, False, B, make, [("sim", T), ("sim", 9)])
        for k in range(3):
            np.testing.assert_array_equal(res["reserved"][k], res["always"][k])
            np.testing.assert_array_equal(res["reserved"][k], ref[k])


def test_reserving_shape_is_refused_by_kernels_that_hold_too_many_registers(riab):
    """Line-of-sight / geodesic place cells hold ~90 registers per lane: two twelve-wave workgroups of them would not
    leave a trajectory workgroup its 224 — the reserving shape is not offered for them, the call takes the gate."""
    os.environ["RIAB_GATE"] = "reserved"
    try:
        t_a, fr_a, _sp, ag = _run(riab, True, 1024, _pc(100, wall_geometry="line_of_sight", save_spikes=False),
                                  [("sim", 30)], {"walls": MAZE})
        torch.cuda.synchronize()
        ag.simulate(12)
        assert ag.last_rate_stage_form() == "one-kernel" and riab._lib.lib.riab_streamer_info(ag._streamer, 3) == 3
        torch.cuda.synchronize()
        assert ag.diagnostics["pipeline_timeouts"] == 0
    finally:
        os.environ.pop("RIAB_GATE", None)
    t_b, fr_b, _sp, _ag = _run(riab, False, 1024, _pc(100, wall_geometry="line_of_sight", save_spikes=False), [("sim", 30)],
                               {"walls": MAZE})
    np.testing.assert_array_equal(fr_a, fr_b)


def test_replayed_argument_block_never_sees_the_earlier_runs_progress(riab):
    """torch.ops.riab.simulate_ on the SAME argument block twice (what a compiled function that
    is called twice does) starts the second run at the step count of the first.  The progress words hold absolute
    step counts, so the second run's consumers must not take the first run's words for their own: every trajectory
    workgroup resets its word before it announces itself, and a call that does not continue beyond the last one's
    rows takes the started gate.  Checked: the second run's rates are the oracle's rates on the second run's own rows
    (a consumer that ran ahead of the producer reads unwritten / stale positions), twice, with 4096 agents."""
    from ratinabox_amd import ops  # noqa: F401
    np.random.seed(5)
    ag = riab.Agent(riab.Environment(), {"n_agents": 4096, "dt": 0.01, "seed": 3})
    pcs = riab.PlaceCells(ag, {"n": 256, "save_spikes": False, "wall_geometry": "euclidean"})
    K = 40
    a = ag.simulate_args(K)
    oenv = orc.EnvSpec()
    prev = None
    for rep in range(3):
        a.hist.fill_(float("nan"))
        a.rates[0].fill_(float("nan"))
        torch.ops.riab.simulate_(a.state, a.hist, a.rates, a.spikes, a.ctrl, a.diag, a.streamer, a.run, 0, a.rate_rows, a.spike_rows)
        torch.cuda.synchronize()
        # (a streamer's first call and every call that does not continue beyond the last one's rows take the started gate)
        assert riab._lib.lib.riab_streamer_info(ag._streamer, 3) == 3
        hist, fr = a.hist.cpu().numpy(), a.rates[0].cpu().numpy()
        assert np.isfinite(hist).all() and np.isfinite(fr).all()
        sel = np.arange(0, 4096, 131)
        for t in (0, 1, 2, 3, 7, K // 2, K - 1):
            pos = np.stack((hist[t, 0, sel], hist[t, 1, sel]), -1).astype(np.float64)
            ref = orc.place_cells(oenv, pos, pcs.place_cell_centres, pcs.place_cell_widths)
            np.testing.assert_allclose(fr[t][:, sel], ref, rtol=1e-5, atol=1e-37)
        if prev is not None:
            assert not np.array_equal(prev, hist), "the state did not advance"
        prev = hist
    assert ag.diagnostics["pipeline_timeouts"] == 0


def test_repeated_simulate_with_a_feedforward_layer(riab):
    """the second plain simulate() of an agent whose populations include a FeedForwardLayer (no
    noisy population in the list) must not take the short road with a descriptor that needs the plan index."""
    res = []
    for native in (True, False):
        os.environ["RIAB_NO_NATIVE"] = "0" if native else "1"
        try:
            np.random.seed(8)
            ag = riab.Agent(riab.Environment(), {"n_agents": 256, "dt": 0.01, "seed": 2})
            np.random.seed(9)
            pcs = riab.PlaceCells(ag, {"n": 32, "save_spikes": False})
            ff = riab.FeedForwardLayer(ag, {"n": 6, "input_layers": [pcs], "save_spikes": False,
                                            "activation_function": {"activation": "relu", "gain": 1.0, "threshold": 0.0}})
            for _ in range(3):
                ag.simulate(24)
            torch.cuda.synchronize()
            if native:
                assert ag.engine_runs["native"] == 3
            res.append((ag.get_history_tensor().cpu(), pcs.get_history_tensors()[0].cpu(), ff.get_history_tensors()[0].cpu()))
        finally:
            os.environ.pop("RIAB_NO_NATIVE", None)
    for x, y in zip(*res):
        assert torch.equal(x, y)
    assert res[0][2].shape == (72, 6, 256) and float(res[0][2].abs().sum()) > 0


def test_form_selection_follows_the_measured_step_and_store_rates(riab):
    """populations form or chunk form is decided by comparing the lead population's stores of a row with
    1.5 trajectory steps next to it — measured on this chip from the device-clock stamps an earlier call left (one
    read per streamer), the MI355X constants until then.  Faking the figures flips the form; the rows are the same."""
    L = riab._lib

    def world():
        np.random.seed(12)
        ag = riab.Agent(riab.Environment(), {"n_agents": 2048, "dt": 0.01, "seed": 6})
        np.random.seed(13)
        return ag, riab.PlaceCells(ag, {"n": 512, "save_spikes": False}), riab.BoundaryVectorCells(ag, {"n": 8, "save_spikes": False})

    rows = {}
    for label, step_ns in (("constants", None), ("slow trajectory", 50_000), ("fast trajectory", 100)):
        ag, pcs, bvs = world()
        ag.simulate(16)                                # (creates the streamer; nothing measured yet)
        torch.cuda.synchronize()
        if step_ns is not None:
            L.check(L.lib.riab_streamer_configure(ag._streamer, L.STREAMER_OPT_STEP_NS, step_ns), "configure")
            assert L.lib.riab_streamer_info(ag._streamer, 0) == step_ns and L.lib.riab_streamer_info(ag._streamer, 2) == 0
        ag.simulate(64)
        torch.cuda.synchronize()
        form = ag.last_rate_stage_form()
        if label == "slow trajectory":                 # 4.2 MB per row at ~6.5 TB/s = 0.65 us < 1.5 x 50 us
            assert form == "chunks"
        elif label == "fast trajectory":
            assert form == "populations"
        else:                                          # the open-room constants: 0.65 us < 1.5 x 0.9 us -> chunks ...
            assert form == "chunks"
            # ... and this second multi-population call found the stream idle: it has read the first call's stamps
            assert L.lib.riab_streamer_info(ag._streamer, 2) == 1
            step, mbps = L.lib.riab_streamer_info(ag._streamer, 0), L.lib.riab_streamer_info(ag._streamer, 1)
            assert 200 < step < 20_000, step           # a trajectory step of a 16-step call: a few microseconds at most
            assert mbps == 6_500_000                   # (no row-following kernel ran: the store rate stays the constant)
        rows[label] = (ag.get_history_tensor().cpu(), pcs.get_history_tensors()[0].cpu(), bvs.get_history_tensors()[0].cpu())
        assert ag.diagnostics["pipeline_timeouts"] == 0
    for label in ("slow trajectory", "fast trajectory"):
        for x, y in zip(rows["constants"], rows[label]):
            assert torch.equal(x, y), label



def test_aborted_pipeline_flags_are_settled_on_the_next_host_read(riab):
    """A wait that gave up (here: the abort / timeout words set by hand after complete runs) must not pass silently:
    the first host read settles it — the rates of the runs since the last check are recomputed from the complete
    trajectory (a warning, diagnostics["pipeline_recovered"]) — clears the control words, and the next run is clean.
    What cannot be recomputed still raises: populations with additive noise."""
    env, ag, pop = _world(riab, 1024, _pc(64, save_spikes=False), None)
    ag.simulate(8)
    _ = ag.pos                                   # a clean run: reading is fine
    ag.simulate(8)
    torch.cuda.synchronize()
    before = pop.get_history_tensors()[0].clone()
    L = riab._lib
    ag.simulate(8)
    ag._ctrl[L.CTRL_TIMEOUTS] = 3
    ag._ctrl[L.CTRL_ABORT] = 1
    with pytest.warns(RuntimeWarning, match="recomputed"):
        _ = ag.history["pos"]
    d = ag.diagnostics
    assert d["pipeline_timeouts"] == 0 and d["pipeline_recovered"] == 1 and d["pipeline_timeouts_recovered"] == 3
    after = pop.get_history_tensors()[0]
    assert torch.equal(after[:16], before)       # (the recomputation wrote the same rows: nothing else was touched)
    ag.simulate(8)
    assert np.isfinite(np.asarray(ag.pos)).all() and np.isfinite(pop.firingrate).all()
    assert len(ag.history["t"]) == 32
    # additive noise: the noise state has advanced with the aborted run — not recomputable
    env, ag, pop = _world(riab, 256, _pc(16, noise_std=0.1), None)
    ag.simulate(8)
    ag._ctrl[L.CTRL_TIMEOUTS] = 1
    ag._ctrl[L.CTRL_ABORT] = 1
    with pytest.raises(L.RiabError, match="aborted"):
        _ = ag.history["pos"]
    ag.simulate(8)
    assert ag.diagnostics["pipeline_timeouts"] == 0


@pytest.mark.parametrize("pops", ["one_kernel", "chunks"])
def test_a_rate_stage_that_gives_up_is_recovered_bit_for_bit(riab, pops):
    """a REAL abort — the waits of the rate stage are given one poll (RIAB_STREAMER_OPT_SPIN_LIMIT), so its
    first waves give up before the trajectory kernel has published anything, set the abort flag, and every later wait
    of that call and of the next call returns at once: rate rows unwritten or computed from stale positions.  The
    trajectory kernel waits for nobody outside its workgroup and finishes.  The next host read recomputes both runs'
    rates with the stream-ordered kernels: every row of every population equal, bit for bit, to the undisturbed
    pipeline's."""
    L = riab._lib

    def world():
        np.random.seed(17)
        env = riab.Environment({"walls": [[[0.5, 0.0], [0.5, 0.4]]]} if pops == "chunks" else {})
        ag = riab.Agent(env, {"n_agents": 1024, "dt": 0.01, "seed": 5})
        np.random.seed(18)
        if pops == "one_kernel":
            ps = [riab.PlaceCells(ag, {"n": 96, "wall_geometry": "euclidean", "max_fr": 30})]
        else:
            ps = [riab.PlaceCells(ag, {"n": 40, "wall_geometry": "line_of_sight"}), riab.BoundaryVectorCells(ag, {"n": 12}),
                  riab.GridCells(ag, {"n": 24, "save_spikes": True, "max_fr": 20})]
            ps.append(riab.FeedForwardLayer(ag, {"n": 6, "input_layers": [ps[0], ps[2]],
                                                 "activation_function": {"activation": "tanh", "gain": 1.0, "threshold": 0.0}}))
        return ag, ps

    def rows(ag, ps):
        torch.cuda.synchronize()
        out = [ag.get_history_tensor().cpu()]
        for p in ps:
            fr, sp = p.get_history_tensors()
            out += [fr.cpu(), sp.cpu()]
        return out + [ag.state_tensor.cpu()]

    ag, ps = world()
    for n in (16, 40, 24, 8):
        ag.simulate(n)
    ref = rows(ag, ps)
    assert ag.diagnostics["pipeline_timeouts"] == 0 and ag.diagnostics["pipeline_recovered"] == 0

    ag, ps = world()
    ag.simulate(16)
    _ = ag.pos
    L.check(L.lib.riab_streamer_configure(ag._streamer, L.STREAMER_OPT_SPIN_LIMIT, 1), "configure")
    ag.simulate(40)
    ag.simulate(24)
    L.check(L.lib.riab_streamer_configure(ag._streamer, L.STREAMER_OPT_SPIN_LIMIT, 0), "configure")
    torch.cuda.synchronize()
    assert int(ag._ctrl[L.CTRL_ABORT].item()) == 1 and int(ag._ctrl[L.CTRL_TIMEOUTS].item()) >= 1
    with pytest.warns(RuntimeWarning, match="recomputed"):
        _ = ps[0].history["firingrate"]
    d = ag.diagnostics
    assert d["pipeline_recovered"] == 2 and d["pipeline_timeouts"] == 0 and d["pipeline_timeouts_recovered"] >= 1
    ag.simulate(8)                                # the pipeline itself again, clean
    got = rows(ag, ps)
    assert ag.diagnostics["pipeline_timeouts"] == 0
    for x, y in zip(ref, got):
        assert torch.equal(x, y)


@pytest.mark.parametrize("pops", ["one_kernel", "populations"])
def test_strict_mode_equals_default_mode_and_is_capturable(riab, pops):
    """the b2-conforming mode of riab_simulate (include/riab_hip.h "Two modes").  Same rows as the
    default mode bit for bit; four launches for the one-kernel form (opening kernel, trajectory, started gate, rates);
    and a call on a stream that is being captured is strict by itself: Agent.simulate() inside torch.cuda.graph, the
    graph replayed a hundred times, leaves — every time — the rows of the uncaptured call."""
    def world():
        np.random.seed(23)
        ag = riab.Agent(riab.Environment(), {"n_agents": 1024, "dt": 0.01, "seed": 9})
        np.random.seed(24)
        ps = [riab.PlaceCells(ag, {"n": 128, "wall_geometry": "euclidean", "save_spikes": pops != "one_kernel", "max_fr": 20})]
        if pops != "one_kernel":
            ps += [riab.HeadDirectionCells(ag, {"n": 12}), riab.BoundaryVectorCells(ag, {"n": 8})]
        return ag, ps

    def rows(ag, ps):
        torch.cuda.synchronize()
        out = [ag.get_history_tensor().cpu(), ag.state_tensor.cpu()]
        for p in ps:
            fr, sp = p.get_history_tensors()
            out += [fr.cpu(), sp.cpu()]
        return out

    ag, ps = world()
    for n in (12, 30, 12):
        ag.simulate(n)
    ref = rows(ag, ps)
    assert not ag.pipeline_info()["strict_last_call"]

    ag, ps = world()
    ag.pipeline_mode(strict=True)
    ag.simulate(12)
    info = ag.pipeline_info()
    assert info["strict_last_call"] and (pops != "one_kernel" or info["launches_last_call"] == 4), info
    ag.pipeline_mode(strict=False)
    ag.simulate(30)                               # default mode after a strict call: re-bases once, then counts on
    ag.pipeline_mode(strict=True)
    ag.simulate(12)
    got = rows(ag, ps)
    assert ag.diagnostics["pipeline_timeouts"] == 0
    for x, y in zip(ref, got):
        assert torch.equal(x, y)

    # captured: the same 12-step call, replayed
    ag, ps = world()
    ag.simulate(12)
    ag.simulate(30)
    torch.cuda.synchronize()
    state0 = ag.state_tensor.clone()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ag.simulate(12)
    torch.cuda.synchronize()
    assert ag.pipeline_info()["strict_last_call"]
    g.replay()                                    # (a capture records, it does not run: this is the call's first run)
    first = rows(ag, ps)
    for x, y in zip(ref, first):                  # ... the third call of `ref`
        assert torch.equal(x, y)
    for k in range(100):
        ag.state_tensor.copy_(state0)             # (a replay runs the same steps again: from the same state)
        g.replay()
    again = rows(ag, ps)
    assert ag.diagnostics["pipeline_timeouts"] == 0
    for x, y in zip(ref, again):
        assert torch.equal(x, y)
    ag.simulate(6)                                # default mode after a captured call: re-bases every time from now on
    torch.cuda.synchronize()
    assert ag.diagnostics["pipeline_timeouts"] == 0 and len(ag.history["t"]) == 60


def test_fused_falls_back_for_uncovered_populations(riab):
    """one_hot PlaceCells (and anything else the stream kernel does not cover) take the chunked path and give
    the same answer as before; nothing is left half-reserved in the histories."""
    os.environ.pop("RIAB_NO_FUSED", None)
    env, ag, pcs = _world(riab, 512, _pc(32, description="one_hot"))
    ag.simulate(10)
    ag.simulate(6)
    torch.cuda.synchronize()
    assert len(ag.history["t"]) == 16 and pcs.history["firingrate"].shape == (16, 32, 512)
    assert np.all(pcs.history["firingrate"].sum(axis=1) == 1.0)


def test_history_view_after_plan_steps(riab):
    """read Ag.history, step a plan, read again — the second read must see the new rows."""
    env, ag, pcs = _world(riab, 256, _pc(16))
    plan = ag.make_step_plan(capacity=32)
    plan.step(5)
    assert ag.history["pos"].shape[0] == 5 and pcs.history["firingrate"].shape[0] == 5
    plan.step(5)
    assert ag.history["pos"].shape[0] == 10 and pcs.history["firingrate"].shape[0] == 10
    assert len(ag.get_history_arrays()["t"]) == 10
    # an eager Neurons.update() / reset_history() closes the plan instead of sharing its open rows
    ag.update()
    pcs.update()
    assert ag._plan is None and pcs.history["firingrate"].shape[0] == 11
    with pytest.raises(RuntimeError):
        plan.step(1)


def test_two_agent_objects_draw_different_noise(riab):
    np.random.seed(3)
    env = riab.Environment()
    a0 = riab.Agent(env, {"n_agents": 64, "dt": 0.01})
    a1 = riab.Agent(env, {"n_agents": 64, "dt": 0.01})
    a1.pos, a1.velocity = a0.pos, a0.velocity
    a1.head_direction, a1.measured_velocity = a0.head_direction, a0.measured_velocity
    for _ in range(5):
        a0.update()
        a1.update()
    assert not np.allclose(a0.pos, a1.pos), "two Agent objects replayed the same Philox streams"


def _bench(args, extra_env=None, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(extra_env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract_line_on_the_drivers_command(riab):
    """`bench.py --gpus 1 --steps 20 --warmup 5` must carry a non-null roofline whose numbers are
    consistent with the line's own value."""
    out = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5 and out["repeats"] >= 3
    rf = out["roofline"]
    # (the kernel's duration: HIP events in ten extra regions; the device-clock figure of the timed regions beside it)
    assert rf is not None and rf["bound"] == "hbm" and rf["launches"] == 10 and rf["rate_stage_form"] == "one-kernel"
    assert 0.0 < rf["avg_launch_ms_device_clock"] <= rf["avg_launch_ms"] * 1.5   # (the same kernel, minus its ramp and write-back; shared hosts)
    assert 0.0 < rf["frac"] <= 1.0 and rf["units_per_launch"] == 4096 * 20
    # the dominant kernel cannot take longer than the region it is timed in
    assert rf["avg_launch_ms"] <= out["timed_region_ms"]["max"]
    assert out["diagnostics"].get("pipeline_timeouts", 0) == 0
    assert abs(out["value"] - 4096 * 20 / (out["timed_region_ms"]["median"] * 1e-3)) / out["value"] < 1e-3


def bench_cpus(text):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    return bench._parse_cpulist(text)


def test_bench_launches_its_own_ranks(riab):
    """`python bench.py --gpus 2` (no launcher) must start two ranks and say so.  Both ranks share
    this box's one GPU (RIAB_BENCH_SHARE_GPU: gloo control plane), which exercises everything but RCCL."""
    out = _bench(["--gpus", "2", "--steps", "64", "--warmup", "8", "--no-cpu-baseline", "--secondary-timeout", "150"],
                 {"RIAB_BENCH_SHARE_GPU": "1"}, timeout=400)
    assert "secondary_error" not in out, out["secondary_error"]
    assert out["n_gpus"] == 2 and out["steps"] == 64
    assert out["config"]["parallelism"].startswith("agent-sharded x2")
    assert out["value"] > 1e6 and out["diagnostics"].get("pipeline_timeouts", 0) == 0
    assert out["scaling"] == "weak" and out["config"]["agents_per_gpu"] == 4096
    # what a multi-rank line carries — where each rank's host thread runs (cores of its GPU's NUMA node,
    # disjoint between the ranks), every rank's own timed regions, and BASELINE's 8-GPU configurations
    hb = out["config"]["host_binding"]
    assert hb["binding"] in ("numa", "none") and len(out["config"]["host_binding_per_rank"]) == 2
    if hb["binding"] == "numa":
        a, b = (set(bench_cpus(x["cpus"])) for x in out["config"]["host_binding_per_rank"])
        assert a and b and not (a & b), "the two ranks share cores"
    per = out["timed_region_ms_per_rank"]
    assert [x["rank"] for x in per] == [0, 1] and all(0 < x["min"] <= x["median"] <= x["max"] for x in per)
    assert out["timed_region_ms"]["median"] >= max(x["min"] for x in per)
    sec = out["secondary"]
    closed = {"cfg2_closed_loop_plan", "cfg2_closed_loop_per_step", "cfg2_closed_loop_task", "cfg2_closed_loop_task_world"}
    assert set(sec) == {"cfg2_T1024", "cfg4", "cfg5"} | closed, sec.keys()
    for name, blk in sec.items():
        assert "error" not in blk, blk
        assert blk["n_gpus"] == 2 and blk["value"] > 1e6 and len(blk["timed_region_ms_per_rank"]) == 2
        assert blk["diagnostics"].get("pipeline_timeouts", 0) == 0
        if name in closed and "task" not in name:   # the one-launch step served the loop
            assert blk["plan"]["fused_steps"] > 0, blk["plan"]
        assert 0 < blk["frac_whole_path"] < 1 and blk["roofline"]["frac"] > 0
    # the driver's region length with two ranks on the chip: neither rank's pipeline serialised or timed out, and every
    # rank says where its host spent the region
    short = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"],
                   {"RIAB_BENCH_SHARE_GPU": "1"}, timeout=400)
    for x in short["timed_region_ms_per_rank"]:
        # (two processes on ONE chip delay each other's dispatches: a rate stage that finds every row published then is
        # contention, not a shared hardware queue — seen on 4 to 11 of the rank's 31+ calls here, box by box; queue sharing
        # shows on EVERY call)
        assert x["pipeline_timeouts"] == 0 and x["pipeline_serialised"] <= 20, x
        assert set(x["host_us"]) == {"python_before_native_call", "in_native_call", "call_return_to_synchronised"}
        assert 0 < x["host_us"]["in_native_call"] < 1000
    strong = _bench(["--gpus", "2", "--strong", "--steps", "64", "--warmup", "8", "--no-cpu-baseline"],
                    {"RIAB_BENCH_SHARE_GPU": "1"}, timeout=400)
    assert strong["scaling"] == "strong" and strong["config"]["agents_per_gpu"] == 2048 and strong["n_gpus"] == 2


# ----------------------------------------------------------------------------- the unchanged per-step loop
def _loop(riab, auto, script, B=256, seed=11):
    """Run `script(riab, env, ag, pops)` — reference-style per-object update() calls — with the AutoStepper on/off."""
    os.environ["RIAB_NO_AUTO_PLAN"] = "0" if auto else "1"
    try:
        np.random.seed(seed)
        env = riab.Environment({"walls": [[[0.5, 0.0], [0.5, 0.35]]]})
        ag = riab.Agent(env, {"n_agents": B, "dt": 0.02, "seed": 4})
        np.random.seed(seed + 1)
        pops = [riab.PlaceCells(ag, {"n": 40, "wall_geometry": "line_of_sight"}), riab.BoundaryVectorCells(ag, {"n": 12}), riab.HeadDirectionCells(ag, {"n": 8}),
                riab.GridCells(ag, {"n": 16, "noise_std": 0.1})]
        ff = riab.FeedForwardLayer(ag, {"n": 6, "input_layers": [pops[0], pops[3]],
                                        "activation_function": {"activation": "tanh", "gain": 1.0, "threshold": 0.0}})
        pops.append(ff)
        used = script(riab, env, ag, pops)
        torch.cuda.synchronize()
        out = {"traj": ag.get_history_tensor().cpu().numpy(), "state": ag.state_tensor.cpu().numpy(), "t": list(ag.history["t"])}
        for i, p in enumerate(pops):
            fr, sp = p.get_history_tensors()
            out[f"fr{i}"], out[f"sp{i}"], out[f"t{i}"] = fr.cpu().numpy(), sp.cpu().numpy(), list(p.history["t"])
            out[f"last{i}"] = p.firingrate
        return out, used
    finally:
        os.environ.pop("RIAB_NO_AUTO_PLAN", None)


def _plain(riab, env, ag, pops):
    for _ in range(40):
        ag.update()
        for p in pops:
            p.update()
    return ag._plan is not None and type(ag._plan).__name__ == "AutoStepper"


def _with_edits(riab, env, ag, pops):
    engaged = []
    for t in range(60):
        if t == 15:
            ag.speed_mean = 0.2                      # reference tests/test_advanced.py:47-48
        if t == 22:
            pops[0].place_cell_centres[-1] = [0.9, 0.9]   # in-place edit of a tuning array (:59)
        if t == 30:
            ag.pos = np.full((ag.n_agents, 2), 0.25)  # the populations must read the edited state
        if t == 38:
            env.add_wall([[0.2, 0.6], [0.8, 0.6]])
        ag.update()
        for i, p in enumerate(pops):
            if not (i == 2 and t % 3 == 0):          # a population that is not updated every step
                p.update()
        if t == 45:
            ag.update(drift_velocity=np.array([0.1, 0.0]))   # a call that is not plain
            pops[0].update()
        engaged.append(ag._plan is not None and type(ag._plan).__name__ == "AutoStepper")
    return any(engaged[5:14]) and any(engaged[50:])


def _closed_loop(riab, env, ag, pops):
    """The closed loop of the reference (a policy producing drift_velocity every step, TaskEnvironment.py:399-408):
    a device tensor per agent, a plain array for everyone, another strength ratio, plain steps in between."""
    B = ag.n_agents
    target = torch.tensor([0.8, 0.8], dtype=torch.float64, device="cuda")
    engaged = []
    for t in range(70):
        pos = ag.state_tensor[:2, :B].t()                       # (B, 2) on the device
        v = 0.3 * (target - pos) / (target - pos).norm(dim=1, keepdim=True).clamp_min(1e-9)
        if t < 25:
            ag.update(drift_velocity=v)
        elif t < 35:
            ag.update(drift_velocity=v.t().contiguous(), drift_to_random_strength_ratio=3.0)   # (2, B), another ratio
        elif t < 45:
            ag.update()
        elif t < 55:
            ag.update(drift_velocity=np.array([0.05, -0.1]))
        else:
            ag.update(dt=ag.dt, drift_velocity=v.float())        # the way TaskEnvironment.step calls it; float32 policy
        for p in pops:
            p.update()
        engaged.append(ag._plan is not None and type(ag._plan).__name__ == "AutoStepper")
    return all(engaged[8:])


@pytest.mark.parametrize("script", [_plain, _with_edits, _closed_loop], ids=["plain", "with_edits", "closed_loop"])
def test_unchanged_reference_loop_is_served_natively_and_bit_identical(riab, script):
    """`Ag.update(); N.update() ...` from Python, no plan made by the caller: after a few rounds the
    calls are served by plan.AutoStepper; histories, states, spikes, times: identical to the eager calls, also
    across attribute edits, skipped populations, added walls and calls with arguments."""
    a, used = _loop(riab, True, script)
    b, used_b = _loop(riab, False, script)
    assert used and not used_b
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], list):
            assert a[k] == b[k], k
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


# ----------------------------------------------------------------------------- several populations, one native call
L_ROOM = [[0, 0], [1, 0], [1, 0.5], [0.5, 0.5], [0.5, 1], [0, 1]]


def _multi_world(riab, B, seed=5, polygon=False):
    np.random.seed(seed)
    if polygon:   # an L-shaped room with a hole: the general wall arithmetic, re-sampling, the slow chunk ramp
        env = riab.Environment({"boundary": L_ROOM, "holes": [[[0.15, 0.15], [0.3, 0.15], [0.3, 0.3], [0.15, 0.3]]],
                                "walls": [[[0.7, 0.0], [0.7, 0.3]]]})
    else:
        env = riab.Environment({"walls": [[[0.5, 0.0], [0.5, 0.35]], [[0.2, 0.7], [0.6, 0.7]]]})
    env.add_object([0.3, 0.3], type=0)
    env.add_object([0.8, 0.6], type=1)
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.02, "seed": 4})
    np.random.seed(seed + 1)
    pops = [riab.PlaceCells(ag, {"n": 40, "wall_geometry": "line_of_sight", "save_spikes": True}),
            riab.BoundaryVectorCells(ag, {"n": 12}),
            riab.HeadDirectionCells(ag, {"n": 8}),
            riab.GridCells(ag, {"n": 16, "noise_std": 0.1, "save_spikes": True}),
            riab.ObjectVectorCells(ag, {"n": 6}),
            riab.BoundaryVectorCells(ag, {"n": 5, "reference_frame": "egocentric"})]
    pops.append(riab.FeedForwardLayer(ag, {"n": 6, "input_layers": [pops[0], pops[3]], "noise_std": 0.05,
                                           "activation_function": {"activation": "tanh", "gain": 1.0, "threshold": 0.0}}))
    return env, ag, pops


@pytest.mark.parametrize("B, schedule, drift, polygon", [(64, [20], None, False), (256, [150, 7], [0.05, -0.02], False),
                                                        (128, [300], None, False), (128, [300, 40], None, True)])
def test_native_multi_population_simulate_equals_chunked_pipeline(riab, B, schedule, drift, polygon):
    """Agent.simulate() with several populations is ONE native call (riab_simulate: every chunk of rows behind a
    gate, each population's ordinary kernel, noise pass and spikes after it) and gives, bit for bit, what the
    Python-driven chunked pipeline gives: place / grid (+ OU noise) / head direction / boundary (allo- and egocentric) /
    object vector cells and a FeedForwardLayer reading two of them, spikes included."""
    got = {}
    for native in (True, False):
        os.environ["RIAB_NO_NATIVE"] = "0" if native else "1"
        try:
            env, ag, pops = _multi_world(riab, B, polygon=polygon)
            for n in schedule:
                ag.simulate(n, drift_velocity=drift)
            torch.cuda.synchronize()
            got[native] = dict(traj=ag.get_history_tensor().cpu().numpy(), state=ag.state_tensor.cpu().numpy(),
                               t=list(ag.history["t"]), diag=ag.diagnostics,
                               pops=[tuple(x.cpu().numpy() for x in N.get_history_tensors()) for N in pops],
                               last=[np.array(N.firingrate) for N in pops])
        finally:
            os.environ.pop("RIAB_NO_NATIVE", None)
    a, b = got[True], got[False]
    assert a["diag"]["pipeline_timeouts"] == 0 and "pipeline_timeouts" not in b["diag"]   # (only the native path has them)
    np.testing.assert_array_equal(a["traj"], b["traj"])
    np.testing.assert_array_equal(a["state"], b["state"])
    assert a["t"] == b["t"] and len(a["t"]) == sum(schedule)
    for (fa, sa), (fb, sb), la, lb in zip(a["pops"], b["pops"], a["last"], b["last"]):
        np.testing.assert_array_equal(fa, fb)
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(la, lb)
        assert fa.shape[0] == sum(schedule) and np.isfinite(fa).all()
    # and the eager per-step loop continues from there on both
    env, ag, pops = _multi_world(riab, B, polygon=polygon)
    ag.simulate(schedule[0], drift_velocity=drift)
    ag.update()
    for N in pops:
        N.update()
    assert len(ag.history["t"]) == schedule[0] + 1 and np.isfinite(pops[-1].firingrate).all()


def test_population_major_form_equals_chunked_pipeline(riab):
    """Several populations whose store-bound members write enough per row to keep pace with the trajectory kernel:
    those run first, one kernel each over the whole run (following the trajectory through its progress words), the
    others afterwards over all rows (riab_simulate, RIAB_FORM_POPULATIONS).  Same rows, bit for bit, as the chunk form
    (RIAB_NO_FUSED=1) and as the Python-driven pipeline; a room with many walls (a slower trajectory kernel) keeps the
    chunks."""
    def world(walls=()):
        np.random.seed(21)
        env = riab.Environment({"walls": list(walls)})
        ag = riab.Agent(env, {"n_agents": 1024, "dt": 0.02, "seed": 8})
        np.random.seed(22)
        pcs = riab.PlaceCells(ag, {"n": 2400, "save_spikes": False})                       # lead: 9.8 MB per row
        bvc = riab.BoundaryVectorCells(ag, {"n": 8, "save_spikes": True})                  # rest
        hdc = riab.HeadDirectionCells(ag, {"n": 16, "save_spikes": True})                  # lead, listed after a rest one
        gcn = riab.GridCells(ag, {"n": 12, "noise_std": 0.1, "save_spikes": True})         # rest (OU noise)
        ffl = riab.FeedForwardLayer(ag, {"n": 5, "input_layers": [hdc, bvc, gcn], "save_spikes": False,
                                         "activation_function": {"activation": "tanh", "gain": 1.0, "threshold": 0.0}})
        return ag, [pcs, bvc, hdc, gcn, ffl]

    got = {}
    for mode, envs in (("populations", {}), ("chunks", {"RIAB_NO_FUSED": "1"}), ("python", {"RIAB_NO_NATIVE": "1"})):
        os.environ.update(envs)
        try:
            ag, pops = world()
            for n in (70, 1100):     # (the second run is longer than a rest population's 1024-row launches)
                ag.simulate(n)
            torch.cuda.synchronize()
            if mode != "python":
                assert ag.last_rate_stage_form() == mode and ag.diagnostics["pipeline_timeouts"] == 0
            got[mode] = [ag.get_history_tensor().cpu(), ag.state_tensor.cpu()] + \
                        [t.cpu() for N in pops for t in N.get_history_tensors() if t is not None]
        finally:
            for k in envs:
                os.environ.pop(k, None)
    for other in ("chunks", "python"):
        assert len(got[other]) == len(got["populations"])
        for x, y in zip(got["populations"], got[other]):
            assert torch.equal(x, y), other
    # nine walls: the trajectory kernel's step outlasts the lead populations' stores of a row
    maze = [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]
    ag, pops = world(maze)
    ag.simulate(40)
    torch.cuda.synchronize()
    assert ag.last_rate_stage_form() == "chunks" and ag.diagnostics["pipeline_timeouts"] == 0


def test_very_long_runs_head_and_pieces(riab):
    """More than 2048 rows of one store-bound population: the row-following kernel serves the first 256 rows, the
    population's ordinary kernel the rest (512 rows per launch behind progress gates).  Bit-equal to the row-following
    kernel over all rows (RIAB_HEAD_ROWS=65535) and to the Python-driven pipeline; so is the same run with a second
    population (at this batch size: the chunk form)."""
    def world(two):
        np.random.seed(31)
        ag = riab.Agent(riab.Environment({}), {"n_agents": 256, "dt": 0.02, "seed": 9})
        np.random.seed(32)
        pops = [riab.PlaceCells(ag, {"n": 96, "save_spikes": True})]
        if two:
            pops.append(riab.BoundaryVectorCells(ag, {"n": 6, "save_spikes": False}))
        return ag, pops

    for two, form in ((False, "head+pieces"), (True, "chunks")):
        got = {}
        for mode, envs in (("default", {}), ("all rows", {"RIAB_HEAD_ROWS": "65535"}), ("python", {"RIAB_NO_NATIVE": "1"})):
            os.environ.update(envs)
            try:
                ag, pops = world(two)
                ag.simulate(2300)
                torch.cuda.synchronize()
                if mode == "default":
                    assert ag.last_rate_stage_form() == form
                if mode == "all rows" and not two:
                    assert ag.last_rate_stage_form() == "one-kernel"
                assert ag.diagnostics.get("pipeline_timeouts", 0) == 0
                got[mode] = [ag.get_history_tensor().cpu()] + [t.cpu() for N in pops for t in N.get_history_tensors() if t is not None]
            finally:
                for k in envs:
                    os.environ.pop(k, None)
        for other in ("all rows", "python"):
            for x, y in zip(got["default"], got[other]):
                assert torch.equal(x, y), (two, other)


# ----------------------------------------------------------------------------- imported trajectories through plans
def _replay_world(riab, B=8):
    np.random.seed(2)
    env = riab.Environment({"walls": [[[0.5, 0.0], [0.5, 0.4]]]})
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.02, "seed": 3})
    np.random.seed(4)
    pcs = riab.PlaceCells(ag, {"n": 24, "wall_geometry": "line_of_sight"})
    hdc = riab.HeadDirectionCells(ag, {"n": 6})
    tt = np.linspace(0, 3.0, 61)
    base = np.stack((0.5 + 0.35 * np.cos(2.1 * tt), 0.5 + 0.3 * np.sin(1.3 * tt + 0.4)), axis=-1)       # (61, 2)
    per_agent = base[:, None, :] + 0.05 * np.random.RandomState(9).uniform(-1, 1, (1, B, 2))              # (61, B, 2)
    ag.import_trajectory(times=tt, positions=per_agent)
    return env, ag, [pcs, hdc]


def test_imported_trajectory_through_plans_equals_eager_loop(riab):
    """An agent replaying an imported trajectory (reference Agent.py:255-266): the unchanged per-object loop served by
    the automatic step plan, and an explicit StepPlan, against the eager loop (one interpolation + one upload per
    update()) — bit for bit, across the wrap-around of the trajectory and a re-import in the middle."""
    T = 230   # (3.0 s of trajectory at dt = 0.02: wraps after 150 steps)

    def loop(auto):
        os.environ["RIAB_NO_AUTO_PLAN"] = "0" if auto else "1"
        try:
            env, ag, pops = _replay_world(riab)
            engaged = 0
            for i in range(T):
                ag.update()
                for N in pops:
                    N.update()
                engaged += int(ag._plan is not None)
                if i == 120:   # a new trajectory: the recorded positions are dropped, the loop goes on
                    tt = np.linspace(0, 2.0, 41)
                    ag.import_trajectory(times=tt, positions=np.stack((0.2 + 0.3 * tt, 0.8 - 0.25 * tt), axis=-1))
            torch.cuda.synchronize()
            return (np.array(ag.history["pos"]), np.array(ag.history["vel"]), np.array(ag.history["t"]),
                    [np.array(N.history["firingrate"]) for N in pops], engaged)
        finally:
            os.environ.pop("RIAB_NO_AUTO_PLAN", None)

    pa, va, ta, ra, engaged = loop(True)
    pb, vb, tb, rb, none = loop(False)
    assert engaged > T - 20 and none == 0
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(va, vb)
    np.testing.assert_array_equal(ta, tb)
    for x, y in zip(ra, rb):
        np.testing.assert_array_equal(x, y)
    # explicit plan, several steps per call
    env, ag, pops = _replay_world(riab)
    plan = ag.make_step_plan(capacity=64)
    for n in (1, 7, 64, 40, 8):
        plan.step(n)
    env2, ag2, pops2 = _replay_world(riab)
    os.environ["RIAB_NO_AUTO_PLAN"] = "1"
    try:
        for _ in range(120):
            ag2.update()
            for N in pops2:
                N.update()
    finally:
        os.environ.pop("RIAB_NO_AUTO_PLAN", None)
    np.testing.assert_array_equal(np.array(ag.history["pos"]), np.array(ag2.history["pos"]))
    np.testing.assert_array_equal(np.array(pops[0].history["firingrate"]), np.array(pops2[0].history["firingrate"]))
    with pytest.raises(NotImplementedError):
        plan.step(1, drift_velocity=[0.1, 0.0])


def test_simulate_pops_argument_errors_launch_nothing(riab):
    """riab_simulate validates before it launches: bad argument sets come back as negative codes and leave the
    agent state and the control words untouched."""
    L = riab._lib
    env, ag, pops = _multi_world(riab, 64)
    ag.simulate(4)                                    # (creates the streamer and the control block)
    torch.cuda.synchronize()
    before = ag.state_tensor.clone()
    ctrl_before = ag._ctrl.clone()
    envs, _w = env.device_tables(ag._device)
    m = ag._motion(ag.dt, False, 1, {})
    index, structs = {}, []
    for N in pops:
        structs.append(N._population(index))
        index[N] = len(index)
    T = 8
    outs = [torch.empty((T, int(N.n), ag._Bp), dtype=torch.float32, device="cuda") for N in pops]
    hist = torch.empty((T, L.HIST_ROWS, ag._Bp), dtype=torch.float32, device="cuda")

    def call(n_pops=None, B=None, cap=T, ff_input=None, kind=None):
        arr = (L.RiabPopulation * len(pops))()
        for i, (p, o) in enumerate(zip(structs, outs)):
            L.C.memmove(L.C.byref(arr, i * L.C.sizeof(L.RiabPopulation)), L.C.byref(p), L.C.sizeof(L.RiabPopulation))
            arr[i].rates_base, arr[i].spikes_base, arr[i].capacity_rows = o.data_ptr(), None, cap
        if ff_input is not None:
            arr[len(pops) - 1].input_index[0] = ff_input
        if kind is not None:
            arr[1].kind = kind
        run = L.RiabSimulate()
        run.env, run.motion = L.C.pointer(envs), L.C.pointer(m)
        run.state, run.B, run.agent_id0 = ag._state.data_ptr(), ag._Bp if B is None else B, 0
        run.seed, run.step0, run.T = int(ag.rng_seed), int(ag._step_index), T
        run.hist, run.diag, run.ctrl = hist.data_ptr(), ag._diag.data_ptr(), ag._ctrl.data_ptr()
        run.pops, run.n_pops = L.C.cast(arr, L.C.POINTER(L.RiabPopulation)), len(pops) if n_pops is None else n_pops
        run.timed_pop = -1
        return L.lib.riab_simulate(ag._streamer, L.C.byref(run), L.current_stream())

    assert call(n_pops=-1) == L.EINVAL                      # (0 populations: the trajectory kernel alone)
    assert call(B=ag._Bp + 2) == L.EALIGN                  # not whole quads of agents
    assert call(cap=T - 1) == L.EINVAL                      # rows for the whole run are required
    assert call(ff_input=len(pops) - 1) == L.EINVAL         # a layer reading itself / a later population
    assert call(kind=L.POP_KINDS["velocity"]) == L.EUNSUPPORTED
    torch.cuda.synchronize()
    assert torch.equal(before, ag.state_tensor) and torch.equal(ctrl_before, ag._ctrl)
    assert call() == 0                                      # and the well-formed call runs
    torch.cuda.synchronize()
    assert not torch.equal(before, ag.state_tensor) and all(torch.isfinite(o).all() for o in outs)


def test_native_multi_population_rings_equal_chunked_pipeline(riab):
    """save_history=False on some / all populations (and on the agent): their rates stream through a ring, one
    native call per ring length; the newest rows, the saved histories of the others and the state are those of the
    Python-driven pipeline."""
    res = []
    for native in (True, False):
        os.environ["RIAB_NO_NATIVE"] = "0" if native else "1"
        try:
            env, ag, pops = _multi_world(riab, 128)
            ag.save_history = False
            for i, N in enumerate(pops):
                N.save_history = i in (1, 3)          # BVCs and the noisy GridCells keep theirs; the FF layer reads a ring
            ag.simulate(700)
            ag.simulate(33)
            torch.cuda.synchronize()
            res.append(dict(state=ag.state_tensor.cpu().numpy(), last=[np.array(N.firingrate) for N in pops],
                            hist=[np.array(N.history["firingrate"]) for i, N in enumerate(pops) if i in (1, 3)],
                            nrows=[len(N.history["t"]) for N in pops], t=ag.t))
            if native:
                assert ag._streamer is not None and ag.diagnostics["pipeline_timeouts"] == 0
        finally:
            os.environ.pop("RIAB_NO_NATIVE", None)
    a, b = res
    np.testing.assert_array_equal(a["state"], b["state"])
    assert a["t"] == b["t"] and a["nrows"] == b["nrows"] and a["nrows"][1] == 733 and a["nrows"][0] == 0
    for x, y in zip(a["last"] + a["hist"], b["last"] + b["hist"]):
        np.testing.assert_array_equal(x, y)


def test_empty_and_tiny_inputs(riab):
    """Empty and ragged inputs: no positions -> (n, 0) rates; zero steps -> an empty trajectory and no history row;
    one agent / an agent count that is not a multiple of four / 64 / one more than a wave, one cell, run lengths 1, 2, 7
    through every path (fused, native multi-population, chunked, per-step)."""
    env = riab.Environment({"walls": [[[0.5, 0.0], [0.5, 0.4]]]})
    for B in (1, 3, 64, 65):
        np.random.seed(1)
        ag = riab.Agent(env, {"n_agents": B, "dt": 0.02})
        pops = [riab.PlaceCells(ag, {"n": 1}), riab.GridCells(ag, {"n": 5}), riab.BoundaryVectorCells(ag, {"n": 3}),
                riab.HeadDirectionCells(ag, {"n": 7})]
        rows = 0
        for n in (1, 0, 2, 7):
            traj = ag.simulate(n)
            assert traj.shape[0] == n
            rows += n
        for _ in range(6):
            ag.update()
            for p in pops:
                p.update()
        rows += 6
        assert len(ag.history["t"]) == rows and np.asarray(ag.history["pos"]).shape[0] == rows
        for p in pops:
            fr = np.asarray(p.history["firingrate"])
            assert fr.shape[:2] == (rows, int(p.n)) and np.isfinite(fr).all()
            for P in (0, 1, 5):
                out = p.get_state(evaluate_at=None, pos=np.full((P, 2), 0.3), head_direction=np.tile([0.0, 1.0], (P, 1)))
                assert out.shape == (int(p.n), P) and np.isfinite(out).all()
    solo = riab.Agent(env, {"n_agents": 256, "dt": 0.02})
    pcs = riab.PlaceCells(solo, {"n": 4})
    assert solo.simulate(0).shape[0] == 0 and len(solo.history["t"]) == 0 and solo._streamer is None


_SERIAL_WORKER = r"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import ratinabox_amd as riab
np.random.seed(0)
ag = riab.Agent(riab.Environment(), {"n_agents": 4096, "dt": 0.01, "seed": 1})
pcs = riab.PlaceCells(ag, {"n": 256, "save_spikes": False, "wall_geometry": "euclidean"})
ag.simulate(64); torch.cuda.synchronize()
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ag.simulate(64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
import warnings
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    d = ag.diagnostics
print(json.dumps({"serialised": d["pipeline_serialised"], "timeouts": d["pipeline_timeouts"], "ms": 1e3 * float(np.median(ts)),
                  "checksum": float(pcs.get_history_tensors()[0].double().sum())}))
"""


def test_serialised_pipeline_is_detected(riab, tmp_path):
    """the trajectory kernel and the rate stage are meant to run side by side on two hardware queues.  When
    they run one after the other (the process's streams share a queue) nothing is wrong with the results, only with the
    time — so the rate stage's first wave counts the calls (>= 8 rows) that find every row published already, and
    `Agent.diagnostics` reports them (and warns once).  (i) deterministic: the trajectory kernel on the caller's own
    stream (RIAB_SIDE_STREAM=2) — every call is counted, in both forms of the rate stage, and the rows are unchanged;
    (ii) ONE hardware queue for the process (GPU_MAX_HW_QUEUES=1, second stream at the default priority): if the
    runtime really serialised the two kernels — visible in the time — the counter has seen it."""
    import json
    import subprocess
    import sys
    import warnings
    sched = [("sim", 40), ("sim", 7), ("sim", 24), ("sim", 16), ("sim", 16)]
    ref = _run(riab, True, 1024, _pc(128, save_spikes=False), sched)
    assert ref[3].diagnostics["pipeline_serialised"] <= 1
    os.environ["RIAB_SIDE_STREAM"] = "2"
    try:
        got = _run(riab, True, 1024, _pc(128, save_spikes=False), sched)
        with pytest.warns(RuntimeWarning, match="one after the other"):   # (systematic: at least 3 calls and 5 % of them)
            d = got[3].diagnostics
        # (the 7-step call is too short to tell; a call whose second launch the host issued late is not counted)
        assert 3 <= d["pipeline_serialised"] <= 4 and d["pipeline_timeouts"] == 0
        for k in range(3):
            np.testing.assert_array_equal(got[k], ref[k])
        # the chunk form (two populations): its first gate does the counting
        np.random.seed(1)
        ag = riab.Agent(riab.Environment({"walls": MAZE}), {"n_agents": 512, "dt": 0.01, "seed": 5})
        riab.GridCells(ag, {"n": 32, "save_spikes": False})
        riab.BoundaryVectorCells(ag, {"n": 8, "save_spikes": False})
        ag.simulate(48)
        torch.cuda.synchronize()
        assert ag.last_rate_stage_form() == "chunks"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert ag.diagnostics["pipeline_serialised"] == 1
    finally:
        os.environ.pop("RIAB_SIDE_STREAM", None)
    script = tmp_path / "serial_worker.py"
    script.write_text(_SERIAL_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, extra in (("default", {}), ("one_queue", {"GPU_MAX_HW_QUEUES": "1", "RIAB_SIDE_STREAM": "1"})):
        env = dict(os.environ, **extra)
        p = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out[name] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    print("serialised-pipeline probe:", out)
    assert out["default"]["serialised"] <= 1 and out["default"]["timeouts"] == 0 and out["one_queue"]["timeouts"] == 0
    assert out["default"]["checksum"] == out["one_queue"]["checksum"]
    assert out["one_queue"]["serialised"] > 0 or out["one_queue"]["ms"] < 1.25 * out["default"]["ms"], out


def test_the_second_stream_pool_is_bounded(riab):
    """simulate() from twenty different caller streams: the same rows as from one stream, and the process-wide pool of
    second streams (csrc/riab_simulate.hip side_stream_for) holds a bounded number of HIP streams — eight callers'
    entries, a ninth takes over the least recently used one; rejected candidates are tried again beyond two dozen."""
    def world():
        np.random.seed(1)
        env = riab.Environment()
        ag = riab.Agent(env, {"n_agents": 256, "dt": 0.01, "seed": 5})
        pcs = riab.PlaceCells(ag, {"n": 32, "wall_geometry": "euclidean", "save_spikes": False})
        return ag, pcs

    ag, pcs = world()
    streams = [torch.cuda.Stream() for _ in range(20)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ag.simulate(8)
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    held = ag.pipeline_info()["second_stream"]["streams_held_by_the_pool"]
    assert 1 <= held <= 8 + 24 + 6, held
    ref_ag, ref_pcs = world()
    ref_ag.simulate(8 * len(streams))
    torch.cuda.synchronize()
    assert np.array_equal(np.asarray(ag.history["pos"]), np.asarray(ref_ag.history["pos"]))
    assert np.array_equal(np.asarray(pcs.history["firingrate"]), np.asarray(ref_pcs.history["firingrate"]))

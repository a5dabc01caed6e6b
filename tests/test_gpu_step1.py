"""GPU tests of the closed-loop step in ONE launch (csrc/riab_step1.hip, riab_plan_set_fused).

The one-launch step evaluates the motion step with the functions `riab_agent_step(T = 1)` calls and the rates with the
functors the rate kernels call, so the requirement is BIT-IDENTITY with the two-launch step (`riab_set_option
(RIAB_OPT_FUSED_STEP, 0)`), which `tests/test_gpu_parity.py` pins against the reference goldens and the oracle — every
state value, history row, rate and spike — plus a direct oracle check of the rates on the rows it produced, the
launch accounting (one kernel per step), and the arrival protocol's counter (no write-back ever gave up)."""
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


def _collect(ag, pops):
    torch.cuda.synchronize()
    out = {"traj": ag.get_history_tensor().cpu().numpy(), "state": ag.state_tensor.cpu().numpy(), "t": list(ag.history["t"]),
           "diag": dict(ag.diagnostics)}
    for i, p in enumerate(pops):
        fr, sp = p.get_history_tensors()
        out[f"fr{i}"] = fr.cpu().numpy()
        out[f"sp{i}"] = None if sp is None else sp.cpu().numpy()
        out[f"last{i}"] = np.array(p.firingrate)
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], (list, dict)) or a[k] is None:
            assert a[k] == b[k], k
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def _plan_run(riab, fused, make_world, steps, drift_from=None, capacity=64, batch=1):
    old = riab._lib.set_option("fused_step", 1 if fused else 0)
    try:
        env, ag, pops = make_world(riab)
        plan = ag.make_step_plan(capacity=capacity)
        drift = None
        k = 0
        while k < steps:
            if drift_from is not None and k >= drift_from:
                pos = ag.state_tensor[:2, :ag.n_agents].t()
                drift = 0.2 * (torch.tensor([0.5, 0.5], dtype=torch.float64, device="cuda") - pos)
            plan.step(batch, drift_velocity=drift)
            k += batch
        info = plan.info()
        out = _collect(ag, pops)
        plan.close()
        return out, info
    finally:
        riab._lib.set_option("fused_step", old)


def _w_cfg2_small(B=512, n=200, env_params=None, pop="place", spikes=False, dt=0.01):
    def make(riab):
        np.random.seed(3)
        env = riab.Environment(env_params or {})
        ag = riab.Agent(env, {"n_agents": B, "dt": dt, "seed": 17})
        np.random.seed(4)
        if pop == "place":
            p = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "save_spikes": spikes, "max_fr": 20 if spikes else 1})
        elif pop == "place_dog":
            p = riab.PlaceCells(ag, {"n": n, "wall_geometry": "euclidean", "description": "diff_of_gaussians", "save_spikes": spikes})
        elif pop == "grid":
            p = riab.GridCells(ag, {"n": n, "save_spikes": spikes, "max_fr": 15 if spikes else 1})
        else:
            p = riab.HeadDirectionCells(ag, {"n": n, "save_spikes": spikes, "max_fr": 15 if spikes else 1})
        return env, ag, [p]
    return make


CASES = {
    "open_place": (_w_cfg2_small(), 40, None),
    "open_place_ragged_cells": (_w_cfg2_small(B=256, n=37), 25, None),
    "maze_place_spikes": (_w_cfg2_small(env_params={"walls": MAZE}, spikes=True, dt=0.05), 60, None),
    "periodic_dog": (_w_cfg2_small(env_params={"boundary_conditions": "periodic"}, pop="place_dog", dt=0.05), 40, None),
    "grid_drift": (_w_cfg2_small(B=1024, n=64, pop="grid", spikes=True), 30, 10),
    "hdc": (_w_cfg2_small(B=256, n=48, pop="hdc"), 30, 5),
    "wide_many_cells": (_w_cfg2_small(B=2048, n=3000), 6, None),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_one_launch_step_equals_two_launch_step(riab, case):
    make, steps, drift_from = CASES[case]
    a, ia = _plan_run(riab, True, make, steps, drift_from)
    b, ib = _plan_run(riab, False, make, steps, drift_from)
    assert ia["fused_enabled"] and ia["fused_steps"] == steps and ia["fused_population"] == 0, ia
    assert ia["launches"] == steps, ia                     # ONE kernel per step
    assert ib["fused_steps"] == 0 and ib["launches"] >= 2 * steps, ib
    _same(a, b)
    assert a["traj"].shape[0] == steps and np.isfinite(a["state"][:11]).all()


def test_one_launch_step_rates_match_the_oracle_on_its_own_rows(riab):
    make = _w_cfg2_small(B=512, n=120, env_params={"walls": [[[0.5, 0.0], [0.5, 0.5]]]})
    out, info = _plan_run(riab, True, make, 12)
    assert info["fused_steps"] == 12
    env, ag, pops = make(riab)   # (same seeds: same centres)
    oenv = orc.EnvSpec(walls=[[[0.5, 0.0], [0.5, 0.5]]])
    for t in (0, 5, 11):
        pos = np.stack((out["traj"][t, 0, :512], out["traj"][t, 1, :512]), -1).astype(np.float64)
        ref = orc.place_cells(oenv, pos, pops[0].place_cell_centres, 0.2)
        np.testing.assert_allclose(out["fr0"][t][:, :512], ref, rtol=1e-5, atol=1e-30)


def test_store_bound_populations_ride_in_the_launch_and_the_others_follow(riab):
    """Several populations: every covered one rides in the agent step's launch, every other population is launched
    after it in list order (a FeedForwardLayer reads the fused populations' fresh rows); populations the kernel does not
    cover (line-of-sight place cells, additive noise, boundary vector cells) never do."""
    def make(riab):
        np.random.seed(5)
        env = riab.Environment({"walls": [[[0.5, 0.0], [0.5, 0.35]]]})
        ag = riab.Agent(env, {"n_agents": 256, "dt": 0.02, "seed": 4})
        np.random.seed(6)
        pops = [riab.PlaceCells(ag, {"n": 300, "wall_geometry": "line_of_sight"}),
                riab.HeadDirectionCells(ag, {"n": 8}),
                riab.GridCells(ag, {"n": 96, "save_spikes": True}),
                riab.PlaceCells(ag, {"n": 200, "wall_geometry": "euclidean", "noise_std": 0.1}),
                riab.BoundaryVectorCells(ag, {"n": 12})]
        pops.append(riab.FeedForwardLayer(ag, {"n": 6, "input_layers": [pops[2], pops[1]],
                                               "activation_function": {"activation": "tanh", "gain": 1.0, "threshold": 0.0}}))
        return env, ag, pops
    a, ia = _plan_run(riab, True, make, 20, drift_from=8, capacity=7)   # (chunk rollovers in the middle)
    b, ib = _plan_run(riab, False, make, 20, drift_from=8, capacity=7)
    assert ia["fused_populations"] == [1, 2] and ia["fused_steps"] == 20, ia
    assert ia["launches"] == ib["launches"] - 2 * 20, (ia, ib)        # exactly two kernels less per step
    _same(a, b)


@pytest.mark.parametrize("batch", [1, 5])
def test_cfg5_mix_is_one_launch_plus_the_boundary_vector_cells(riab, batch):
    """BASELINE configs[4]'s mix (Place + Grid + BVC + HeadDirection cells with Poisson spikes; reference loop:
    tests/test_advanced.py:159-176 updates every population after every agent step), at a small size: the three
    store-bound populations are written by the agent step's kernel — their cell groups one after the other on the
    grid's cell axis, ragged last groups included —, the boundary vector cells follow: two kernels per step.  Every
    rate and spike against the kernel-by-kernel plan, bit for bit; the PlaceCells' rows against the oracle."""
    def make(riab):
        np.random.seed(31)
        env = riab.Environment()
        ag = riab.Agent(env, {"n_agents": 1024, "dt": 0.01, "seed": 77})
        np.random.seed(32)
        pops = [riab.PlaceCells(ag, {"n": 203, "wall_geometry": "euclidean", "save_spikes": True, "max_fr": 30}),
                riab.GridCells(ag, {"n": 101, "save_spikes": True, "max_fr": 30}),
                riab.BoundaryVectorCells(ag, {"n": 24, "save_spikes": True, "max_fr": 30}),
                riab.HeadDirectionCells(ag, {"n": 37, "save_spikes": True, "max_fr": 30})]
        return env, ag, pops
    steps = 30
    a, ia = _plan_run(riab, True, make, steps, drift_from=17, capacity=16, batch=batch)
    b, ib = _plan_run(riab, False, make, steps, drift_from=17, capacity=16, batch=batch)
    assert ia["fused_populations"] == [0, 1, 3] and ia["fused_steps"] == steps, ia
    assert ia["launches"] == 2 * steps and ib["launches"] == 5 * steps, (ia, ib)
    _same(a, b)
    assert a["sp0"].any() and a["sp1"].any() and a["sp3"].any()
    env, ag, pops = make(riab)
    for t in (0, steps - 1):
        pos = np.stack((a["traj"][t, 0, :1024], a["traj"][t, 1, :1024]), -1).astype(np.float64)
        np.testing.assert_allclose(a["fr0"][t][:, :1024], 30 * orc.place_cells(orc.EnvSpec(), pos, pops[0].place_cell_centres, 0.2),
                                   rtol=1e-5, atol=1e-30)


def test_more_store_bound_populations_than_the_launch_has_room_for(riab):
    """Five covered populations: the four that write most ride in the launch (list order kept), the smallest follows as
    its own kernel."""
    def make(riab):
        np.random.seed(41)
        env = riab.Environment({"boundary_conditions": "periodic"})
        ag = riab.Agent(env, {"n_agents": 256, "dt": 0.02, "seed": 5})
        np.random.seed(42)
        pops = [riab.GridCells(ag, {"n": 40, "description": "shifted_cosines"}),
                riab.HeadDirectionCells(ag, {"n": 6}),
                riab.PlaceCells(ag, {"n": 90, "description": "gaussian_threshold"}),
                riab.PlaceCells(ag, {"n": 50, "description": "top_hat", "save_spikes": True, "max_fr": 40}),
                riab.GridCells(ag, {"n": 20})]
        return env, ag, pops
    a, ia = _plan_run(riab, True, make, 15)
    b, ib = _plan_run(riab, False, make, 15)
    assert ia["fused_populations"] == [0, 2, 3, 4] and ia["launches"] == 2 * 15 and ib["launches"] == 6 * 15, (ia, ib)
    _same(a, b)


def test_batched_plan_steps_and_no_history(riab):
    def make(riab):
        np.random.seed(9)
        env = riab.Environment()
        ag = riab.Agent(env, {"n_agents": 768, "dt": 0.01, "seed": 2, "save_history": False})
        np.random.seed(10)
        return env, ag, [riab.PlaceCells(ag, {"n": 64, "wall_geometry": "euclidean", "save_history": False})]
    for batch in (1, 8):
        old = riab._lib.set_option("fused_step", 1)
        try:
            env, ag, pops = make(riab)
            plan = ag.make_step_plan()
            for _ in range(24 // batch):
                plan.step(batch)
            info = plan.info()
            torch.cuda.synchronize()
            sa, fa = ag.state_tensor.cpu().numpy(), np.array(pops[0].firingrate)
            plan.close()
            riab._lib.set_option("fused_step", 0)
            env, ag, pops = make(riab)
            plan = ag.make_step_plan()
            for _ in range(24):
                plan.step()
            torch.cuda.synchronize()
            sb, fb = ag.state_tensor.cpu().numpy(), np.array(pops[0].firingrate)
            plan.close()
        finally:
            riab._lib.set_option("fused_step", old)
        assert info["fused_steps"] == 24 and info["launches"] == 24
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(fa, fb)


def test_unchanged_per_step_loop_takes_the_one_launch_step(riab):
    """`Ag.update(); PCs.update()` from Python (reference demos/simple_example.ipynb cell 4): once the automatic
    stepper serves the loop, Agent.update() launches the one kernel — the population's row is written ahead — and
    PlaceCells.update() only moves its cursor; edits between the two calls still reach the rates."""
    def loop(auto, fused):
        os.environ["RIAB_NO_AUTO_PLAN"] = "0" if auto else "1"
        old = riab._lib.set_option("fused_step", 1 if fused else 0)
        try:
            np.random.seed(12)
            env = riab.Environment()
            ag = riab.Agent(env, {"n_agents": 512, "dt": 0.01, "seed": 8})
            np.random.seed(13)
            pcs = riab.PlaceCells(ag, {"n": 128, "wall_geometry": "euclidean", "save_spikes": True, "max_fr": 25})
            hdc = riab.HeadDirectionCells(ag, {"n": 16})
            infos = []
            for t in range(50):
                ag.update()
                if t == 30:
                    pcs.place_cell_centres[3] = [0.1, 0.9]     # an edit between Agent.update() and PlaceCells.update()
                if t != 40:                                    # one step on which the population is not updated
                    pcs.update()
                hdc.update()
                if ag._plan is not None and hasattr(ag._plan, "info"):
                    infos.append(ag._plan.info())
            return _collect(ag, [pcs, hdc]), infos
        finally:
            riab._lib.set_option("fused_step", old)
            os.environ.pop("RIAB_NO_AUTO_PLAN", None)
    a, ia = loop(True, True)
    b, _ = loop(False, False)
    c, ic = loop(True, False)
    assert ia and max(i["fused_steps"] for i in ia) >= 10, ia[-3:]
    assert ic and all(i["fused_steps"] == 0 for i in ic)
    _same(a, b)
    _same(c, b)


def test_a_loop_that_never_updates_the_population_stops_fusing(riab):
    np.random.seed(1)
    env = riab.Environment()
    ag = riab.Agent(env, {"n_agents": 256, "dt": 0.01, "seed": 8})
    pcs = riab.PlaceCells(ag, {"n": 32, "wall_geometry": "euclidean"})
    for _ in range(40):
        ag.update()
    plan = ag._plan
    assert plan is not None and type(plan).__name__ == "AutoStepper"
    assert plan.info()["fused_steps"] <= 3, plan.info()
    pcs.update()
    torch.cuda.synchronize()
    pos = np.asarray(ag.pos, dtype=np.float32).astype(np.float64)
    np.testing.assert_allclose(pcs.firingrate, orc.place_cells(orc.EnvSpec(), pos, pcs.place_cell_centres, 0.2), rtol=1e-5, atol=1e-30)


def test_one_launch_steps_inside_a_captured_graph(riab):
    """the entry point neither allocates nor synchronises nor touches anything process-wide: a block of
    plan steps can be captured in a hipGraph and replayed."""
    np.random.seed(2)
    env = riab.Environment()
    ag = riab.Agent(env, {"n_agents": 512, "dt": 0.01, "seed": 21})
    pcs = riab.PlaceCells(ag, {"n": 96, "wall_geometry": "euclidean"})
    plan = ag.make_step_plan(capacity=64)
    plan.step(4)                                # (code objects resolved outside the capture)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            plan.step(8)
    torch.cuda.synchronize()
    assert plan.info()["fused_steps"] == 12
    g.replay()                                  # the same eight rows again, from the state the capture pass left...
    torch.cuda.synchronize()
    assert np.isfinite(ag.state_tensor.cpu().numpy()[:11]).all()
    plan.close()


def test_cfg2_shape_one_launch_per_step(riab):
    """BASELINE configs[1] stepped through a plan: 4096 agents x 1024 PlaceCells, one kernel per step, equal to the
    two-launch step; properties at full width (agents stay in the box, rates in [0, 1])."""
    def make(riab):
        np.random.seed(0)
        env = riab.Environment()
        ag = riab.Agent(env, {"n_agents": 4096, "dt": 0.01, "seed": 1234})
        np.random.seed(1)
        return env, ag, [riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean", "save_spikes": False})]
    a, ia = _plan_run(riab, True, make, 48, capacity=48)
    b, ib = _plan_run(riab, False, make, 48, capacity=48)
    assert ia["fused_steps"] == 48 and ia["launches"] == 48 and ib["launches"] == 96
    _same(a, b)
    assert (a["traj"][:, :2] > 0).all() and (a["traj"][:, :2] < 1).all()
    assert a["fr0"].min() >= 0 and a["fr0"].max() <= 1


@pytest.mark.parametrize("walls", ["maze", "comb"])
def test_boundary_vector_cells_ray_exchange_in_a_plan_changes_no_bit(riab, walls):
    """A step plan's one-row BVC launches in rooms with interior walls: the workgroups that share a tile each cast a share of
    its rays and read the others' through the exchange rows (csrc/riab_bvc.hip, RiabPopulation.bvc_xch).  Against the eager
    per-step loop (`Ag.update(); N.update()` with the automatic plan off: every workgroup casts every ray): every rate."""
    import os
    import bench

    def run(plan):
        os.environ["RIAB_NO_AUTO_PLAN"] = "1"
        try:
            np.random.seed(8)
            env = riab.Environment({"walls": MAZE if walls == "maze" else bench.comb_walls(60)})
            ag = riab.Agent(env, {"n_agents": 1024, "dt": 0.01, "seed": 3})
            np.random.seed(9)
            pops = [riab.GridCells(ag, {"n": 64}), riab.BoundaryVectorCells(ag, {"n": 96, "save_spikes": True, "max_fr": 20}),
                    riab.BoundaryVectorCells(ag, {"n": 40, "reference_frame": "egocentric"})]
            if plan:
                p = ag.make_step_plan(capacity=16)
                for _ in range(40):
                    p.step()
                p.close()
                p = ag.make_step_plan(capacity=16)       # (a second plan on the same scratch: its counters start over)
                for _ in range(10):
                    p.step()
                p.close()
            else:
                for _ in range(50):
                    ag.update()
                    for q in pops:
                        q.update()
            torch.cuda.synchronize()
            return [x.cpu().numpy() for q in pops for x in q.get_history_tensors()] + [ag.get_history_tensor().cpu().numpy()]
        finally:
            os.environ.pop("RIAB_NO_AUTO_PLAN", None)

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)

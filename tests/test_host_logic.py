"""Host-side logic on CPU tensors (no kernels launched): the parameter protocol, the
Environment's wall table, seeded init-time sampling against the reference's own values
(tests/golden/update_init.npz, G5), history buffers."""
import os
import warnings

import numpy as np
import pytest
import torch

import ratinabox_amd as riab
from ratinabox_amd._history import DeviceHistory
from tests import golden_util as gu

MAZE = [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]
CPU = {"device": "cpu"}


def test_default_params_protocol():
    d = riab.PlaceCells.get_all_default_params()
    assert d["n"] == 10 and d["noise_std"] == 0 and d["description"] == "gaussian" and d["widths"] == 0.2
    assert riab.BoundaryVectorCells.get_all_default_params()["dtheta"] == 2
    assert riab.Agent.get_all_default_params()["dt"] == 0.05
    env = riab.Environment()
    with pytest.warns(UserWarning, match="unexpected params key"):
        riab.Agent(env, dict(CPU, not_a_param=1))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ag = riab.Agent(env, dict(CPU, speed_mean=0.2))
    assert ag.speed_mean == 0.2 and ag.params["speed_mean"] == 0.2 and ag.thigmotaxis == 0.5
    assert [a.name for a in env.Agents] == ["agent_0", "agent_1"]


def test_environment_walls_and_unsupported():
    g = gu.load("update_init.npz")
    env = riab.Environment({"walls": MAZE})
    assert np.array_equal(env.walls, g["init0_maze_walls"])
    assert env.walls.shape == (9, 2, 2) and env.D == 2 and list(env.extent) == [0, 1, 0, 1]
    env.add_wall([[0.1, 0.1], [0.2, 0.2]])
    assert env.walls.shape == (10, 2, 2)
    per = riab.Environment({"boundary_conditions": "periodic"})
    assert per.walls.shape == (0, 2, 2)
    assert riab.Environment({"scale": 2, "aspect": 1.5}).extent.tolist() == [0, 3, 0, 2]
    assert env.flattened_discrete_coords.shape == (10000, 2)
    with pytest.raises(NotImplementedError):
        riab.Environment({"dimensionality": "1D"})
    # a polygonal boundary cannot be periodic: the reference's warning (Environment.py:130-136), and what it announces
    with pytest.warns(UserWarning, match="Changing boundary conditions to 'solid'"):
        tri = riab.Environment({"boundary": [[0, 0], [1, 0], [0, 1]], "boundary_conditions": "periodic"})
    assert tri.boundary_conditions == "solid" and tri.params["boundary_conditions"] == "solid" and tri.walls.shape == (3, 2, 2)


@pytest.mark.parametrize("seed", [0, 7])
def test_seeded_init_matches_reference(seed):
    """Same np.random seed => the same agent start state and cell tables as the reference."""
    g = gu.load("update_init.npz")
    np.random.seed(seed)
    env = riab.Environment()
    ag = riab.Agent(env, CPU)
    np.testing.assert_allclose(ag.pos, g[f"init{seed}_agent_pos"], rtol=0, atol=0)
    np.testing.assert_allclose(ag.velocity, g[f"init{seed}_agent_vel"], rtol=1e-15)
    P = riab.PlaceCells(ag, {"n": 100})
    assert np.array_equal(P.place_cell_centres, g[f"init{seed}_pc_centres"])
    P2 = riab.PlaceCells(ag, {"n": 37, "place_cell_centres": "random"})
    assert np.array_equal(P2.place_cell_centres, g[f"init{seed}_pc_random_centres"])
    G = riab.GridCells(ag, {"n": 32})
    assert np.array_equal(G.gridscales, g[f"init{seed}_gc_gridscales"])
    assert np.array_equal(G.phase_offsets, g[f"init{seed}_gc_phase"])
    assert np.array_equal(G.orientations, g[f"init{seed}_gc_orient"])
    Bv = riab.BoundaryVectorCells(ag, {"n": 20})
    for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles"]:
        np.testing.assert_allclose(getattr(Bv, k), g[f"init{seed}_bvc_{k}"], rtol=1e-15, err_msg=k)
    envw = riab.Environment({"walls": MAZE})
    assert np.array_equal(envw.sample_positions(16, "uniform"), g[f"init{seed}_sample_uniform16"])


def test_bvc_tables_and_hdc():
    g = gu.load("rates.npz")
    env = riab.Environment()
    ag = riab.Agent(env, CPU)
    B = riab.BoundaryVectorCells(ag, {"n": 8})
    assert B.n_test_angles == 180 and B.test_angles[0] == 0 and B.test_angles[1] == 0  # the reference's duplicate
    assert np.array_equal(B.test_angles, g["bvc_open_test_angles"])
    np.testing.assert_allclose(B.test_directions, g["bvc_open_test_directions"], atol=1e-16)
    H = riab.HeadDirectionCells(ag, {"n": 8})
    np.testing.assert_allclose(H.preferred_angles, np.arange(8) * np.pi / 4)
    with pytest.raises(RuntimeError):
        riab.VectorCells(ag)


def test_agent_attribute_shapes_and_setters():
    env = riab.Environment()
    one = riab.Agent(env, CPU)
    assert one.pos.shape == (2,) and np.ndim(one.rotational_velocity) == 0 and one.head_direction.shape == (2,)
    np.testing.assert_allclose(np.linalg.norm(one.velocity), 0.08)
    np.testing.assert_allclose(np.linalg.norm(one.head_direction), 1.0)
    many = riab.Agent(env, dict(CPU, n_agents=6))
    assert many.pos.shape == (6, 2) and many.state_tensor.shape == (12, 8)
    many.pos = np.arange(12.0).reshape(6, 2) / 20
    assert np.array_equal(many.pos, np.arange(12.0).reshape(6, 2) / 20)
    many.rotational_velocity = np.arange(6.0)
    assert np.array_equal(many.rotational_velocity, np.arange(6.0))
    one.pos = [0.3, 0.4]
    assert one.pos.tolist() == [0.3, 0.4]


def test_motion_parameter_resolution():
    """Which values come from kwargs and which from attributes (Agent.py:280-285, 310, 340, 375, 439)."""
    env = riab.Environment()
    ag = riab.Agent(env, dict(CPU, speed_std=0.0))
    m = ag._motion(0.02, True, 3.0, {"speed_mean": 0.5, "thigmotaxis": 0.9, "rotational_velocity_std": 1.0})
    assert m.speed_mean_kw == 0.5 and m.speed_mean == 0.08 and m.speed_std_is_zero == 1
    assert m.thigmotaxis_kw == 0.9 and m.wall_repel_distance_kw == 0.1
    np.testing.assert_allclose(m.drift_theta, 3.0 / 0.7)
    np.testing.assert_allclose(m.rot_sigma_kw, np.sqrt(2 * 1.0 / (0.08 * 0.02)))
    np.testing.assert_allclose(m.speed_sigma_kw, np.sqrt(2 / (0.7 * 0.02)))


def test_device_history_buffers():
    h = DeviceHistory((3, 4), torch.float32, torch.device("cpu"), chunk_bytes=3 * 4 * 4 * 5)
    assert h.chunk_rows == 5
    for i in range(12):
        h.reserve(1)[:] = i
    big = h.reserve(7)
    big[:] = 99
    st = h.stack()
    assert st.shape == (19, 3, 4) and st[:12, 0, 0].tolist() == list(range(12)) and (st[12:] == 99).all()
    assert h.last()[0, 0] == 99 and len(h) == 19
    h.preallocate(10)
    n_chunks = len(h.chunks)
    h.reserve(10)
    assert len(h.chunks) == n_chunks  # used the preallocated rows
    h.reset()
    assert len(h) == 0 and h.stack().shape == (0, 3, 4)
    # a plan that opens the same number of rows in every history caps it by the shortest free tail — a
    # history whose chunk is FULL then opens a full-sized chunk and hands out a short view, not a short chunk
    h.preallocate(6)
    h.reserve(6)
    assert h.free_rows() == 0
    v = h.open_rows(3, 50)
    assert v.shape[0] == 3 and h.chunks[-1].shape[0] == 50 and h.free_rows() == 50
    h.commit(3)
    assert h.open_rows(100, 50).shape[0] == 47 and len(h) == 9      # (the rest of that chunk, however large the request)


def test_samplers():
    np.random.seed(1)
    m = riab.utils.distribution_sampler("modules", (0.3, 0.5, 0.8), (10,))
    assert m.tolist() == [0.3] * 3 + [0.5] * 3 + [0.8] * 4
    assert riab.utils.distribution_sampler("delta", 2.0, (3,)).tolist() == [2.0] * 3
    lg = riab.utils.distribution_sampler("logarithmic", (0.1, 10), (3,))
    np.testing.assert_allclose(lg, [0.1, 1, 10])
    with pytest.raises(ValueError):
        riab.utils.distribution_sampler("nope", (1,), (3,))
    d, a, sd, sa = riab.utils.create_random_assembly(tuning_distance=[0.1, 0.2], sigma_angle=(20, 20),
                                                     sigma_angle_distribution="delta")
    assert len(d) == 2 and np.allclose(sd, 0.08 + np.array([0.1, 0.2]) / 12) and np.allclose(sa, np.radians(20))


def test_field_of_view_manifolds_match_reference():
    g = gu.load("rates.npz")
    env = riab.Environment({"walls": MAZE})
    ag = riab.Agent(env, CPU)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # no spurious "ignoring n" / unknown-key warnings
        div = riab.FieldOfViewBVCs(ag)
        uni = riab.FieldOfViewBVCs(ag, {"cell_arrangement": "uniform_manifold", "distance_range": [0.05, 0.3],
                                        "angle_range": [0, 120], "spatial_resolution": 0.05})
    for tag, X in (("div", div), ("uni", uni)):
        assert X.reference_frame == "egocentric" and X.n == len(g[f"fov_{tag}_tuning_distances"])
        for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles"]:
            assert np.array_equal(getattr(X, k), g[f"fov_{tag}_{k}"]), (tag, k)
    with pytest.warns(UserWarning, match="Ignoring 'n'"):
        riab.FieldOfViewBVCs(ag, {"n": 7})


def test_simulate_chunk_schedule():
    """Chunks handed to the fused path add up and never exceed `chunk`."""
    from ratinabox_amd.Agent import Agent
    for n, c in [(1024, 128), (128, 128), (100, 128), (10, 256), (300, 128), (1, 128), (1000, 256), (257, 128), (40, 16)]:
        s = Agent._chunk_schedule(n, c)
        assert sum(s) == n and all(0 < x <= c for x in s)
    assert Agent._chunk_schedule(1024, 128) == [128] * 8
    assert Agent._chunk_schedule(300, 128) == [128, 128, 44]
    assert Agent._chunk_schedule(0, 128) == []


def test_small_api_helpers_match_reference():
    """return_list_of_neurons, GridCells grid phase offsets (reference Neurons.py:779-810, 1238-1256)."""
    np.random.seed(0)
    env = riab.Environment()
    Ag = riab.Agent(env, dict(CPU))
    np.random.seed(4)
    GCs = riab.GridCells(Ag, {"n": 11, "phase_offset_distribution": "grid"})
    assert GCs.phase_offsets.shape == (11, 2)
    dx = 2 * np.pi / 3
    assert np.allclose(GCs.phase_offsets[:9, 0], np.repeat([dx / 2, 3 * dx / 2, 5 * dx / 2], 3))
    assert np.allclose(GCs.phase_offsets[:9, 1], np.tile([dx / 2, 3 * dx / 2, 5 * dx / 2], 3))
    assert np.all((GCs.phase_offsets[9:] >= 0) & (GCs.phase_offsets[9:] < 2 * np.pi))
    assert GCs.return_list_of_neurons("all") == list(range(11)) or list(GCs.return_list_of_neurons("all")) == list(range(11))
    assert list(GCs.return_list_of_neurons("3")) == [0, 5, 10]
    assert GCs.return_list_of_neurons([1.0, 4.0]) == [1, 4]
    assert len(set(GCs.return_list_of_neurons("4rand"))) == 4


def test_agent_registry_lookup():
    """Environment.Agents / agents_dict / agent_lookup / remove_agent (reference Environment.py:220-328)."""
    np.random.seed(0)
    env = riab.Environment()
    a = riab.Agent(env, dict(CPU))
    b = riab.Agent(env, dict(CPU, name="rat"))
    assert [x.name for x in env.Agents] == ["agent_0", "rat"] and (a.agent_idx, b.agent_idx) == (0, 1)
    assert env.agent_lookup("rat") == [b] and env.agent_lookup(["rat", "agent_0"]) == [b, a]
    assert env.agent_lookup(None) is None
    with pytest.raises(ValueError):
        env.agent_lookup("mouse")
    with pytest.warns(UserWarning):
        c = riab.Agent(env, dict(CPU, name="rat"))
    assert c.name == "agent_2"
    env.remove_agent("rat")
    assert env.Agents == [a, c] and "rat" not in env.agents_dict


def test_bvc_direction_windows_cover_every_significant_term():
    """Host side of the BVC direction windows (Neurons.BoundaryVectorCells._call): the regrouped table rows are
    a permutation of the cells, every window is whole quads of directions, and the directions outside a row's
    window carry at most BVC_WINDOW_SHARE = 1e-6 of the row's summed von Mises weight — the bound on the change of
    the normalised rate (rate = sum_k g_k w_k / sum_k w_k with g_k <= 1)."""
    np.random.seed(3)
    env = riab.Environment()
    Ag = riab.Agent(env, dict(CPU))
    for n in (256, 37, 5):
        BVs = riab.BoundaryVectorCells(Ag, {"n": n})
        d = BVs._call(None, None)
        K = d["K"]
        rows, win, vm = d["cell_rows"].numpy(), d["windows"].numpy(), d["vm_table"].numpy()
        assert sorted(rows.tolist()) == list(range(n)) and win.shape == ((n + 3) // 4, 2)
        assert (win % 4 == 0).all() and (win[:, 0] >= 0).all() and (win[:, 0] < K).all()
        assert (win[:, 1] > 0).all() and (win[:, 1] <= K).all()
        # the table rows are in regrouped order: row i belongs to cell rows[i]
        kappa = 1 / np.asarray(BVs.sigma_angles) ** 2
        diff = np.asarray(BVs.test_angles)[None, :] - np.asarray(BVs.tuning_angles)[:, None]
        full = (np.log2(np.e) * kappa[:, None] * (np.cos(diff) - 1))[rows]
        np.testing.assert_allclose(vm[:, :K], full, rtol=1e-5, atol=1e-5)
        for i in range(n):
            k0, length = win[i // 4]
            inside = ((np.arange(K) - k0) % K) < length
            w = np.exp2(full[i])
            assert w[~inside].sum() <= BVs.BVC_WINDOW_SHARE * w.sum() * (1 + 1e-9), (n, i)
        if n == 256:
            assert win[:, 1].mean() < 0.82 * K and BVs._window_stats["cells_need"] < 0.75


def test_velocity_speed_and_agent_vector_cell_construction():
    """Host side of VelocityCells / SpeedCell / AgentVectorCells / FieldOfViewAVCs (reference Neurons.py:2151-2651):
    defaults, the fixed single speed cell, the speed scale taken at construction, the manifold of the
    field-of-view cells, and the lane-matching rule of the batched extension."""
    np.random.seed(0)
    env = riab.Environment()
    Ag = riab.Agent(env, dict(CPU, speed_mean=0.1, speed_std=0.04))
    VCs = riab.VelocityCells(Ag, {"n": 6})
    assert VCs.n == 6 and np.isclose(VCs.one_sigma_speed, 0.14) and VCs.name == "VelocityCells"
    np.testing.assert_allclose(VCs.preferred_angles, np.linspace(0, 2 * np.pi, 7)[:-1])
    with pytest.warns(UserWarning):
        SC = riab.SpeedCell(Ag, {"n": 4})
    assert SC.n == 1 and SC.firingrate.shape == (1,) and np.isclose(SC.one_sigma_speed, 0.14)
    Ag.speed_mean = 0.3  # the scale was fixed when the cells were made
    assert np.isclose(VCs.one_sigma_speed, 0.14)
    other = riab.Agent(env, dict(CPU))
    AV = riab.AgentVectorCells(Ag, other, {"n": 7})
    assert AV.n == 7 and AV.wall_geometry == "line_of_sight" and AV.tuning_type_agent is other
    assert riab.AgentVectorCells(Ag, other, {"walls_occlude": False}).wall_geometry == "euclidean"
    FA = riab.FieldOfViewAVCs(Ag, other, {"spatial_resolution": 0.1})
    assert FA.reference_frame == "egocentric" and FA.n == len(FA.tuning_distances) > 3
    many = riab.Agent(env, dict(CPU, n_agents=8))
    with pytest.raises(ValueError):  # 8 lanes cannot be matched with 3
        riab.AgentVectorCells(many, riab.Agent(env, dict(CPU, n_agents=3)))
    assert riab.AgentVectorCells(many, other).n == 10  # a single-agent Other_Agent is seen by every lane
    d = riab.VelocityCells.get_all_default_params()
    assert d["name"] == "VelocityCells" and "angular_spread_degrees" in d and "noise_std" in d


def test_history_view_sees_rows_a_plan_commits_lazily():
    """a HistoryView cached on DeviceHistory.version missed rows a step plan had written but not yet
    committed (the plan publishes in sync(), which ran only inside the materialise call — after the version had
    been compared).  The owners' version callables now publish first."""
    from ratinabox_amd._history import HistoryView
    hist = DeviceHistory((2, 4), torch.float32, torch.device("cpu"), chunk_bytes=1 << 12)
    pending = {"n": 0}
    calls = {"materialise": 0}

    def sync():  # what StepPlan.sync() does
        n, pending["n"] = pending["n"], 0
        hist.commit(n)

    def materialise():
        calls["materialise"] += 1
        sync()
        return {"x": hist.stack().numpy().copy()}

    view = HistoryView(("x",), materialise, lambda: (sync(), hist.version)[1])
    rows = hist.open_rows(10)
    rows[:5] = 1.0
    pending["n"] = 5
    assert view["x"].shape[0] == 5
    rows[5:10] = 2.0
    pending["n"] = 5  # the plan stepped five more times; nothing has bumped the version yet
    assert view["x"].shape[0] == 10 and float(view["x"][9, 0, 0]) == 2.0
    n = calls["materialise"]
    assert view["x"].shape[0] == 10 and calls["materialise"] == n, "an unchanged history is served from the cache"


def test_second_agent_object_gets_its_own_rng_key():
    """Philox streams are keyed by (seed, global agent id, population, step); two Agent objects of one
    Environment with the default seed and id range must not replay each other's noise."""
    env = riab.Environment()
    a0 = riab.Agent(env, CPU)
    a1 = riab.Agent(env, CPU)
    a2 = riab.Agent(env, dict(CPU, seed=5))
    assert a0.rng_seed == a0.seed == 0
    assert a1.rng_seed != a0.rng_seed and 0 <= a1.rng_seed < 2 ** 64
    assert a2.rng_seed == 5, "an explicit seed is taken as given"
    # shards of one logical population carry explicit ids: same key wherever they are constructed
    assert riab.Agent(env, dict(CPU, agent_id0=4096)).rng_seed == 0
    with pytest.warns(UserWarning, match="IDENTICAL noise"):
        riab.Agent(env, dict(CPU, seed=5))


def test_bench_refuses_a_mismatched_world_size():
    """`--gpus N` with a different WORLD_SIZE must not print a line for the wrong rank count."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and p.stdout.strip() == "" and "WORLD_SIZE=2" in p.stderr


@pytest.mark.parametrize("tag", ["lroom", "holes", "both"])
def test_polygon_environment_host_side(tag):
    """Row a6 on the host: walls in the reference's order (boundary edges, user walls, hole edges), the strict
    inside test point by point, and the three samplers drawing from np.random in the reference's order
    (tests/golden/polygon.npz, generated by the reference)."""
    g = gu.load("polygon.npz")
    params = dict(gu.product_env_params(g, tag + "_"), walls=g[f"{tag}_user_walls"].tolist())
    env = riab.Environment(params)
    assert np.array_equal(env.walls, g[f"{tag}_walls"])
    np.testing.assert_array_equal(env.extent, g[f"{tag}_extent"])
    assert env.is_rectangular == (tag == "holes") and env.has_holes == (tag != "lroom")
    got = np.array([env.check_if_position_is_in_environment(p) for p in g[f"{tag}_points"]])
    assert np.array_equal(got, g[f"{tag}_inside"])
    for method in ("random", "uniform", "uniform_jitter"):
        np.random.seed(77)
        np.testing.assert_array_equal(env.sample_positions(n=53, method=method), g[f"{tag}_sample_{method}"])
    # hole edges added later keep their flag wherever they land in the table
    env.add_wall([[0.05, 0.05], [0.06, 0.3]])
    env.add_hole([[0.3, 0.05], [0.34, 0.05], [0.32, 0.09]])
    assert env._wall_is_hole[-4:] == [False, True, True, True] and len(env._wall_is_hole) == len(env.walls)
    assert not env.check_if_position_is_in_environment([0.32, 0.06])


def test_public_helpers_vs_reference():
    """ratinabox_amd.utils carries the reference's public geometry / statistics helpers (utils.py: vector_intercepts,
    shortest_vectors_from_points_to_lines, get_*_between, get_angle / get_bearing on vectors and segments, wall_bounce,
    pi_domain, ornstein_uhlenbeck, the Rayleigh <-> normal maps, gaussian, von_mises, activate and its derivatives)
    for user code written against `ratinabox.utils`: their outputs on the fixture's inputs (tests/golden/helpers.npz,
    made by importing the reference: make_golden.py `helpers`)."""
    from ratinabox_amd import utils as U
    g = gu.load("helpers.npz")
    a, b, p, q, x, th, vs, segs = (g[k] for k in ("a", "b", "p", "q", "x", "th", "vs", "segs"))
    close = lambda got, key, **kw: np.testing.assert_allclose(got, g[key], **dict(dict(rtol=1e-12, atol=1e-14), **kw))  # noqa: E731
    close(U.vector_intercepts(a, b), "vi")
    assert np.array_equal(U.vector_intercepts(a, b, return_collisions=True), g["vi_hit"])
    both = U.vector_intercepts(a, b, return_collisions="as_well")
    close(both[0], "vi"), np.array_equal(both[1], g["vi_hit"])
    close(U.shortest_vectors_from_points_to_lines(p, b), "sv")
    assert np.array_equal(U.get_line_segments_between(p, q), g["segs_pq"])
    assert np.array_equal(U.get_vectors_between(p, q), g["vec_pq"]) and np.array_equal(U.get_distances_between(p, q), g["dist_pq"])
    close(U.get_angle(vs, is_array=True), "angle_vs"), close(U.get_angle(segs, is_array=True), "angle_segs")
    close(U.get_bearing(vs, is_array=True), "bearing_vs"), close(U.get_bearing(segs, is_array=True), "bearing_segs")
    assert np.isclose(U.get_angle(segs[0]), g["angle_segs"][0]) and np.isclose(U.get_bearing(vs[2]), g["bearing_vs"][2])
    close(U.get_perpendicular(vs), "perp_vs")
    close(np.stack([U.wall_bounce(v, w) for v, w in zip(vs, segs)]), "bounce")
    close(U.pi_domain(x), "pi_domain")
    z = g["ou_z"]
    saved, np.random.normal = np.random.normal, (lambda loc=0.0, scale=1.0, size=None: z * scale)
    try:
        close(U.ornstein_uhlenbeck(0.01, x, drift=0.5, noise_scale=0.3, coherence_time=0.7), "ou")
    finally:
        np.random.normal = saved
    close(U.normal_to_rayleigh(x / 3, 0.08), "n2r")
    close(U.rayleigh_to_normal(g["speeds"], 0.08), "r2n")
    for k, norm in (("d", None), ("1", 1), ("2p5", 2.5)):
        close(U.gaussian(th, 0.4, 0.3, norm), "gauss_" + k), close(U.von_mises(th, 0.4, 0.3, norm), "vm_" + k)
    for name in ("linear", "sigmoid", "relu", "tanh", "retanh", "softmax"):
        oa = {"max_fr": 3, "min_fr": 0.5, "mid_x": 0.2, "width_x": 1.5} if name == "sigmoid" else {"gain": 2.0, "threshold": 0.3}
        for tag, args in (("dflt", {}), ("args", oa)):
            for deriv in (False, True):
                close(U.activate(x / 3, name, deriv, dict(args)), f"act_{name}_{tag}_{int(deriv)}")
    assert U.activate(x, other_args={"function": lambda x, deriv: 7}) == 7
    close(U.activate(x / 3, "relu", False, {"activation": "tanh"}), "act_tanh_dflt_0")
    # two public methods of the vector cells (host side): the ray -> wall preference and the tuning tables
    assert np.array_equal(riab.BoundaryVectorCells.boundary_vector_preference_function(None, g["pref_lam"]), g["pref"])
    np.random.seed(1)
    bv = riab.BoundaryVectorCells(riab.Agent(riab.Environment(), CPU), {"n": 5})
    four = bv.set_tuning_parameters(**bv.params)
    assert len(four) == 4 and all(isinstance(v, np.ndarray) and len(v) == 5 for v in four)
    bv.cell_arrangement = lambda **kw: ([0.1, 0.2], [0.0, 1.0], [0.05, 0.06], [0.2, 0.3])
    assert [list(v) for v in bv.set_tuning_parameters()] == [[0.1, 0.2], [0.0, 1.0], [0.05, 0.06], [0.2, 0.3]]
    bv.cell_arrangement = "hexagonal"
    with pytest.raises(ValueError):
        bv.set_tuning_parameters()


def test_bench_pins_ranks_to_their_gpus_numa_node(tmp_path, monkeypatch):
    """every rank of a multi-GPU bench line runs on cores of its GPU's NUMA node, and ranks whose GPUs share
    a node get disjoint sets of WHOLE cores — worked out from sysfs alone (no HIP call).  A fake sysfs tree: 8 GPUs on 2
    nodes (KFD nodes 2-9 after two CPU nodes), 16 cores x 2 hardware threads per node."""
    import bench
    sysfs = tmp_path / "sys"
    kfd = sysfs / "class/kfd/kfd/topology/nodes"
    for node in (0, 1):   # CPU nodes
        (kfd / str(node)).mkdir(parents=True)
        (kfd / str(node) / "properties").write_text("cpu_cores_count 32\nsimd_count 0\n")
    cpus_of = {0: list(range(0, 16)) + list(range(32, 48)), 1: list(range(16, 32)) + list(range(48, 64))}
    for g in range(8):
        numa = 0 if g < 4 else 1
        (kfd / str(2 + g)).mkdir(parents=True)
        (kfd / str(2 + g) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
        pci = sysfs / f"devices/pci0000:00/0000:{g:02x}:00.0"
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text(f"{numa}\n")
        (pci / "local_cpulist").write_text(bench._format_cpulist(cpus_of[numa]) + "\n")
        drm = sysfs / f"class/drm/renderD{128 + g}"
        drm.mkdir(parents=True)
        os.symlink(pci, drm / "device")
    for c in range(64):
        d = sysfs / f"devices/system/cpu/cpu{c}/topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(f"{c % 32},{c % 32 + 32}\n")
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    gpus = bench.gpu_numa_nodes(str(sysfs))
    assert [g[0] for g in gpus] == [0, 0, 0, 0, 1, 1, 1, 1] and gpus[5][2] == "0000:05:00.0"
    got = [bench.bind_rank_to_gpu_numa(r, 8, str(sysfs), apply=False) for r in range(8)]
    sets = [set(bench._parse_cpulist(b["cpus"])) for b in got]
    for r, b in enumerate(got):
        assert b["binding"] == "numa" and b["numa_node"] == (0 if r < 4 else 1) and b["ranks_on_this_node"] == 4
        assert b["n_cpus"] == 8 and sets[r] <= set(cpus_of[b["numa_node"]])
        assert all((c % 32) in sets[r] and (c % 32 + 32) in sets[r] for c in sets[r]), "a core was split between ranks"
        for q in range(r):
            assert not (sets[r] & sets[q]), "two ranks share a core"
    # the runtime's device selection is applied before the lookup
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6,1")
    assert [g[2] for g in bench.gpu_numa_nodes(str(sysfs))] == ["0000:06:00.0", "0000:01:00.0"]
    two = [bench.bind_rank_to_gpu_numa(r, 2, str(sysfs), apply=False) for r in range(2)]
    assert [b["numa_node"] for b in two] == [1, 0] and all(b["n_cpus"] == 32 and b["ranks_on_this_node"] == 1 for b in two)
    # anything unexpected: unbound, with the reason, never an exception
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")
    assert bench.bind_rank_to_gpu_numa(0, 1, str(sysfs), apply=False)["binding"] == "none"
    assert bench.bind_rank_to_gpu_numa(0, 1, str(tmp_path / "nothing"), apply=False)["binding"] == "none"
    assert bench._parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and bench._format_cpulist([3, 1, 2, 7]) == "1-3,7"


def test_wall_grid_masks_are_supersets_of_what_a_step_can_need():
    """Environment.wall_grid (the broad phase of the motion step in wall-heavy rooms, include/riab_hip.h RiabMotion.wall_grid):
    for random points of a 64-wall comb maze and of a room of random walls, the cell's first mask holds the nearest wall
    and every wall within the repel distance, its second mask every wall a step of at most `lmax` from the point crosses —
    brute force over all walls in float64."""
    import ratinabox_amd as riab
    import bench
    rs = np.random.RandomState(5)
    rooms = [bench.comb_walls(60), [[list(rs.uniform(0, 1, 2)), list(rs.uniform(0, 1, 2))] for _ in range(40)]]
    for walls in rooms:
        env = riab.Environment({"walls": walls})
        wd, lmax = 0.1, 0.02
        tab, G, wd_, lmax_ = env.wall_grid("cpu", wd, lmax)
        tab = tab.numpy().view(np.uint64)
        W = np.asarray(env.walls, float).reshape(-1, 4)
        assert tab.shape == (G * G, 2) and (wd_, lmax_) == (wd, lmax) and len(W) <= 64
        P = rs.uniform(0, 1, (4000, 2))
        P[:500] = np.clip(np.round(P[:500] * G) / G + rs.choice([-1e-12, 0, 1e-12], (500, 2)), 0, 1)   # on the cell borders
        a, s_ = W[:, :2], W[:, 2:] - W[:, :2]
        lam = np.clip(((P[:, None] - a[None]) * s_[None]).sum(-1) / (s_ ** 2).sum(1)[None], 0, 1)
        d = np.linalg.norm(P[:, None] - (a[None] + lam[..., None] * s_[None]), axis=-1)
        ix = np.clip((P[:, 0] * G).astype(int), 0, G - 1)
        iy = np.clip((P[:, 1] * G).astype(int), 0, G - 1)
        near, coll = tab[iy * G + ix, 0], tab[iy * G + ix, 1]
        bit = lambda m, w: (m >> np.uint64(w)) & np.uint64(1)   # noqa: E731
        nearest = d.argmin(1)
        assert all(bit(near[i], nearest[i]) for i in range(len(P)))
        for w in range(len(W)):
            inside = d[:, w] <= wd * (1 + 1e-6)
            assert bit(near[inside], w).all(), w
            assert bit(coll[d[:, w] <= lmax], w).all(), w      # (a step of length <= lmax can only cross a wall that close)
        # the masks do cull: far fewer walls than the room has
        assert np.mean([bin(int(m)).count("1") for m in near]) < 0.6 * len(W)
        assert np.mean([bin(int(m)).count("1") for m in coll]) < 0.35 * len(W)

"""Pin the CPU oracle (oracle/riab_oracle.py) against golden vectors produced by
the reference itself (tests/golden/make_golden.py).  float64 vs float64."""
import numpy as np
import pytest

from oracle import riab_oracle as orc
from tests import golden_util as gu

RT = 1e-11


@pytest.mark.parametrize("fname", gu.MOTION_FILES)
def test_single_steps(fname):
    g = gu.load(fname)
    env = gu.env_from(g)
    assert np.array_equal(env.walls, g["ref_walls"])
    p, kw, dt = gu.params_from(g)
    st = gu.state_from_rows(g["pre"])
    drift = g["drift"] if g["drift"].shape[0] else None
    out = orc.agent_step(env, st, dt, g["z"][:, 0], g["z"][:, 1], params=p, kwargs=kw, drift_velocity=drift,
                         drift_to_random_strength_ratio=float(g["drift_ratio"]), resample_pos=g["resample"])
    post = g["post"]
    assert np.array_equal(out["n_bounces"], g["n_bounces"])
    assert np.array_equal(out["bc_applied"], g["bc_applied"] > 0)
    # the steps that drew a random replacement are exactly the ones the oracle sends to the resample branch
    assert np.array_equal(np.isfinite(g["resample"][:, 0]), out["bc_applied"] & orc.env_needs_resample(env, out["pos_before_bc"]))
    for k, s in gu.PRE_SLICES.items():
        np.testing.assert_allclose(out[k], post[:, s], rtol=RT, atol=1e-13, err_msg=k)
    # measured rotational velocity is an angle difference / dt: absolute tolerance
    np.testing.assert_allclose(out["measured_rotational_velocity"], post[:, 10], rtol=1e-9, atol=1e-8)
    fin = np.isfinite(post[:, 11])
    np.testing.assert_allclose(out["distance_to_closest_wall"][fin], post[fin, 11], rtol=RT)


@pytest.mark.parametrize("fname", gu.MOTION_FILES)
def test_rollout(fname):
    """G3: replay the reference's noise stream for every agent; the whole rollout
    must track the reference (float64, same discrete decisions)."""
    g = gu.load(fname)
    if g["drift"].shape[0]:
        pytest.skip("drift schedule is per (step, agent); covered by single steps")
    env = gu.env_from(g)
    p, kw, dt = gu.params_from(g)
    st = gu.state_from_rows(g["roll_state0"])
    T = g["roll_z"].shape[0]
    for t in range(T):
        tele = np.isfinite(g["roll_teleport"][t, :, 0])  # (the generator put these agents somewhere else first)
        st["pos"] = np.where(tele[:, None], g["roll_teleport"][t], st["pos"])
        st = orc.agent_step(env, st, dt, g["roll_z"][t, :, 0], g["roll_z"][t, :, 1], params=p, kwargs=kw,
                            resample_pos=g["roll_resample"][t])
        np.testing.assert_allclose(st["pos"], g["roll_pos"][t + 1], rtol=1e-8, atol=1e-10, err_msg=f"step {t}")
    np.testing.assert_allclose(st["head_direction"], g["roll_final"][:, 7:9], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(st["distance_travelled"], g["roll_final"][:, 9], rtol=1e-8)


def _rates():
    return gu.load("rates.npz")


@pytest.mark.parametrize("desc", ["gaussian", "gaussian_threshold", "diff_of_gaussians", "one_hot", "top_hat"])
def test_place_cells_descriptions(desc):
    g = _rates()
    env = orc.EnvSpec()
    got = orc.place_cells(env, g["pos"], g[f"pc_{desc}_centres"], g[f"pc_{desc}_widths"], description=desc,
                          min_fr=0.1, max_fr=2.0, widths_scalar=0.2)
    np.testing.assert_allclose(got, g[f"pc_{desc}_rates"], rtol=1e-12, atol=1e-300)


def test_place_cells_geometries():
    g = _rates()
    got = orc.place_cells(orc.EnvSpec(), g["pos"][:64], g["pc_big_centres"], 0.2)
    np.testing.assert_allclose(got, g["pc_big_rates"], rtol=1e-12, atol=1e-300)
    maze = orc.EnvSpec(walls=g["maze_walls"][4:])
    got = orc.place_cells(maze, g["pos"], g["pc_los_centres"], 0.25, wall_geometry="line_of_sight")
    np.testing.assert_allclose(got, g["pc_los_rates"], rtol=1e-12, atol=1e-300)
    geo = orc.EnvSpec(walls=g["geo_walls"][4:])
    got = orc.place_cells(geo, g["pos"], g["pc_geo_centres"], 0.2, description="gaussian_threshold",
                          wall_geometry="geodesic")
    np.testing.assert_allclose(got, g["pc_geo_rates"], rtol=1e-12, atol=1e-15)
    per = orc.EnvSpec(boundary_conditions="periodic")
    got = orc.place_cells(per, g["pos"], g["pc_per_centres"], 0.15)
    np.testing.assert_allclose(got, g["pc_per_rates"], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("tag,kw", [("rectified_cosines", dict(description="rectified_cosines", max_fr=1.5)),
                                    ("shifted_cosines", dict(description="shifted_cosines", max_fr=1.5)),
                                    ("rand", dict(width_ratio=0.5))])
def test_grid_cells(tag, kw):
    g = _rates()
    w = orc.grid_cell_w(g[f"gc_{tag}_orientations"])
    np.testing.assert_allclose(w, g[f"gc_{tag}_w"], rtol=0, atol=1e-16)
    got = orc.grid_cells(g["pos"], g[f"gc_{tag}_gridscales"], g[f"gc_{tag}_phase_offsets"], w, **kw)
    np.testing.assert_allclose(got, g[f"gc_{tag}_rates"], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("tag", ["open", "maze"])
def test_bvc_allocentric(tag):
    g = _rates()
    angles, dirs = orc.bvc_test_angles(2)
    assert np.array_equal(angles, g[f"bvc_{tag}_test_angles"])
    np.testing.assert_allclose(dirs, g[f"bvc_{tag}_test_directions"], rtol=0, atol=1e-16)
    np.testing.assert_allclose(orc.bvc_fr_norm(angles, g[f"bvc_{tag}_sigma_angles"]), g[f"bvc_{tag}_cell_fr_norm"],
                               rtol=1e-12)
    walls = orc.EnvSpec().walls if tag == "open" else g["maze_walls"]
    got = orc.bvc(g["pos"], walls, g[f"bvc_{tag}_tuning_distances"], g[f"bvc_{tag}_tuning_angles"],
                  g[f"bvc_{tag}_sigma_distances"], g[f"bvc_{tag}_sigma_angles"])
    np.testing.assert_allclose(got, g[f"bvc_{tag}_rates"], rtol=1e-10, atol=1e-14)


def test_sixty_four_walls():
    """A room at RIAB_MAX_WALLS (tests/golden/make_golden.py make_walls64: a comb maze of 60 interior segments): boundary
    vector cells, line-of-sight PlaceCells and the wall vectors of the oracle against the reference.  (The motion steps of
    the same room are motion_comb60_*.npz: test_motion_single_steps / rollouts pick them up by name.)"""
    g = gu.load("walls64.npz")
    env = orc.EnvSpec(walls=g["walls"])
    np.testing.assert_array_equal(env.walls, g["ref_walls"])
    assert len(env.walls) == 64
    got = orc.bvc(g["pos"], env.walls, g["bvc_tuning_distances"], g["bvc_tuning_angles"], g["bvc_sigma_distances"],
                  g["bvc_sigma_angles"])
    np.testing.assert_allclose(got, g["bvc_rates"], rtol=1e-10, atol=1e-14)
    got = orc.place_cells(env, g["pos"], g["pc_los_centres"], 0.12, wall_geometry="line_of_sight")
    np.testing.assert_allclose(got, g["pc_los_rates"], rtol=1e-11, atol=1e-13)


def test_bvc_egocentric():
    g = _rates()
    got = orc.bvc(g["pos"][:48], g["maze_walls"], g["bvc_ego_tuning_distances"], g["bvc_ego_tuning_angles"],
                  g["bvc_ego_sigma_distances"], g["bvc_ego_sigma_angles"], head_direction=g["hd"][:48],
                  min_fr=0.5, max_fr=3.0)
    np.testing.assert_allclose(got, g["bvc_ego_rates"], rtol=1e-10, atol=1e-14)


def test_head_direction_cells():
    g = _rates()
    got = orc.head_direction_cells(g["hd"], 24, angular_spread_degrees=30, min_fr=0.25, max_fr=2.0)
    np.testing.assert_allclose(got, g["hdc_rates"], rtol=1e-12)


def test_update_noise_and_spikes():
    """Neurons.update end to end (Neurons.py:145-171, 681-687): rates + OU noise, spikes."""
    g = gu.load("update_init.npz")
    dt = float(g["upd_dt"])
    noise = np.zeros(50)
    env = orc.EnvSpec()
    for t in range(g["upd_pos"].shape[0]):
        noise = noise + orc.ou_increment(noise, dt, 0.0, 0.5, 0.2, g["upd_z"][t])
        fr = orc.place_cells(env, g["upd_pos"][t][None], g["upd_centres"], 0.2, max_fr=40.0)[:, 0] + noise
        np.testing.assert_allclose(noise, g["upd_noise"][t], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(fr, g["upd_fr"][t], rtol=1e-11, atol=1e-13)
        assert np.array_equal(orc.spikes_ref(fr, g["upd_u"][t], dt), g["upd_spikes"][t])
    # the fp32 spike rule agrees with the float64 reference wherever the margin is not razor thin
    assert float(g["upd_min_rel_margin"]) > 1e-5
    assert np.array_equal(orc.spikes_f32(g["upd_fr"].astype(np.float32), g["upd_u"].astype(np.float32), dt),
                          g["upd_spikes"])


def test_philox_known_answers():
    """Random123 known-answer vectors for Philox4x32-10."""
    assert [int(x) for x in orc.philox4x32_10(0, 0, 0, 0, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert [int(x) for x in orc.philox4x32_10(f, f, f, f, f, f)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    got = orc.philox4x32_10(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
    assert [int(x) for x in got] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    z = np.concatenate(orc.motion_normals(1234, 7, np.arange(50000))[:2])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    u = orc.spike_uniforms(1, 3, 0, 64, 1024)
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


@pytest.mark.parametrize("tag", ["div", "uni"])
def test_field_of_view_bvcs(tag):
    """FieldOfViewBVCs (Neurons.py:1847-1888) = the egocentric BVC on a radial manifold."""
    g = _rates()
    got = orc.bvc(g["pos"][:32], g["maze_walls"], g[f"fov_{tag}_tuning_distances"], g[f"fov_{tag}_tuning_angles"],
                  g[f"fov_{tag}_sigma_distances"], g[f"fov_{tag}_sigma_angles"], head_direction=g["hd"][:32])
    np.testing.assert_allclose(got, g[f"fov_{tag}_rates"], rtol=1e-10, atol=1e-14)


def _replay_forced(g, prefix, positions, dt, state0):
    st = state0
    env = orc.EnvSpec()
    for t in range(len(positions)):
        st = orc.agent_step(env, st, dt, None, None, forced_pos=positions[t][None])
        np.testing.assert_allclose(st["measured_velocity"][0], g[f"{prefix}_vel"][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(st["head_direction"][0], g[f"{prefix}_head_direction"][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(st["measured_rotational_velocity"][0], g[f"{prefix}_rot_vel"][t], rtol=1e-8, atol=1e-7)
        np.testing.assert_allclose(st["distance_travelled"][0], g[f"{prefix}_distance_travelled"][t], rtol=1e-10)
    return st


def test_forced_and_imported_trajectories():
    """forced_next_position and import_trajectory playback (Agent.py:229-266, 543-659)."""
    from scipy.interpolate import interp1d
    g = gu.load("imported.npz")
    st0 = gu.state_from_rows(g["forced_state0"])
    st = _replay_forced(g, "forced", g["forced_pos"], 0.02, st0)
    np.testing.assert_allclose(st["velocity"][0], g["forced_final_velocity"], rtol=1e-9)  # overwritten by measured
    # imported: the reference interpolates with a cubic spline at t % max(t) and starts at pos_interp(0)
    f = interp1d(g["imp_times"], g["imp_positions"], axis=0, kind="cubic", fill_value="extrapolate")
    ts = 0.05 * np.arange(1, 301)
    pos = f(ts % g["imp_times"].max())
    np.testing.assert_allclose(pos, g["imp_pos"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("act", sorted(gu.FF_ACTS))
def test_feedforward_layer(act):
    """FeedForwardLayer over PlaceCells + GridCells (Neurons.py:2797-2847, utils.py:919-1026)."""
    g = gu.load("feedforward.npz")
    env = orc.EnvSpec()
    pc = orc.place_cells(env, g["pos"], g["pc_centres"], 0.2)
    gc = orc.grid_cells(g["pos"], g["gc_gridscales"], g["gc_phase"], orc.grid_cell_w(g["gc_orient"]))
    f, _ = orc.feedforward([pc, gc], [g["w_pc"], g["w_gc"]], g["bias"], gu.FF_ACTS[act])
    np.testing.assert_allclose(f, g[f"ff_{act}_rates"], rtol=1e-11, atol=1e-13)
    p1 = g["agent_pos"][None]
    pc1 = orc.place_cells(env, p1, g["pc_centres"], 0.2)
    gc1 = orc.grid_cells(p1, g["gc_gridscales"], g["gc_phase"], orc.grid_cell_w(g["gc_orient"]))
    f, d = orc.feedforward([pc1, gc1], [g["w_pc"], g["w_gc"]], g["bias"], gu.FF_ACTS[act])
    np.testing.assert_allclose(f[:, 0], g[f"ff_{act}_last"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(d[:, 0], g[f"ff_{act}_prime"], rtol=1e-11, atol=1e-13)
    if act == "relu":
        f2, _ = orc.feedforward([orc.feedforward([pc, gc], [g["w_pc"], g["w_gc"]], g["bias"], gu.FF_ACTS["relu"])[0]],
                                [g["w2"]], np.zeros(5), gu.FF_ACTS["tanh"])
        pcs = orc.place_cells(env, g["pos"], g["pc_centres"], 0.2)
        np.testing.assert_allclose(f2, g["ff_stack_rates"], rtol=1e-11, atol=1e-13)


OVC_CASES = [("allo", dict(walls_occlude=True), False), ("allo_nowalls", dict(walls_occlude=False), False),
             ("ego", dict(walls_occlude=True, min_fr=0.2, max_fr=4.0), True), ("fov", dict(walls_occlude=True), True)]


@pytest.mark.parametrize("tag,kw,ego", OVC_CASES)
def test_object_vector_cells(tag, kw, ego):
    """ObjectVectorCells / FieldOfViewOVCs (Neurons.py:1892-2150)."""
    g = gu.load("ovc.npz")
    env = orc.EnvSpec(walls=g["walls"][4:])
    got = orc.object_vector_cells(env, g["pos"], g["objects"], g["object_types"], g[f"ovc_{tag}_tuning_distances"],
                                  g[f"ovc_{tag}_tuning_angles"], g[f"ovc_{tag}_sigma_distances"],
                                  g[f"ovc_{tag}_sigma_angles"], g[f"ovc_{tag}_tuning_types"],
                                  head_direction=g["hd"] if ego else None, **kw)
    np.testing.assert_allclose(got, g[f"ovc_{tag}_rates"], rtol=1e-10, atol=1e-14)


def _long_run_stats(speed, rot, dwall, pos):
    return dict(speed_mean=speed.mean(), speed_std=speed.std(), speed_q=np.quantile(speed, [0.1, 0.5, 0.9]),
                rot_std=rot.std(), dwall_hist=np.histogram(dwall, bins=10, range=(0, 0.5))[0] / dwall.size,
                pos_hist=np.histogram2d(pos[..., 0].ravel(), pos[..., 1].ravel(), bins=4, range=[[0, 1], [0, 1]])[0]
                / (pos.size / 2))


def assert_long_run_stats(got, g, name):
    """Stationary statistics against the reference's (tests/golden/stats.npz): the tolerance on the mean
    speed is 4 standard errors of the reference's own estimate (from its per-agent means) + 1 %."""
    se = g[f"{name}_speed_agent_means"].std() / np.sqrt(len(g[f"{name}_speed_agent_means"]))
    ref_mean = float(g[f"{name}_speed_mean"])
    assert abs(got["speed_mean"] - ref_mean) < 4 * se + 0.01 * ref_mean, (got["speed_mean"], ref_mean, se)
    np.testing.assert_allclose(got["speed_std"], g[f"{name}_speed_std"], rtol=0.06)
    np.testing.assert_allclose(got["speed_q"], g[f"{name}_speed_q"], rtol=0.08)
    np.testing.assert_allclose(got["rot_std"], g[f"{name}_rot_std"], rtol=0.04)
    np.testing.assert_allclose(got["dwall_hist"], g[f"{name}_dwall_hist"], atol=0.025)
    # (occupancy of a 4x4 grid mixes slowly: 40 reference agents x 75 s leave +-0.02 of sampling noise per cell)
    np.testing.assert_allclose(got["pos_hist"], g[f"{name}_pos_hist"], atol=0.045)


@pytest.mark.parametrize("name", ["open", "wall"])
def test_long_run_statistics_vs_reference(name):
    """G6: the oracle driven by its own NumPy normals reproduces the reference's stationary statistics
    (speed distribution, rotational-velocity spread, distance-to-wall and occupancy histograms)."""
    g = gu.load("stats.npz")
    rs = np.random.RandomState(5)
    env = orc.EnvSpec(walls=g[f"{name}_walls"])
    B, T, burn, dt = 160, 1500, 250, float(g["dt"])
    st = orc.init_state(env, B, 0.08, rs)
    speed, rot, dwall, pos = [], [], [], []
    for t in range(T):
        z = rs.standard_normal((2, B))
        st = orc.agent_step(env, st, dt, z[0], z[1])
        if t >= burn and t % 5 == 0:
            speed.append(np.linalg.norm(st["velocity"], axis=1))
            rot.append(st["rotational_velocity"].copy())
            dwall.append(st["distance_to_closest_wall"].copy())
            pos.append(st["pos"].copy())
    assert_long_run_stats(_long_run_stats(*map(np.array, (speed, rot, dwall, pos))), g, name)


def test_velocity_and_speed_cells():
    """VelocityCells / SpeedCell (Neurons.py:2534-2651) along the reference's own run."""
    g = gu.load("velocity.npz")
    oss = float(g["one_sigma_speed"])
    got = orc.velocity_cells(g["vel"], int(g["n"]), oss, float(g["spread"]), float(g["vc_min"]), float(g["vc_max"]))
    np.testing.assert_allclose(got, g["vc_rates"].T, rtol=1e-12)
    got = orc.speed_cell(g["mvel"], oss, float(g["sc_min"]), float(g["sc_max"]))
    # the reference's SpeedCell keeps firingrate at the base class's default length 10 (n is set to 1 only
    # after Neurons.__init__ sized the arrays): ten copies of the one rate
    assert (g["sc_rates"] == g["sc_rates"][:, :1]).all()
    np.testing.assert_allclose(got[0], g["sc_rates"][:, 0], rtol=1e-12)
    # away from the agent: direction from the kwarg, scale from the agent's own velocity
    got = orc.velocity_cells(g["gs_vel"], int(g["n"]), oss, float(g["spread"]), float(g["vc_min"]), float(g["vc_max"]),
                             scale_velocity=np.broadcast_to(g["gs_agent_vel"], g["gs_vel"].shape))
    np.testing.assert_allclose(got, g["gs_vc"], rtol=1e-12)
    np.testing.assert_allclose(orc.speed_cell(g["gs_vel"], oss, float(g["sc_min"]), float(g["sc_max"])), g["gs_sc"],
                               rtol=1e-12)


def test_environment_queries():
    """Environment.get_vectors/distances_between___accounting_for_environment, vectors_from_walls,
    check_wall_collisions, apply_boundary_conditions, called directly (Environment.py:657-894)."""
    g = gu.load("env_queries.npz")
    p1, p2 = g["p1"], g["p2"]
    maze = orc.EnvSpec(walls=g["maze_walls"][4:])
    np.testing.assert_allclose(orc.env_vectors_between(maze, p1, p2), g["maze_vec"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(orc.env_distances(maze, p1, p2), g["maze_euclid"], rtol=1e-14)
    np.testing.assert_allclose(orc.env_distances(maze, p1, p2, "line_of_sight"), g["maze_los"], rtol=1e-14)
    one = orc.EnvSpec(walls=g["one_walls"][4:])
    np.testing.assert_allclose(orc.env_distances(one, p1, p2, "geodesic"), g["one_geo"], rtol=1e-14)
    per = orc.EnvSpec(boundary_conditions="periodic")
    np.testing.assert_allclose(orc.env_vectors_between(per, p1, p2), g["per_vec"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(orc.env_distances(per, p1, p2), g["per_dist"], rtol=1e-14)
    np.testing.assert_allclose(orc.shortest_vectors_from_walls(g["pts"], g["maze_walls"]), g["maze_vfw"], rtol=1e-12,
                               atol=1e-15)
    steps = g["steps"]
    assert np.array_equal(orc.segments_collide(steps, g["maze_walls"]), g["maze_coll"])
    inside = orc.env_is_inside(maze, g["far"])
    assert np.array_equal(inside, g["solid_inside"])
    bc = np.where(inside[:, None], g["far"], orc.env_apply_boundary_conditions(maze, g["far"]))
    np.testing.assert_allclose(bc, g["solid_bc"], rtol=0, atol=1e-15)
    bc = np.where(inside[:, None], g["far"], orc.env_apply_boundary_conditions(per, g["far"]))
    np.testing.assert_allclose(bc, g["per_bc"], rtol=0, atol=1e-15)


RS_CASES = {"open": {}, "one": {}, "maze": {}, "per": {"boundary_conditions": "periodic"}}


@pytest.mark.parametrize("name", sorted(RS_CASES))
def test_random_spatial_neurons(name):
    g = gu.load("random_spatial.npz")
    env = orc.EnvSpec(walls=g[f"{name}_walls"], **RS_CASES[name])
    geom = str(g[f"{name}_geometry"])
    if geom == "geodesic" and len(env.walls) <= 4:
        geom = "euclidean"
    got = orc.random_spatial_neurons(env, g["pos"], g[f"{name}_X"], g[f"{name}_targets"], float(g[f"{name}_lengthscale"]),
                                     geom)
    np.testing.assert_allclose(got, g[f"{name}_rates"], rtol=1e-11)


AVC_CASES = {"allo": dict(min_fr=0.1, max_fr=3.0), "nowalls": dict(walls_occlude=False), "ego": dict(ego=True),
             "fov": dict(ego=True)}


@pytest.mark.parametrize("tag", sorted(AVC_CASES))
def test_agent_vector_cells(tag):
    g = gu.load("avc.npz")
    kw = dict(AVC_CASES[tag])
    hd = g["hd"] if kw.pop("ego", False) else None
    env = orc.EnvSpec(walls=g["walls"][4:])
    got = orc.agent_vector_cells(env, g["p1"], g["p2"], g[f"{tag}_tuning_distances"], g[f"{tag}_tuning_angles"],
                                 g[f"{tag}_sigma_distances"], g[f"{tag}_sigma_angles"], head_direction=hd, **kw)
    np.testing.assert_allclose(got, g[f"{tag}_rates"].T, rtol=1e-10, atol=1e-14)


# ----------------------------------------------------------------------------- row a6: polygons and holes
@pytest.mark.parametrize("tag", ["lroom", "holes", "both"])
def test_polygon_environment(tag):
    """Wall table order, strict inside test (random points + points exactly on edges / corners) and
    apply_boundary_conditions with the replacements the reference drew (tests/golden/polygon.npz)."""
    g = gu.load("polygon.npz")
    boundary, holes = gu.shape_from(g, tag + "_")
    env = orc.EnvSpec(walls=g[f"{tag}_user_walls"], boundary=boundary, holes=holes)
    assert np.array_equal(env.walls, g[f"{tag}_walls"])
    np.testing.assert_array_equal(env.extent, g[f"{tag}_extent"])
    pts = g[f"{tag}_points"]
    inside = orc.env_is_inside(env, pts)
    assert np.array_equal(inside, g[f"{tag}_inside"])
    assert 0.2 < inside.mean() < 0.9
    rs = g[f"{tag}_bc_resample"]
    assert np.array_equal(np.isfinite(rs[:, 0]), orc.env_needs_resample(env, pts))
    out = np.where(inside[:, None], pts, orc.env_apply_boundary_conditions(env, pts, rs))
    np.testing.assert_array_equal(out, g[f"{tag}_bc_out"])
    assert orc.env_is_inside(env, out).all()


def test_polygon_place_cells():
    g = gu.load("polygon.npz")
    boundary, holes = gu.shape_from(g, "lroom_")
    env = orc.EnvSpec(walls=g["lroom_user_walls"], boundary=boundary, holes=holes)
    for geom in ("euclidean", "line_of_sight"):
        got = orc.place_cells(env, g["pc_pos"], g[f"pc_{geom}_centres"], 0.15, description="gaussian_threshold",
                              wall_geometry=geom)
        np.testing.assert_allclose(got, g[f"pc_{geom}_rates"], rtol=1e-12, atol=1e-15)
    quad = orc.EnvSpec(walls=g["quad_user_walls"], boundary=g["quad_boundary"])
    got = orc.place_cells(quad, g["quad_pos"], g["quad_centres"], 0.2, wall_geometry="geodesic")
    np.testing.assert_allclose(got, g["quad_rates"], rtol=1e-12, atol=1e-15)


def test_cfg1_rollout():
    """BASELINE cfg 1 (1 agent, 100 PlaceCells, dt 10 ms, 6000 steps of the reference with its OU normals recorded):
    the oracle tracks the whole minute of trajectory and the firing rates along it."""
    g = gu.load("cfg1.npz")
    env = orc.EnvSpec()
    st = gu.state_from_rows(g["state0"][None])
    for t in range(6000):
        st = orc.agent_step(env, st, 0.01, g["z"][t, 0:1], g["z"][t, 1:2])
        if t % 50 == 49:
            np.testing.assert_allclose(st["pos"][0], g["pos"][t + 1], rtol=1e-10, atol=1e-12)
            fr = orc.place_cells(env, st["pos"], g["centres"], g["widths"])
            np.testing.assert_allclose(fr[:, 0], g["rates_every_50"][t // 50], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(st["distance_travelled"][0], float(g["distance_travelled"]), rtol=1e-12)

"""Generate the golden vectors in tests/golden/*.npz by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the resulting .npz
files are data (inputs + the reference's outputs) and are committed.  Nothing
from the reference's source travels.

    MPLBACKEND=Agg python tests/golden/make_golden.py

How the reference is driven
---------------------------
* `shapely` is absent from the image; `oracle/ref_shims/shapely` (a strict
  point-in-polygon stand-in) is put on sys.path so `import ratinabox` works.
* `np.random.normal` is wrapped: draws with `scale` 1e-6 / 1e-9 (the reference's
  geometric anti-degeneracy jitter, utils.py:64-69, 143-144) return zeros; every
  other draw is a real draw from a private RandomState, recorded as a standard
  normal (`value/scale`) so the same noise can be fed to the oracle / the GPU.
* `np.random.uniform` is wrapped likewise to record the spike uniforms.
"""
import os
import sys
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

warnings.filterwarnings("ignore")
np.seterr(all="ignore")

import ratinabox  # noqa: E402
from ratinabox.Environment import Environment  # noqa: E402
from ratinabox.Agent import Agent  # noqa: E402
from ratinabox.Neurons import (  # noqa: E402
    PlaceCells, GridCells, BoundaryVectorCells, HeadDirectionCells, FieldOfViewBVCs)
from ratinabox import utils as rutils  # noqa: E402

_real = np.random.RandomState(12345)


def _section(name):
    """Every section starts its private noise stream from its own seed, so any subset of sections — and the full
    run in any order — reproduces the committed files (tools/check_golden.py regenerates and compares)."""
    import zlib
    _real.seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    print(name)


_rec = {"normal": [], "uniform": []}
_orig_normal = np.random.normal
_orig_uniform = np.random.uniform
_capture = {"on": False}


def _patched_normal(loc=0.0, scale=1.0, size=None):
    if isinstance(scale, float) and scale in (1e-6, 1e-9):
        return np.zeros(size) + loc
    if not _capture["on"]:
        return _orig_normal(loc=loc, scale=scale, size=size)
    z = _real.standard_normal(size)
    _rec["normal"].append((np.shape(z), np.array(z, dtype=np.float64)))
    return loc + scale * z


def _patched_uniform(low=0.0, high=1.0, size=None):
    if not _capture["on"]:
        return _orig_uniform(low, high, size)
    u = _real.random_sample(size)
    _rec["uniform"].append(np.array(u, dtype=np.float64))
    return low + (high - low) * u


np.random.normal = _patched_normal
np.random.uniform = _patched_uniform

MAZE_WALLS = [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]],
              [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]

_bounces = {"n": 0}
_orig_bounce = rutils.wall_bounce


def _count_bounce(v, w):
    _bounces["n"] += 1
    return _orig_bounce(v, w)


rutils.wall_bounce = _count_bounce
_bc = {"n": 0}
_orig_abc = Environment.apply_boundary_conditions


_resampled = {"depth": 0, "pos": None}
_orig_sample = Environment.sample_positions


def _recording_sample(self, n=10, method="uniform_jitter"):
    """(inside apply_boundary_conditions: the position the outermost sample_positions(n=1, "random") call finally
    returns is the one the agent is put at; the recursive retries of rejected draws are its own business)"""
    if _resampled["depth"] == 0:
        return _orig_sample(self, n=n, method=method)
    _resampled["depth"] += 1
    out = _orig_sample(self, n=n, method=method)
    _resampled["depth"] -= 1
    if _resampled["depth"] == 1:
        _resampled["pos"] = np.array(out, dtype=np.float64).reshape(-1)
    return out


def _count_abc(self, pos):
    _bc["n"] += 1
    _resampled["depth"] = 1
    try:
        return _orig_abc(self, pos)
    finally:
        _resampled["depth"] = 0


Environment.apply_boundary_conditions = _count_abc
Environment.sample_positions = _recording_sample

STATE_KEYS = ["pos", "velocity", "rotational_velocity", "measured_velocity", "head_direction",
              "distance_travelled"]


def _get_state(ag):
    return [np.array(ag.pos, float), np.array(ag.velocity, float), float(ag.rotational_velocity),
            np.array(ag.measured_velocity, float), np.array(ag.head_direction, float),
            float(ag.distance_travelled)]


def f32exact(x):
    """Round to values exactly representable in fp32 (so the same vectors serve
    the float64 oracle test and the fp32 GPU test)."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


# ------------------------------------------------------------------ G2 / G3 -- #
def motion_records(name, env_params, agent_params, n_agents, n_steps, seed, drift=None, ratio=1.0,
                   update_kwargs=None, keep=1500, teleports=None):
    """`teleports` {(t, i): (x, y)}: agent i is put at (x, y) before its update of step t, like a user writing
    `Ag.pos = ...` — how agents get into holes / outside a polygonal boundary, where the step ends in the resample
    branch of apply_boundary_conditions."""
    teleports = teleports or {}
    np.random.seed(seed)
    Env = Environment(env_params)
    agents = [Agent(Env, dict(agent_params)) for _ in range(n_agents)]
    update_kwargs = update_kwargs or {}
    pre, post, zs, nb, bc, rsp = [], [], [], [], [], []
    zroll_rs = np.full((n_steps, n_agents, 2), np.nan)
    tele_roll = np.full((n_steps, n_agents, 2), np.nan)
    traj = np.zeros((n_steps + 1, n_agents, 2))
    traj[0] = [a.pos for a in agents]
    zroll = np.zeros((n_steps, n_agents, 2))
    state0 = [_get_state(a) for a in agents]
    _capture["on"] = True
    for t in range(n_steps):
        for i, ag in enumerate(agents):
            if (t, i) in teleports:
                ag.pos = np.array(teleports[(t, i)], dtype=float)
                tele_roll[t, i] = ag.pos
            s0 = _get_state(ag)
            _rec["normal"].clear()
            _bounces["n"] = 0
            _bc["n"] = 0
            _resampled["pos"] = None
            dv = None if drift is None else np.array(drift[(t + i) % len(drift)], float)
            ag.update(drift_velocity=dv, drift_to_random_strength_ratio=ratio, **update_kwargs)
            scal = [z for shp, z in _rec["normal"] if shp == ()]
            assert len(scal) == 2, len(scal)
            s1 = _get_state(ag)
            pre.append(np.concatenate([np.ravel(x) for x in s0]))
            post.append(np.concatenate([np.ravel(x) for x in s1] +
                                       [[float(ag.measured_rotational_velocity)],
                                        [float(ag.distance_to_closest_wall)]]))
            zs.append([float(scal[0]), float(scal[1])])
            nb.append(_bounces["n"])
            bc.append(_bc["n"])
            rsp.append(_resampled["pos"] if _resampled["pos"] is not None else np.full(2, np.nan))
            zroll_rs[t, i] = rsp[-1]
            zroll[t, i] = zs[-1]
            traj[t + 1, i] = ag.pos
    _capture["on"] = False
    pre, post, zs, rsp = np.array(pre), np.array(post), np.array(zs), np.array(rsp)
    nb, bc = np.array(nb), np.array(bc)
    # keep every "interesting" record plus a random subsample of the rest
    interesting = np.nonzero((nb > 0) | (bc > 0))[0]
    rest = np.setdiff1d(np.arange(len(pre)), interesting)
    rs = np.random.RandomState(seed)
    take = rs.choice(rest, size=min(len(rest), max(0, keep - len(interesting))), replace=False)
    sel = np.sort(np.concatenate([interesting[:keep], take]))
    drift_arr = np.zeros((0, 2))
    if drift is not None:
        idx = np.array([(t + i) % len(drift) for t in range(n_steps) for i in range(n_agents)])
        drift_arr = np.array(drift, float)[idx][sel]
    out = dict(
        env_scale=float(Env.scale), env_aspect=float(Env.aspect),
        env_bc=str(Env.boundary_conditions), user_walls=np.array(env_params.get("walls", []), float).reshape(-1, 2, 2),
        ref_walls=np.array(Env.walls, float).reshape(-1, 2, 2),
        dt=float(agents[0].dt),
        params_keys=np.array(sorted(agent_params.keys())),
        params_vals=np.array([float(agent_params[k]) for k in sorted(agent_params.keys())]),
        kw_keys=np.array(sorted(update_kwargs.keys())),
        kw_vals=np.array([float(update_kwargs[k]) for k in sorted(update_kwargs.keys())]),
        pre=pre[sel], post=post[sel], z=zs[sel], n_bounces=nb[sel], bc_applied=bc[sel],
        # polygonal boundary / holes (empty: the rectangular box); resample positions (NaN: none drawn)
        env_boundary=np.array(env_params.get("boundary") if env_params.get("boundary") is not None else np.zeros((0, 2)),
                              dtype=float).reshape(-1, 2),
        env_holes=np.array([c for h in env_params.get("holes", []) for c in h], dtype=float).reshape(-1, 2),
        env_hole_sizes=np.array([len(h) for h in env_params.get("holes", [])], dtype=np.int64),
        resample=rsp[sel], roll_resample=zroll_rs, roll_teleport=tele_roll,
        drift=drift_arr, drift_ratio=float(ratio),
        # rollout (G3): first 16 agents, all steps
        roll_state0=np.array([np.concatenate([np.ravel(x) for x in s]) for s in state0]),
        roll_z=zroll, roll_pos=traj,
        roll_final=np.array([np.concatenate([np.ravel(x) for x in _get_state(a)]) for a in agents]),
    )
    print(f"  {name}: {len(sel)} step records ({int((nb[sel] > 0).sum())} with bounces, "
          f"{int((nb[sel] > 1).sum())} multi-bounce, {int((bc[sel] > 0).sum())} boundary conditions, "
          f"{int(np.isfinite(rsp[sel][:, 0]).sum())} resampled)")
    np.savez_compressed(os.path.join(HERE, f"motion_{name}.npz"), **out)


L_ROOM = [[0, 0], [1, 0], [1, 0.5], [0.5, 0.5], [0.5, 1], [0, 1]]
HOLE_A = [[0.35, 0.35], [0.65, 0.35], [0.65, 0.65], [0.35, 0.65]]
HOLE_B = [[0.1, 0.7], [0.25, 0.7], [0.22, 0.9], [0.12, 0.88]]  # (the reference needs equal corner counts: np.array(holes).ndim == 3)


def make_motion():
    _section("motion (G2 single steps + G3 rollouts)")
    motion_records("open_dt10ms", {}, {"dt": 0.01}, 16, 400, seed=1)
    motion_records("maze_dt10ms", {"walls": MAZE_WALLS}, {"dt": 0.01}, 16, 400, seed=2)
    motion_records("maze_dt50ms", {"walls": MAZE_WALLS}, {"dt": 0.05, "thigmotaxis": 0.2}, 16, 300, seed=3)
    motion_records("maze_fast", {"walls": MAZE_WALLS},
                   {"dt": 0.1, "speed_mean": 0.6, "thigmotaxis": 0.9, "wall_repel_strength": 0.3}, 16, 250, seed=4)
    motion_records("open_norepel_fast", {}, {"dt": 0.1, "speed_mean": 0.5, "wall_repel_strength": 0.0}, 8, 300, seed=5)
    motion_records("open_drift", {"scale": 2.0, "aspect": 1.5}, {"dt": 0.02, "speed_std": 0.0}, 8, 200, seed=6,
                   drift=[[0.3, 0.1], [-0.2, 0.25], [0.0, -0.4]], ratio=3.0)
    motion_records("periodic_wall", {"boundary_conditions": "periodic", "walls": [[[.5, .2], [.5, .8]]]},
                   {"dt": 0.05, "speed_mean": 0.3}, 8, 300, seed=7)
    motion_records("open_kwargs", {}, {"dt": 0.01, "head_direction_smoothing_timescale": 0.005}, 8, 150, seed=8,
                   update_kwargs={"thigmotaxis": 0.8, "wall_repel_distance": 0.2, "speed_mean": 0.12,
                                  "rotational_velocity_std": 1.0, "speed_coherence_time": 0.3,
                                  "rotational_velocity_coherence_time": 0.2, "wall_repel_strength": 1.5})
    # row a6: polygonal boundary and holes (Environment.py:128-163, 781-818, 855-894).  Some agents are put into a
    # hole / outside the polygon now and then: those steps end in the resample branch.
    rs = np.random.RandomState(99)
    tele = {}
    for t in range(10, 300, 17):
        tele[(t, int(rs.randint(16)))] = (0.5 + 0.4 * rs.rand(), 0.5 + 0.4 * rs.rand())      # the notch of the L
    for t in range(5, 300, 23):
        tele[(t, int(rs.randint(16)))] = (0.05 + 0.4 * rs.rand(), 0.05 + 0.9 * rs.rand())    # a legal jump
    motion_records("lroom_dt20ms", {"boundary": L_ROOM, "walls": [[[0.25, 0.0], [0.25, 0.3]]]},
                   {"dt": 0.02, "speed_mean": 0.15}, 16, 300, seed=9, teleports=tele)
    tele = {}
    for t in range(8, 300, 13):
        hole = HOLE_A if (t // 13) % 2 == 0 else HOLE_B
        c = np.mean(hole, axis=0)
        tele[(t, int(rs.randint(16)))] = tuple(c + 0.02 * (rs.rand(2) - 0.5))                # inside a hole
    for t in range(3, 300, 29):
        tele[(t, int(rs.randint(16)))] = (1.0 + 0.01 * rs.rand(), 0.3 + 0.4 * rs.rand())      # outside the box: clamp
    motion_records("box_holes_dt20ms", {"holes": [HOLE_A, HOLE_B]}, {"dt": 0.02, "speed_mean": 0.15, "thigmotaxis": 0.3},
                   16, 300, seed=10, teleports=tele)


# ----------------------------------------------------------------------- G1 -- #
def test_positions(P, scale=1.0, aspect=1.0, seed=0):
    rs = np.random.RandomState(seed)
    pos = np.stack((rs.uniform(0, aspect * scale, P), rs.uniform(0, scale, P)), -1)
    k = P // 8
    # near the boundary and near the maze walls / their endpoints
    pos[:k, 0] = rs.choice([1e-3, aspect * scale - 1e-3, 0.2 - 1e-3, 0.2 + 1e-3, 0.4 + 2e-3, 0.8 - 2e-3], k)
    pos[k:2 * k, 1] = rs.choice([1e-3, scale - 1e-3, 0.5 + 1e-3, 0.5 - 1e-3, 0.4 + 1e-3, 0.6 - 1e-3], k)
    return f32exact(pos)


def make_rates():
    _section("rates (G1)")
    np.random.seed(11)
    out = {}
    pos = test_positions(192, seed=3)
    hd = np.random.RandomState(5).randn(192, 2)
    hd = f32exact(hd / np.linalg.norm(hd, axis=1, keepdims=True))
    out["pos"] = pos
    out["hd"] = hd

    # --- PlaceCells: all five descriptions, open box
    Env = Environment()
    Ag = Agent(Env)
    for desc in ["gaussian", "gaussian_threshold", "diff_of_gaussians", "one_hot", "top_hat"]:
        PCs = PlaceCells(Ag, {"n": 64, "description": desc, "widths": 0.2, "min_fr": 0.1, "max_fr": 2.0})
        PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
        PCs.place_cell_widths = f32exact(np.linspace(0.1, 0.3, 64))
        out[f"pc_{desc}_centres"] = PCs.place_cell_centres
        out[f"pc_{desc}_widths"] = PCs.place_cell_widths
        out[f"pc_{desc}_rates"] = PCs.get_state(evaluate_at=None, pos=pos)
    # cfg-2 shape table (1024 cells default widths), fewer positions
    PCs = PlaceCells(Ag, {"n": 1024})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    out["pc_big_centres"] = PCs.place_cell_centres
    out["pc_big_rates"] = PCs.get_state(evaluate_at=None, pos=pos[:64])
    # --- PlaceCells: line_of_sight in the 9-wall maze; geodesic with one wall; periodic
    EnvM = Environment({"walls": MAZE_WALLS})
    AgM = Agent(EnvM)
    PCs = PlaceCells(AgM, {"n": 64, "wall_geometry": "line_of_sight", "widths": 0.25})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    out["pc_los_centres"] = PCs.place_cell_centres
    out["pc_los_rates"] = PCs.get_state(evaluate_at=None, pos=pos)
    out["maze_walls"] = np.array(EnvM.walls, float)
    Env1 = Environment({"walls": [[[0.5, 0.0], [0.5, 0.6]]]})
    Ag1 = Agent(Env1)
    PCs = PlaceCells(Ag1, {"n": 49, "wall_geometry": "geodesic", "description": "gaussian_threshold"})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    out["pc_geo_centres"] = PCs.place_cell_centres
    out["pc_geo_rates"] = PCs.get_state(evaluate_at=None, pos=pos)
    out["geo_walls"] = np.array(Env1.walls, float)
    EnvP = Environment({"boundary_conditions": "periodic"})
    AgP = Agent(EnvP)
    PCs = PlaceCells(AgP, {"n": 36, "widths": 0.15})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    out["pc_per_centres"] = PCs.place_cell_centres
    out["pc_per_rates"] = PCs.get_state(evaluate_at=None, pos=pos)

    # --- GridCells
    for desc in ["rectified_cosines", "shifted_cosines"]:
        GCs = GridCells(Ag, {"n": 60, "description": desc, "min_fr": 0.0, "max_fr": 1.5})
        out[f"gc_{desc}_gridscales"] = np.array(GCs.gridscales, float)
        out[f"gc_{desc}_phase_offsets"] = np.array(GCs.phase_offsets, float)
        out[f"gc_{desc}_orientations"] = np.array(GCs.orientations, float)
        out[f"gc_{desc}_w"] = np.array(GCs.w, float)
        out[f"gc_{desc}_rates"] = GCs.get_state(evaluate_at=None, pos=pos)
    GCs = GridCells(Ag, {"n": 40, "gridscale_distribution": "uniform", "gridscale": (0.2, 1.0),
                         "orientation_distribution": "uniform", "orientation": (0, 2 * np.pi), "width_ratio": 0.5})
    out["gc_rand_gridscales"] = np.array(GCs.gridscales, float)
    out["gc_rand_phase_offsets"] = np.array(GCs.phase_offsets, float)
    out["gc_rand_orientations"] = np.array(GCs.orientations, float)
    out["gc_rand_w"] = np.array(GCs.w, float)
    out["gc_rand_rates"] = GCs.get_state(evaluate_at=None, pos=pos)

    # --- BVCs: allocentric open box / maze; egocentric maze (per-position head direction)
    for tag, ag in [("open", Ag), ("maze", AgM)]:
        B = BoundaryVectorCells(ag, {"n": 32, "min_fr": 0.0, "max_fr": 1.0})
        for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles", "test_angles",
                  "test_directions", "cell_fr_norm"]:
            out[f"bvc_{tag}_{k}"] = np.array(getattr(B, k), float)
        out[f"bvc_{tag}_rates"] = B.get_state(evaluate_at=None, pos=pos)
    Be = BoundaryVectorCells(AgM, {"n": 24, "reference_frame": "egocentric", "max_fr": 3.0, "min_fr": 0.5})
    for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles"]:
        out[f"bvc_ego_{k}"] = np.array(getattr(Be, k), float)
    # the reference's egocentric path takes ONE head direction per call; evaluate position by position
    ego = np.zeros((Be.n, 48))
    for j in range(48):
        ego[:, j] = Be.get_state(evaluate_at=None, pos=pos[j:j + 1], head_direction=hd[j])[:, 0]
    out["bvc_ego_rates"] = ego

    # --- HeadDirectionCells
    H = HeadDirectionCells(Ag, {"n": 24, "angular_spread_degrees": 30, "max_fr": 2.0, "min_fr": 0.25})
    hdr = np.zeros((24, 192))
    for j in range(192):
        hdr[:, j] = H.get_state(evaluate_at=None, head_direction=hd[j], pos=pos[j:j + 1])[:, 0]
    out["hdc_rates"] = hdr
    # --- FieldOfViewBVCs (egocentric radial manifolds; no random draws)
    for tag, prm in (("div", {}), ("uni", {"cell_arrangement": "uniform_manifold", "distance_range": [0.05, 0.3],
                                           "angle_range": [0, 120], "spatial_resolution": 0.05})):
        F = FieldOfViewBVCs(AgM, dict(prm))
        for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles"]:
            out[f"fov_{tag}_{k}"] = np.array(getattr(F, k), float)
        fr = np.zeros((F.n, 32))
        for j in range(32):
            fr[:, j] = F.get_state(evaluate_at=None, pos=pos[j:j + 1], head_direction=hd[j])[:, 0]
        out[f"fov_{tag}_rates"] = fr
    np.savez_compressed(os.path.join(HERE, "rates.npz"), **out)
    print("  rates.npz written:", len(out), "arrays")


# ------------------------------------------------------------------ G4 / G5 -- #
def make_update_and_init():
    """Neurons.update() end-to-end (rates + noise OU + spikes) for a single agent,
    and seeded init tables (host-side parameter sampling)."""
    _section("update/spikes (G4) + init tables (G5)")
    out = {}
    np.random.seed(21)
    Env = Environment()
    Ag = Agent(Env, {"dt": 0.01})
    PCs = PlaceCells(Ag, {"n": 50, "max_fr": 40.0, "noise_std": 0.5, "noise_coherence_time": 0.2})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    out["upd_centres"] = PCs.place_cell_centres
    T = 120
    pos_l, fr_l, spk_l, zn_l, u_l, noise_l = [], [], [], [], [], []
    _capture["on"] = True
    for t in range(T):
        _rec["normal"].clear()
        Ag.update()
        Ag.pos = f32exact(Ag.pos)
        _rec["normal"].clear()
        _rec["uniform"].clear()
        PCs.update()
        zn = [z for shp, z in _rec["normal"] if shp == (50,)]
        assert len(zn) == 1 and len(_rec["uniform"]) == 1
        pos_l.append(Ag.pos.copy())
        fr_l.append(PCs.firingrate.copy())
        noise_l.append(PCs.noise.copy())
        spk_l.append(np.array(PCs.history["spikes"][-1]))
        zn_l.append(zn[0])
        u_l.append(_rec["uniform"][0])
    _capture["on"] = False
    out.update(upd_dt=0.01, upd_pos=np.array(pos_l), upd_fr=np.array(fr_l), upd_noise=np.array(noise_l),
               upd_spikes=np.array(spk_l), upd_z=np.array(zn_l), upd_u=np.array(u_l))
    margin = np.abs(np.array(u_l) - 0.01 * np.array(fr_l)) / np.maximum(0.01 * np.abs(np.array(fr_l)), 1e-30)
    out["upd_min_rel_margin"] = float(margin.min())
    print(f"  spikes: {int(np.array(spk_l).sum())} spikes in {np.array(spk_l).size}, "
          f"min relative margin {margin.min():.3e}")

    # G5: seeded init tables
    for seed in (0, 7):
        np.random.seed(seed)
        Env = Environment()
        Ag = Agent(Env)
        out[f"init{seed}_agent_pos"] = np.array(Ag.pos)
        out[f"init{seed}_agent_vel"] = np.array(Ag.velocity)
        P = PlaceCells(Ag, {"n": 100})
        out[f"init{seed}_pc_centres"] = np.array(P.place_cell_centres)
        P2 = PlaceCells(Ag, {"n": 37, "place_cell_centres": "random"})
        out[f"init{seed}_pc_random_centres"] = np.array(P2.place_cell_centres)
        G = GridCells(Ag, {"n": 32})
        out[f"init{seed}_gc_gridscales"] = np.array(G.gridscales)
        out[f"init{seed}_gc_phase"] = np.array(G.phase_offsets)
        out[f"init{seed}_gc_orient"] = np.array(G.orientations)
        Bv = BoundaryVectorCells(Ag, {"n": 20})
        for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles"]:
            out[f"init{seed}_bvc_{k}"] = np.array(getattr(Bv, k))
        EnvW = Environment({"walls": MAZE_WALLS, "scale": 1.0})
        out[f"init{seed}_maze_walls"] = np.array(EnvW.walls)
        out[f"init{seed}_sample_uniform16"] = EnvW.sample_positions(16, "uniform")
    np.savez_compressed(os.path.join(HERE, "update_init.npz"), **out)


def make_imported():
    """Imported (cubic-spline playback) and forced trajectories: Agent.py:229-266, 543-659."""
    _section("imported / forced trajectories")
    out = {}
    times = np.arange(0, 12.01, 0.4)
    positions = np.stack((0.5 + 0.35 * np.sin(0.9 * times), 0.5 + 0.3 * np.cos(1.3 * times + 0.4)), axis=-1)
    out["imp_times"], out["imp_positions"] = times, positions

    def run(ag, n, **kw):
        for _ in range(n):
            ag.update(**kw)
        h = ag.get_history_arrays()
        return {k: np.array(h[k], float) for k in ("t", "pos", "vel", "rot_vel", "head_direction", "distance_travelled")}

    np.random.seed(31)
    Env = Environment()
    Ag = Agent(Env, {"dt": 0.05})
    Ag.import_trajectory(times=times, positions=positions)
    out["imp_state0"] = np.concatenate([np.ravel(x) for x in _get_state(Ag)])
    for k, v in run(Ag, 300).items():          # 15 s > 12 s of data: exercises the loop-around
        out[f"imp_{k}"] = v
    out["imp_final_velocity"] = np.array(Ag.velocity, float)
    out["imp_final_rotvel"] = float(Ag.rotational_velocity)
    # (interpolate=False raises AttributeError in the reference v1.15.3 — Agent.py:657 calls
    #  self.pos_interp unconditionally — so there is nothing to record for that mode)
    # forced_next_position
    Ag3 = Agent(Env, {"dt": 0.02})
    out["forced_state0"] = np.concatenate([np.ravel(x) for x in _get_state(Ag3)])
    rs = np.random.RandomState(4)
    p = np.array(Ag3.pos, float)
    forced = []
    for _ in range(60):
        p = np.clip(p + 0.004 * rs.randn(2), 0.05, 0.95)
        forced.append(p.copy())
        Ag3.update(forced_next_position=p.copy())
    out["forced_pos"] = np.array(forced)
    h = Ag3.get_history_arrays()
    for k in ("t", "pos", "vel", "rot_vel", "head_direction", "distance_travelled"):
        out[f"forced_{k}"] = np.array(h[k], float)
    out["forced_final_velocity"] = np.array(Ag3.velocity, float)
    np.savez_compressed(os.path.join(HERE, "imported.npz"), **out)


def make_feedforward():
    """FeedForwardLayer (Neurons.py:2654-2860) over PlaceCells + GridCells, every named activation."""
    from ratinabox.Neurons import FeedForwardLayer
    _section("feedforward")
    out = {}
    np.random.seed(41)
    Env = Environment()
    Ag = Agent(Env, {"dt": 0.05})
    PCs = PlaceCells(Ag, {"n": 70, "name": "PCs"})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    GCs = GridCells(Ag, {"n": 45, "name": "GCs"})
    out["pc_centres"] = PCs.place_cell_centres
    out["gc_gridscales"], out["gc_phase"], out["gc_orient"] = GCs.gridscales, GCs.phase_offsets, GCs.orientations
    pos = test_positions(96, seed=9)
    out["pos"] = pos
    w_pc = np.random.RandomState(1).randn(37, 70) / np.sqrt(70)
    w_gc = np.random.RandomState(2).randn(37, 45) / np.sqrt(45)
    bias = np.random.RandomState(3).randn(37) * 0.1
    out["w_pc"], out["w_gc"], out["bias"] = w_pc, w_gc, bias
    acts = {"linear": {"activation": "linear"},
            "sigmoid": {"activation": "sigmoid", "max_fr": 5, "min_fr": 0.5, "mid_x": 0.2, "width_x": 1.5},
            "relu": {"activation": "relu", "gain": 2.0, "threshold": 0.1},
            "tanh": {"activation": "tanh", "gain": 1.5, "threshold": -0.2},
            "retanh": {"activation": "retanh", "gain": 1.2, "threshold": 0.05},
            "softmax": {"activation": "softmax", "gain": 0.7, "threshold": 0.3}}
    Ag.pos = f32exact(np.array([0.37, 0.61]))
    PCs.update()
    GCs.update()
    out["agent_pos"] = np.array(Ag.pos)
    for name, spec in acts.items():
        F = FeedForwardLayer(Ag, {"n": 37, "input_layers": [PCs, GCs], "activation_function": dict(spec),
                                  "biases": bias.copy(), "name": "FF_" + name})
        F.inputs["PCs"]["w"] = w_pc.copy()
        F.inputs["GCs"]["w"] = w_gc.copy()
        out[f"ff_{name}_rates"] = F.get_state(evaluate_at=None, pos=pos)
        F.update()
        out[f"ff_{name}_last"] = np.array(F.firingrate)
        out[f"ff_{name}_prime"] = np.array(F.firingrate_prime)
    # two-layer stack: FF2(FF1(PCs, GCs))
    F1 = FeedForwardLayer(Ag, {"n": 37, "input_layers": [PCs, GCs], "activation_function": acts["relu"],
                               "biases": bias.copy(), "name": "F1"})
    F1.inputs["PCs"]["w"], F1.inputs["GCs"]["w"] = w_pc.copy(), w_gc.copy()
    w2 = np.random.RandomState(4).randn(5, 37) / np.sqrt(37)
    F2 = FeedForwardLayer(Ag, {"n": 5, "input_layers": [F1], "activation_function": acts["tanh"], "name": "F2"})
    F2.inputs["F1"]["w"] = w2.copy()
    out["w2"] = w2
    out["ff_stack_rates"] = F2.get_state(evaluate_at=None, pos=pos)
    np.savez_compressed(os.path.join(HERE, "feedforward.npz"), **out)


def make_ovc():
    """ObjectVectorCells / FieldOfViewOVCs (Neurons.py:1892-2150) with occluding walls."""
    from ratinabox.Neurons import ObjectVectorCells, FieldOfViewOVCs
    _section("object vector cells")
    out = {}
    np.random.seed(51)
    Env = Environment({"walls": [[[0.5, 0.0], [0.5, 0.55]], [[0.2, 0.8], [0.6, 0.8]]]})
    rs = np.random.RandomState(6)
    objs = f32exact(rs.uniform(0.05, 0.95, (9, 2)))
    types = [0, 0, 1, 2, 1, 0, 2, 2, 1]
    for o, ty in zip(objs, types):
        Env.add_object(o, type=ty)
    out["objects"], out["object_types"] = np.array(Env.objects["objects"]), np.array(Env.objects["object_types"])
    out["walls"] = np.array(Env.walls, float)
    Ag = Agent(Env)
    pos = test_positions(64, seed=13)
    hd = np.random.RandomState(14).randn(64, 2)
    hd = f32exact(hd / np.linalg.norm(hd, axis=1, keepdims=True))
    out["pos"], out["hd"] = pos, hd
    for tag, cls, prm in (("allo", ObjectVectorCells, {"n": 30}),
                          ("allo_nowalls", ObjectVectorCells, {"n": 12, "walls_occlude": False, "object_tuning_type": 1}),
                          ("ego", ObjectVectorCells, {"n": 20, "reference_frame": "egocentric", "max_fr": 4.0, "min_fr": 0.2}),
                          ("fov", FieldOfViewOVCs, {"object_tuning_type": "random", "angle_range": [0, 100]})):
        O = cls(Ag, dict(prm))
        for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles", "tuning_types"]:
            out[f"ovc_{tag}_{k}"] = np.array(getattr(O, k), float)
        if O.reference_frame == "egocentric":
            fr = np.zeros((O.n, 64))
            for j in range(64):
                fr[:, j] = O.get_state(evaluate_at=None, pos=pos[j:j + 1], head_direction=hd[j])[:, 0]
        else:
            fr = O.get_state(evaluate_at=None, pos=pos)
        out[f"ovc_{tag}_rates"] = fr
    np.savez_compressed(os.path.join(HERE, "ovc.npz"), **out)


def make_avc():
    """AgentVectorCells / FieldOfViewAVCs (Neurons.py:2151-2355): two agents in a walled box updated in
    turn; per step both positions, the observer's head direction and the rates of its cells."""
    from ratinabox.Neurons import AgentVectorCells, FieldOfViewAVCs
    _section("agent vector cells")
    out = {}
    np.random.seed(61)
    Env = Environment({"walls": [[[0.5, 0.0], [0.5, 0.55]], [[0.2, 0.8], [0.6, 0.8]]]})
    out["walls"] = np.array(Env.walls, float)
    Ag1 = Agent(Env, {"dt": 0.05, "speed_mean": 0.2})
    Ag2 = Agent(Env, {"dt": 0.05, "speed_mean": 0.2})
    pops = {"allo": AgentVectorCells(Ag1, Ag2, {"n": 12, "max_fr": 3.0, "min_fr": 0.1}),
            "nowalls": AgentVectorCells(Ag1, Ag2, {"n": 8, "walls_occlude": False}),
            "ego": AgentVectorCells(Ag1, Ag2, {"n": 10, "reference_frame": "egocentric"}),
            "fov": FieldOfViewAVCs(Ag1, Ag2, {"angle_range": [0, 120], "distance_range": [0.05, 0.6],
                                              "spatial_resolution": 0.08})}
    for tag, N in pops.items():
        for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles"]:
            out[f"{tag}_{k}"] = np.array(getattr(N, k), float)
    p1, p2, hd, rates = [], [], [], {t: [] for t in pops}
    for t in range(300):
        Ag1.update()
        Ag2.update()
        Ag1.pos, Ag2.pos = f32exact(Ag1.pos), f32exact(Ag2.pos)
        Ag1.head_direction = f32exact(Ag1.head_direction)
        for tag, N in pops.items():
            N.update()
            rates[tag].append(N.firingrate.copy())
        p1.append(Ag1.pos.copy()); p2.append(Ag2.pos.copy()); hd.append(Ag1.head_direction.copy())
    out["p1"], out["p2"], out["hd"] = np.array(p1), np.array(p2), np.array(hd)
    for tag in pops:
        out[f"{tag}_rates"] = np.array(rates[tag])
    occl = (out["allo_rates"] <= 0.1 + 1e-12).all(axis=1).sum()
    print(f"  other agent out of sight on {int(occl)} of 300 steps; fov cells: {pops['fov'].n}")
    np.savez_compressed(os.path.join(HERE, "avc.npz"), **out)


def make_velocity():
    """VelocityCells / SpeedCell (Neurons.py:2534-2651) along the reference's own run in the maze: per step
    the velocity state (what VelocityCells read), the measured velocity (history["vel"][-1], what the
    SpeedCell reads) and both populations' rates after update(); plus get_state on given velocities."""
    from ratinabox.Neurons import VelocityCells, SpeedCell
    _section("velocity / speed cells")
    np.random.seed(41)
    out = {}
    Env = Environment({"walls": MAZE_WALLS})
    Ag = Agent(Env, {"dt": 0.02, "speed_mean": 0.15})
    VCs = VelocityCells(Ag, {"n": 12, "angular_spread_degrees": 30, "min_fr": 0.2, "max_fr": 3.0})
    SC = SpeedCell(Ag, {"min_fr": 0.5, "max_fr": 2.0})
    vel, mvel, vr, sr = [], [], [], []
    for t in range(400):
        Ag.update()
        VCs.update()
        SC.update()
        vel.append(np.array(Ag.velocity))
        mvel.append(np.array(Ag.history["vel"][-1]))
        vr.append(VCs.firingrate.copy())
        sr.append(SC.firingrate.copy())
    out.update(vel=np.array(vel), mvel=np.array(mvel), vc_rates=np.array(vr), sc_rates=np.array(sr),
               one_sigma_speed=VCs.one_sigma_speed, n=12, spread=30.0, vc_min=0.2, vc_max=3.0, sc_min=0.5, sc_max=2.0)
    differs = np.abs(np.array(vel) - np.array(mvel)).max(axis=1) > 1e-9
    print(f"  velocity != measured velocity on {int(differs.sum())} of 400 steps")
    # get_state away from the agent: the scale still uses the AGENT's speed (Neurons.py:2581)
    rs = np.random.RandomState(3)
    v = f32exact(rs.normal(0, 0.2, (64, 2)))
    out["gs_vel"] = v
    out["gs_agent_vel"] = np.array(Ag.velocity)
    out["gs_vc"] = np.stack([VCs.get_state(evaluate_at=None, velocity=v[i])[:, 0] for i in range(64)], axis=1)
    out["gs_sc"] = np.stack([SC.get_state(evaluate_at=None, vel=v[i]) for i in range(64)], axis=1)
    np.savez_compressed(os.path.join(HERE, "velocity.npz"), **out)


def make_env_queries():
    """Environment geometry queries called directly (Environment.py:657-894): pairwise vectors / distances
    under each wall geometry, vectors_from_walls, check_wall_collisions, apply_boundary_conditions."""
    _section("environment queries")
    rs = np.random.RandomState(17)
    out = {}
    p1, p2 = f32exact(rs.uniform(0, 1, (40, 2))), f32exact(rs.uniform(0, 1, (50, 2)))
    out["p1"], out["p2"] = p1, p2
    maze = Environment({"walls": MAZE_WALLS})
    out["maze_walls"] = np.array(maze.walls)
    out["maze_vec"] = maze.get_vectors_between___accounting_for_environment(p1, p2)
    out["maze_euclid"] = maze.get_distances_between___accounting_for_environment(p1, p2, wall_geometry="euclidean")
    out["maze_los"] = maze.get_distances_between___accounting_for_environment(p1, p2, wall_geometry="line_of_sight")
    one = Environment({"walls": [[[0.5, 0.0], [0.5, 0.6]]]})
    out["one_walls"] = np.array(one.walls)
    out["one_geo"] = one.get_distances_between___accounting_for_environment(p1, p2, wall_geometry="geodesic")
    per = Environment({"boundary_conditions": "periodic"})
    d, v = per.get_distances_between___accounting_for_environment(p1, p2, return_vectors=True)
    out["per_dist"], out["per_vec"] = d, v
    # single-position / single-step queries, as Agent.update makes them
    pts = f32exact(np.concatenate((rs.uniform(0, 1, (60, 2)), rs.uniform(-0.3, 1.3, (20, 2)))))
    out["pts"] = pts
    out["maze_vfw"] = np.array([maze.vectors_from_walls(p.copy()) for p in pts])
    steps = f32exact(np.stack((rs.uniform(0, 1, (300, 2)), rs.uniform(0, 1, (300, 2))), axis=1))
    steps[:, 1] = f32exact(steps[:, 0] + 0.3 * (steps[:, 1] - 0.5))
    out["steps"] = steps
    out["maze_coll"] = np.array([maze.check_wall_collisions(s.copy())[1] for s in steps])
    print(f"  {int(out['maze_coll'].sum())} wall crossings in 300 steps, "
          f"{int((out['maze_los'] == 1000).sum())} of {out['maze_los'].size} pairs without line of sight")
    far = f32exact(rs.uniform(-0.5, 1.5, (80, 2)))
    out["far"] = far
    out["solid_bc"] = np.array([maze.apply_boundary_conditions(p.copy()) for p in far])
    out["per_bc"] = np.array([per.apply_boundary_conditions(p.copy()) for p in far])
    out["solid_inside"] = np.array([maze.check_if_position_is_in_environment(p) for p in far])
    np.savez_compressed(os.path.join(HERE, "env_queries.npz"), **out)


def make_random_spatial():
    """RandomSpatialNeurons (Neurons.py:2865-2960): seeded targets (one multivariate-normal draw over the
    anchor grid) and get_state at given positions, for each wall geometry."""
    from ratinabox.Neurons import RandomSpatialNeurons
    _section("random spatial neurons")
    out = {}
    rs = np.random.RandomState(8)
    pos = f32exact(rs.uniform(0, 1, (150, 2)))
    out["pos"] = pos
    cases = {"open": ({}, "euclidean", 0.1), "one": ({"walls": [[[0.5, 0.0], [0.5, 0.6]]]}, "geodesic", 0.12),
             "maze": ({"walls": MAZE_WALLS}, "geodesic", 0.08), "per": ({"boundary_conditions": "periodic"}, "euclidean", 0.15)}
    for name, (envp, geom, ell) in cases.items():
        np.random.seed(33)
        Env = Environment(envp)
        Ag = Agent(Env)
        N = RandomSpatialNeurons(Ag, {"n": 7, "lengthscale": ell, "wall_geometry": geom, "min_fr": 0.5, "max_fr": 4.0})
        out[f"{name}_geometry"] = N.wall_geometry
        out[f"{name}_lengthscale"] = ell
        out[f"{name}_X"] = np.array(N.X)
        out[f"{name}_targets"] = np.array(N.targets)
        out[f"{name}_rates"] = N.get_state(evaluate_at=None, pos=pos)
        out[f"{name}_walls"] = np.array(envp.get("walls", []), float).reshape(-1, 2, 2)
        print(f"  {name}: {N.wall_geometry}, {N.X.shape[0]} anchors, nan rates {int(np.isnan(out[f'{name}_rates']).sum())}")
    np.savez_compressed(os.path.join(HERE, "random_spatial.npz"), **out)


def make_task():
    """TaskEnvironment.step / reset (contribs/TaskEnvironment.py): single-agent replicas of a
    SpatialGoalEnvironment, one per lane, driven towards their goals; per step the action, the two
    OU normals, the resulting position, reward total, terminal flag and goal / reward cache sizes."""
    from ratinabox.contribs.TaskEnvironment import (SpatialGoalEnvironment, SpatialGoal, Reward, get_goal_vector)
    _section("task environment")
    presets = {"constant": 0, "linear": 1, "exponential": 2, "none": 3}
    scenarios = {
        # name: (env params, goal positions, radius, rewards per goal (None = reward_default), goalcachekws,
        #        terminate delay, teleport, n lanes, n steps, speed factor)
        "open_nonseq": ({}, [[0.2, 0.25], [0.8, 0.7], [0.5, 0.1]], None, None,
                        dict(reset_n_goals=2, reset_orders_goal=True, goalorder="nonsequential", agentmode="noninteract"),
                        0.0, False, 8, 500, 11.0),
        # the first goal sits behind a wall for lanes that start on the right: inside the radius, no line of sight
        "wall_seq_delay": ({"walls": [[[0.5, 0.0], [0.5, 0.6]]]}, [[0.45, 0.3], [0.25, 0.8], [0.75, 0.8]], 0.12,
                           [dict(init_state=2.0, expire_clock=0.3, decay="constant", decay_knobs=[0.5]),
                            dict(init_state=1.0, expire_clock=0.5, decay="exponential", decay_knobs=[0.2]),
                            dict(init_state=-1.0, expire_clock=0.5, decay="linear", decay_knobs=[6], dt=0.02)],
                           dict(reset_n_goals=3, reset_orders_goal=True, goalorder="sequential", agentmode="interact"),
                           0.05, False, 10, 700, 11.0),
        # two goals met on the same step: the second is consumed by the second pass of that step
        "overlap_pair": ({}, [[0.5, 0.5], [0.52, 0.5]], 0.12, None,
                         dict(reset_n_goals=2, reset_orders_goal=True, goalorder="nonsequential", agentmode="noninteract"),
                         0.0, True, 4, 300, 11.0),
        # overlapping goals: several are met on the same step (list-index skip, late completions)
        "overlap_teleport": ({}, [[0.5, 0.5], [0.52, 0.5], [0.5, 0.53], [0.2, 0.8]], 0.12, None,
                             dict(reset_n_goals=4, reset_orders_goal=True, goalorder="nonsequential", agentmode="interact"),
                             0.03, True, 8, 400, 14.0),
    }
    for name, (envp, gpos, radius, rewards, gckws, delay, teleport, n_lanes, n_steps, speed) in scenarios.items():
        rec = {k: [] for k in ("action", "z", "pos", "reward", "terminal", "goals_left", "n_rewards", "reset",
                               "teleport_pos", "pos0", "state0", "late", "episodes")}
        goal_table = None
        for lane in range(n_lanes):
            np.random.seed(1000 + 17 * lane)
            env = SpatialGoalEnvironment(params=dict(envp), possible_goal_positions=np.array(gpos), render_mode="none",
                                         goalcachekws=dict(gckws), episode_terminate_delay=delay,
                                         teleport_on_reset=teleport, goalkws=({} if radius is None else {"goal_radius": radius}))
            pool = env.goal_cache.reset_goals
            if rewards is not None:
                for g, rw in zip(pool, rewards):
                    if rw is not None:
                        g.reward = Reward(**dict({"dt": 0.01}, **rw))
                        g.reward.goal = g
            goal_table = np.array([[g.pos[0], g.pos[1], g.radius, g.reward.state, g.reward.dt, g.reward.expire_clock,
                                    presets[g.reward.preset], g.reward.decay_knobs[0]] for g in pool], float)
            Ag = Agent(env, {"dt": 0.01})
            env.add_agents(Ag)
            rec["state0"].append(np.concatenate([np.ravel(x) for x in _get_state(Ag)]))
            L = {k: [] for k in ("action", "z", "pos", "reward", "terminal", "goals_left", "n_rewards", "reset",
                                 "teleport_pos", "late")}
            _capture["on"] = True
            for t in range(n_steps):
                # policy (input data only): head for the first pending goal (sequential) / the nearest one
                pend = [g.pos for g in env.goal_cache.goals[Ag.name] if isinstance(g, SpatialGoal)]
                if not pend:
                    v = np.zeros(2)
                elif gckws["goalorder"] == "sequential":
                    v = pend[0] - Ag.pos
                else:
                    v = min((p_ - Ag.pos for p_ in pend), key=np.linalg.norm)
                nv = np.linalg.norm(v)
                action = speed * Ag.speed_mean * (v / nv) if nv > 0 else np.array([np.nan, np.nan])
                _rec["normal"].clear()
                active_before = len(env.agents) > 0
                if not active_before:  # the reference would raise (TaskEnvironment.py:384-391)
                    raise RuntimeError("inactive agent stepped")
                obs, rew, term, trunc, info = env.step({Ag.name: np.array(action)})
                scal = [z for shp, z in _rec["normal"] if shp == ()]
                assert len(scal) == 2, len(scal)
                left = len(env.goal_cache.goals[Ag.name])
                late = (left == 0) and not term[Ag.name]
                L["action"].append(action)
                L["z"].append([float(scal[0]), float(scal[1])])
                L["pos"].append(np.array(Ag.pos, float))
                L["reward"].append(float(rew[Ag.name]))
                L["terminal"].append(bool(term[Ag.name]))
                L["goals_left"].append(left)
                L["n_rewards"].append(len(Ag.reward.cache))
                L["late"].append(bool(late))
                if left == 0:  # (reset also after a late completion: the reference's next step would raise)
                    _capture["on"] = False
                    env.reset()
                    _capture["on"] = True
                    L["reset"].append(True)
                    L["teleport_pos"].append(np.array(Ag.pos, float))
                else:
                    L["reset"].append(False)
                    L["teleport_pos"].append(np.array([np.nan, np.nan]))
            _capture["on"] = False
            for k in L:
                rec[k].append(np.array(L[k]))
            ep = env.episodes
            rec["episodes"].append(np.array([[e, s, en, d] for e, s, en, d in
                                             zip(ep["episode"], ep["start"], ep["end"], ep["duration"])], float).reshape(-1, 4))
            rec["pos0"].append(rec["state0"][-1][0:2])
        out = {k: np.array(v) for k, v in rec.items() if k != "episodes"}
        n_ep = max(len(e) for e in rec["episodes"])
        eps = np.full((n_lanes, n_ep, 4), np.nan)
        for i, e in enumerate(rec["episodes"]):
            eps[i, :len(e)] = e
        out.update(episodes=eps, goal_table=goal_table, user_walls=np.array(envp.get("walls", []), float).reshape(-1, 2, 2),
                   env_bc=str(envp.get("boundary_conditions", "solid")), goalorder=str(gckws["goalorder"]),
                   reset_n_goals=int(gckws["reset_n_goals"]), terminate_delay=float(delay), teleport=bool(teleport),
                   dt=0.01, final_episode=np.array([0]))
        print(f"  {name}: {n_lanes} lanes x {n_steps} steps, resets {int(out['reset'].sum())}, "
              f"late completions {int(out['late'].sum())}, max rewards in cache {int(out['n_rewards'].max())}")
        np.savez_compressed(os.path.join(HERE, f"task_{name}.npz"), **out)



def make_task_world():
    """TaskEnvironment.step / reset with SEVERAL agents in ONE environment, agentmode="interact" (the reference's
    default, contribs/TaskEnvironment.py:1030, 1154-1172): one shared episode, a goal is consumed for everybody by the
    first agent — in `agent_names` order — found inside it.  Per step: every agent's action and two OU normals, its
    position and reward total afterwards, its reward-cache size; the returned terminal flag, the length of the shared
    goal list and its pool indices."""
    from ratinabox.contribs.TaskEnvironment import SpatialGoalEnvironment, SpatialGoal, Reward
    _section("task world")
    presets = {"constant": 0, "linear": 1, "exponential": 2, "none": 3}
    scenarios = {
        # name: (env params, goal positions, radius, rewards, goalcachekws, delay, teleport, n agents, n steps, speed,
        #        start positions (None = the reference's own random ones))
        "nonseq": ({}, [[0.2, 0.25], [0.8, 0.7], [0.5, 0.1], [0.3, 0.8], [0.7, 0.2]], None, None,
                   dict(reset_n_goals=4, reset_orders_goal=True, goalorder="nonsequential", agentmode="interact"),
                   0.0, False, 6, 500, 9.0, None),
        # a wall between the goals; sequential order: the head of the shared list moves on within one pass
        "seq_delay": ({"walls": [[[0.5, 0.0], [0.5, 0.6]]]}, [[0.45, 0.3], [0.25, 0.8], [0.75, 0.8], [0.55, 0.3]], 0.12,
                      [dict(init_state=2.0, expire_clock=0.3, decay="constant", decay_knobs=[0.5]),
                       dict(init_state=1.0, expire_clock=0.5, decay="linear", decay_knobs=[3]),
                       dict(init_state=-1.0, expire_clock=0.5, decay="linear", decay_knobs=[6], dt=0.02),
                       dict(init_state=0.5, expire_clock=0.2, decay="none", decay_knobs=[0])],
                      dict(reset_n_goals=4, reset_orders_goal=True, goalorder="sequential", agentmode="interact"),
                      0.05, False, 5, 800, 14.0, None),
        # overlapping goals and agents that travel together: several agents stand in several goals on the same step
        # (list-index skip after a pop, goals that pass to a later agent of the same pass, late completions)
        "overlap_teleport": ({}, [[0.5, 0.5], [0.52, 0.5], [0.5, 0.53], [0.2, 0.8], [0.21, 0.79]], 0.12, None,
                             dict(reset_n_goals=5, reset_orders_goal=True, goalorder="nonsequential", agentmode="interact"),
                             0.03, True, 8, 600, 14.0, "cluster"),
        # sequential, overlapping heads, a cluster of agents: more than one head consumed in one pass
        "seq_overlap": ({}, [[0.5, 0.5], [0.52, 0.5], [0.5, 0.53], [0.8, 0.2]], 0.12, None,
                        dict(reset_n_goals=4, reset_orders_goal=True, goalorder="sequential", agentmode="interact"),
                        0.0, True, 6, 500, 14.0, "cluster"),
    }
    for name, (envp, gpos, radius, rewards, gckws, delay, teleport, n_agents, n_steps, speed, start) in scenarios.items():
        np.random.seed(4000 + len(name))
        env = SpatialGoalEnvironment(params=dict(envp), possible_goal_positions=np.array(gpos), render_mode="none",
                                     goalcachekws=dict(gckws), episode_terminate_delay=delay, teleport_on_reset=teleport,
                                     goalkws=({} if radius is None else {"goal_radius": radius}))
        pool = env.goal_cache.reset_goals
        if rewards is not None:
            for g, rw in zip(pool, rewards):
                g.reward = Reward(**dict({"dt": 0.01}, **rw))
                g.reward.goal = g
        goal_table = np.array([[g.pos[0], g.pos[1], g.radius, g.reward.state, g.reward.dt, g.reward.expire_clock,
                                presets[g.reward.preset], g.reward.decay_knobs[0]] for g in pool], float)
        index_of = {id(g): i for i, g in enumerate(pool)}
        Ags = [Agent(env, {"dt": 0.01}) for _ in range(n_agents)]
        env.add_agents(Ags)
        names = list(env.agent_names)

        def cluster():
            c = np.array([0.78, 0.3])   # (the agents with the higher indices are nearer to the goals: they arrive first)
            for i, A in enumerate(Ags):
                A.pos = c + 0.03 * np.array([-(i % 3), i // 3], float)

        if start == "cluster":
            cluster()
        state0 = np.array([np.concatenate([np.ravel(x) for x in _get_state(A)]) for A in Ags])
        rec = {k: [] for k in ("action", "z", "pos", "reward", "terminal", "goals_left", "goal_list", "n_rewards", "reset",
                               "teleport_pos", "late")}
        _capture["on"] = True
        for t in range(n_steps):
            shared = env.goal_cache.goals[names[0]]
            pend = [g.pos for g in shared if isinstance(g, SpatialGoal)]
            acts = {}
            for nm, A in zip(names, Ags):   # policy (input data only): the head (sequential) / the nearest pending goal
                if not pend:
                    v = np.zeros(2)
                elif gckws["goalorder"] == "sequential":
                    v = pend[0] - A.pos
                else:
                    v = min((p_ - A.pos for p_ in pend), key=np.linalg.norm)
                nv = np.linalg.norm(v)
                acts[nm] = speed * A.speed_mean * (v / nv) if nv > 0 else np.array([np.nan, np.nan])
            _rec["normal"].clear()
            if len(env.agents) != n_agents:
                raise RuntimeError("inactive agents stepped")
            obs, rew, term, trunc, info = env.step({k: np.array(v) for k, v in acts.items()})
            scal = [z for shp, z in _rec["normal"] if shp == ()]
            assert len(scal) == 2 * n_agents, len(scal)
            lists = [env.goal_cache.goals[nm] for nm in names]
            assert all([id(g) for g in l] == [id(g) for g in lists[0]] for l in lists)   # one shared list
            left = len(lists[0])
            assert len(set(term.values())) == 1
            term = bool(term[names[0]])
            rec["action"].append(np.array([acts[nm] for nm in names]))
            rec["z"].append(np.array(scal, float).reshape(n_agents, 2))
            rec["pos"].append(np.array([A.pos for A in Ags], float))
            rec["reward"].append(np.array([float(rew[nm]) for nm in names]))
            rec["terminal"].append(term)
            rec["goals_left"].append(left)
            gl = np.full(16, -1)
            gl[:left] = [index_of.get(id(g), -2) for g in lists[0]]
            rec["goal_list"].append(gl)
            rec["n_rewards"].append(np.array([len(A.reward.cache) for A in Ags]))
            rec["late"].append((left == 0) and not term)
            if left == 0:   # (reset also after a late completion: the reference's next step would raise)
                _capture["on"] = False
                env.reset()
                if start == "cluster" and teleport:   # (keeps the agents travelling together: input data)
                    cluster()
                _capture["on"] = True
                rec["reset"].append(True)
                rec["teleport_pos"].append(np.array([A.pos for A in Ags], float))
            else:
                rec["reset"].append(False)
                rec["teleport_pos"].append(np.full((n_agents, 2), np.nan))
        _capture["on"] = False
        out = {k: np.array(v) for k, v in rec.items()}
        ep = env.episodes
        out.update(episodes=np.array([[e, s, en, d] for e, s, en, d in zip(ep["episode"], ep["start"], ep["end"], ep["duration"])],
                                     float).reshape(-1, 4),
                   state0=state0, goal_table=goal_table, user_walls=np.array(envp.get("walls", []), float).reshape(-1, 2, 2),
                   goalorder=str(gckws["goalorder"]), reset_n_goals=int(gckws["reset_n_goals"]), terminate_delay=float(delay),
                   teleport=bool(teleport), dt=0.01)
        awards = np.diff(np.concatenate([[np.zeros(n_agents)], out["n_rewards"]]), axis=0)
        print(f"  {name}: {n_agents} agents x {n_steps} steps, resets {int(out['reset'].sum())}, late completions "
              f"{int(out['late'].sum())}, steps with awards to > 1 agent {int(((awards > 0).sum(axis=1) > 1).sum())}, "
              f"awards per agent {(np.maximum(awards, 0)).sum(axis=0).astype(int).tolist()}, "
              f"max rewards in a cache {int(out['n_rewards'].max())}")
        np.savez_compressed(os.path.join(HERE, f"taskworld_{name}.npz"), **out)

    # ---- the list logic alone: the reference's GoalCache.check over random "who stands in which goal" tables ----
    from ratinabox.contribs.TaskEnvironment import GoalCache, Goal

    class _TableGoal(Goal):   # (a goal whose check() reads a table: input data; the cache's walk is the reference's)
        def __init__(self, env, index):
            super().__init__(env, reward=Reward(1, 0.01, expire_clock=1, decay="linear"))   # (its own: reward.goal names it)
            self.index = index

        def check(self, agents=None):
            return {a: self.reward for a in self.env._agentnames(agents) if self.env.met[self.env.agent_names.index(a), self.index]}

    class _FakeEnv:
        def __init__(self, n_agents):
            self.agent_names = [f"agent_{i}" for i in range(n_agents)]
            self.Ags = {nm: None for nm in self.agent_names}

        def _agentnames(self, agents=None):
            if agents is None:
                return self.agent_names
            return [agents] if isinstance(agents, str) else list(agents)

    rs = np.random.RandomState(99)
    n_cases, A_MAX, G_MAX, P_MAX = 400, 12, 8, 3
    met_all = np.zeros((n_cases, A_MAX, G_MAX), bool)
    dims = np.zeros((n_cases, 3), int)          # agents, goals, sequential
    award_agent = np.full((n_cases, P_MAX, G_MAX), -1)
    award_goal = np.full((n_cases, P_MAX, G_MAX), -1)
    left_after = np.full((n_cases, P_MAX, G_MAX), -1)
    for c in range(n_cases):
        na, ng, seq = rs.randint(1, A_MAX + 1), rs.randint(1, G_MAX + 1), c % 2
        fe = _FakeEnv(na)
        fe.met = rs.random_sample((na, ng)) < rs.choice([0.05, 0.2, 0.5, 0.9])
        goals = [_TableGoal(fe, i) for i in range(ng)]
        gc = GoalCache(fe, goalorder="sequential" if seq else "nonsequential", agentmode="interact", reset_goals=goals,
                       reset_n_goals=ng, reset_orders_goal=True)
        gc.reset()
        met_all[c, :na, :ng] = fe.met
        dims[c] = na, ng, seq
        for p_ in range(P_MAX):
            before = [g.index for g in gc.goals[fe.agent_names[0]]]
            rewards, agents = gc.check(remove_finished=True)
            after = [g.index for g in gc.goals[fe.agent_names[0]]]
            assert all([g.index for g in gc.goals[nm]] == after for nm in fe.agent_names)
            gone = [g for g in before if g not in after]
            assert sorted(gone) == sorted(r.goal.index for r in rewards) and len(gone) == len(agents)
            award_agent[c, p_, :len(agents)] = [fe.agent_names.index(a) for a in agents]
            award_goal[c, p_, :len(agents)] = [r.goal.index for r in rewards]
            left_after[c, p_, :len(after)] = after
    np.savez_compressed(os.path.join(HERE, "taskworld_list_logic.npz"), met=met_all, dims=dims, award_agent=award_agent,
                        award_goal=award_goal, left_after=left_after)
    print(f"  list logic: {n_cases} tables, {int((award_agent >= 0).sum())} awards, "
          f"{int(((award_agent >= 0).sum(axis=2) > 1).sum())} passes with more than one")


# ------------------------------------------------------------- BASELINE cfg 1 -- #
def make_cfg1():
    """BASELINE.json configs[0] / README.md:43: 1 agent, 1 m box, 100 gaussian PlaceCells, dt = 10 ms, 60 s =
    6000 `Ag.update(); PCs.update()` steps of the reference with the OU normals recorded.  Kept: the normals, the
    trajectory, the firing rates of every 50th step and of the last one."""
    _section("cfg 1: 1 agent x 100 PlaceCells x 6000 steps")
    np.random.seed(0)
    Env = Environment()
    Ag = Agent(Env, {"dt": 0.01})
    PCs = PlaceCells(Ag, {"n": 100})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    T = 6000
    state0 = np.concatenate([np.ravel(x) for x in _get_state(Ag)])
    z = np.zeros((T, 2))
    pos = np.zeros((T + 1, 2))
    pos[0] = Ag.pos
    _capture["on"] = True
    bounces = 0
    for t in range(T):
        _rec["normal"].clear()
        _bounces["n"] = 0
        Ag.update()
        scal = [v for shp, v in _rec["normal"] if shp == ()]
        assert len(scal) == 2
        z[t] = scal
        bounces += _bounces["n"]
        PCs.update()
        pos[t + 1] = Ag.pos
    _capture["on"] = False
    fr = np.array(PCs.history["firingrate"])
    out = dict(state0=state0, z=z, pos=pos, centres=np.array(PCs.place_cell_centres), widths=np.array(PCs.place_cell_widths),
               rates_every_50=fr[49::50], rates_last=fr[-1], head_direction=np.array(Ag.history["head_direction"])[49::50],
               distance_travelled=float(Ag.distance_travelled), n_bounces=bounces, t_end=float(Ag.t))
    np.savez_compressed(os.path.join(HERE, "cfg1.npz"), **out)
    print(f"  cfg1.npz: {T} steps, {bounces} bounces, distance {out['distance_travelled']:.3f} m")


# ------------------------------------------------------------------ row a6 -- #
def make_polygon():
    """Polygonal boundary and holes away from the motion records: wall table order, the strict inside test on
    random and on-the-edge points, the samplers (seeded), apply_boundary_conditions with its recorded random
    replacements, PlaceCells under the three wall geometries in an L-shaped room."""
    _section("polygon boundary + holes")
    out = {}
    envs = {"lroom": {"boundary": L_ROOM, "walls": [[[0.25, 0.0], [0.25, 0.3]]]},
            "holes": {"holes": [HOLE_A, HOLE_B], "walls": [[[0.8, 0.1], [0.8, 0.4]]]},
            "both": {"boundary": [[0, 0], [2, 0], [2.5, 1], [1, 1.5], [-0.2, 1]], "holes": [[[0.5, 0.4], [1.0, 0.4], [0.7, 0.8]]]}}
    rs = np.random.RandomState(4321)
    for tag, ep in envs.items():
        Env = Environment(ep)
        out[f"{tag}_boundary"] = np.array(ep.get("boundary") if ep.get("boundary") is not None else np.zeros((0, 2)), float)
        out[f"{tag}_holes"] = np.array([c for h in ep.get("holes", []) for c in h], float).reshape(-1, 2)
        out[f"{tag}_hole_sizes"] = np.array([len(h) for h in ep.get("holes", [])], np.int64)
        out[f"{tag}_user_walls"] = np.array(ep.get("walls", []), float).reshape(-1, 2, 2)
        out[f"{tag}_walls"] = np.array(Env.walls, float)
        out[f"{tag}_extent"] = np.array(Env.extent, float)
        ex = Env.extent
        P = 400
        pts = np.stack((rs.uniform(ex[0] - 0.1, ex[1] + 0.1, P), rs.uniform(ex[2] - 0.1, ex[3] + 0.1, P)), -1)
        # points exactly on edges and corners of the boundary and of the holes
        polys = [np.array(Env.boundary, float)] + [np.array(h, float) for h in ep.get("holes", [])]
        edge_pts = []
        for poly in polys:
            for i in range(len(poly)):
                a, b = poly[i], poly[(i + 1) % len(poly)]
                edge_pts += [a, 0.5 * (a + b), a + 0.25 * (b - a)]
        pts = np.vstack((pts, np.array(edge_pts)))
        out[f"{tag}_points"] = pts
        out[f"{tag}_inside"] = np.array([Env.check_if_position_is_in_environment(p) for p in pts])
        # the samplers, global np.random seeded (the product draws in the same order)
        for method in ("random", "uniform", "uniform_jitter"):
            np.random.seed(77)
            out[f"{tag}_sample_{method}"] = Env.sample_positions(n=53, method=method)
        # apply_boundary_conditions on all the points, with the replacement each resample drew
        _capture["on"] = False
        np.random.seed(5)
        new, rsp = [], []
        for p in pts:
            _resampled["pos"] = None
            q = Env.apply_boundary_conditions(np.array(p, float))
            new.append(np.array(q, float).reshape(-1))
            rsp.append(_resampled["pos"] if _resampled["pos"] is not None else np.full(2, np.nan))
        out[f"{tag}_bc_out"] = np.array(new)
        out[f"{tag}_bc_resample"] = np.array(rsp)
    # PlaceCells in the L room: euclidean, line_of_sight (internal walls = walls[4:], whatever the boundary has:
    # Environment.py:715-717), and geodesic in a quadrilateral room with one wall
    Env = Environment(envs["lroom"])
    Ag = Agent(Env)
    pos = np.stack((rs.uniform(0, 1, 300), rs.uniform(0, 1, 300)), -1)
    pos = pos[[Env.check_if_position_is_in_environment(p) for p in pos]][:160]
    out["pc_pos"] = f32exact(pos)
    np.random.seed(3)
    for geom in ("euclidean", "line_of_sight"):
        PCs = PlaceCells(Ag, {"n": 40, "widths": 0.15, "wall_geometry": geom, "description": "gaussian_threshold"})
        out[f"pc_{geom}_centres"] = f32exact(PCs.place_cell_centres)
        PCs.place_cell_centres = out[f"pc_{geom}_centres"]
        out[f"pc_{geom}_rates"] = PCs.get_state(evaluate_at=None, pos=out["pc_pos"])
    quad = {"boundary": [[0, 0], [1.2, 0.1], [1.0, 1.0], [0.1, 0.8]], "walls": [[[0.6, 0.05], [0.55, 0.5]]]}
    out["quad_boundary"] = np.array(quad["boundary"], float)
    out["quad_user_walls"] = np.array(quad["walls"], float)
    EnvQ = Environment(quad)
    AgQ = Agent(EnvQ)
    posq = np.stack((rs.uniform(0, 1.2, 400), rs.uniform(0, 1.0, 400)), -1)
    posq = posq[[EnvQ.check_if_position_is_in_environment(p) for p in posq]][:160]
    out["quad_pos"] = f32exact(posq)
    PCs = PlaceCells(AgQ, {"n": 30, "widths": 0.2, "wall_geometry": "geodesic"})
    out["quad_centres"] = f32exact(PCs.place_cell_centres)
    PCs.place_cell_centres = out["quad_centres"]
    out["quad_rates"] = PCs.get_state(evaluate_at=None, pos=out["quad_pos"])
    np.savez_compressed(os.path.join(HERE, "polygon.npz"), **out)
    print("  polygon.npz written:", len(out), "arrays")


def make_stats():
    """G6: long-run statistics of the reference's own motion model (its own NumPy RNG), for validating the
    production (in-kernel Philox) mode, which cannot be compared draw by draw."""
    _section("long-run statistics (this takes a few minutes)")
    out = {}
    for name, envp in (("open", {}), ("wall", {"walls": [[[0.5, 0.0], [0.5, 0.6]]]})):
        np.random.seed(77)
        Env = Environment(dict(envp))
        n_agents, n_steps, burn = 40, 4000, 250
        agents = [Agent(Env, {"dt": 0.02}) for _ in range(n_agents)]
        speed, rot, dwall, pos = [], [], [], []
        for t in range(n_steps):
            for ag in agents:
                ag.update()
            if t >= burn:
                speed.append([np.linalg.norm(a.velocity) for a in agents])
                rot.append([a.rotational_velocity for a in agents])
                dwall.append([a.distance_to_closest_wall for a in agents])
                pos.append([a.pos for a in agents])
        speed, rot, dwall, pos = map(np.array, (speed, rot, dwall, pos))
        out[f"{name}_speed_mean"] = speed.mean()
        out[f"{name}_speed_std"] = speed.std()
        out[f"{name}_speed_q"] = np.quantile(speed, [0.1, 0.5, 0.9])
        out[f"{name}_rot_std"] = rot.std()
        out[f"{name}_dwall_hist"] = np.histogram(dwall, bins=10, range=(0, 0.5))[0] / dwall.size
        out[f"{name}_pos_hist"] = np.histogram2d(pos[..., 0].ravel(), pos[..., 1].ravel(), bins=4, range=[[0, 1], [0, 1]])[0] / (pos.size / 2)
        out[f"{name}_walls"] = np.array(envp.get("walls", []), float).reshape(-1, 2, 2)
        # the per-agent means give the standard error the tests use
        out[f"{name}_speed_agent_means"] = speed.mean(axis=0)
        print(f"  {name}: speed {speed.mean():.4f} +- {speed.mean(axis=0).std() / np.sqrt(n_agents):.4f}, rot std {rot.std():.3f}")
    out["dt"], out["n_steps"], out["burn"] = 0.02, 4000, 250
    np.savez_compressed(os.path.join(HERE, "stats.npz"), **out)


def make_helpers():
    """The reference's public geometry / statistics helpers (ratinabox/utils.py) on seeded inputs: what
    ratinabox_amd/utils.py offers under the same names (tests/test_host_logic.py::test_public_helpers_vs_reference).
    The anti-degeneracy jitter of vector_intercepts / shortest_vectors_from_points_to_lines is patched to zero like
    everywhere else in this file; ornstein_uhlenbeck's normals are recorded."""
    _section("helpers")
    rs = np.random.RandomState(77)
    out = {}
    a, b, p, q = rs.rand(7, 2, 2), rs.rand(5, 2, 2), rs.rand(9, 2), rs.rand(4, 2)
    x, th = rs.randn(4, 5) * 6, rs.rand(3, 4) * 6
    vs, segs = rs.randn(6, 2), rs.rand(6, 2, 2)
    out.update(a=a, b=b, p=p, q=q, x=x, th=th, vs=vs, segs=segs)
    out["vi"] = rutils.vector_intercepts(a, b)          # (np.random.normal is patched: jitter of scale 1e-9 / 1e-6 -> 0)
    out["vi_hit"] = rutils.vector_intercepts(a, b, return_collisions=True)
    out["sv"] = rutils.shortest_vectors_from_points_to_lines(p, b)
    out["segs_pq"] = rutils.get_line_segments_between(p, q)
    out["vec_pq"] = rutils.get_vectors_between(p, q)
    out["dist_pq"] = rutils.get_distances_between(p, q)
    out["angle_vs"] = rutils.get_angle(vs, is_array=True)
    out["angle_segs"] = rutils.get_angle(segs, is_array=True)
    out["bearing_vs"] = rutils.get_bearing(vs, is_array=True)
    out["bearing_segs"] = rutils.get_bearing(segs, is_array=True)
    out["perp_vs"] = np.stack([rutils.get_perpendicular(v) for v in vs])
    out["bounce"] = np.stack([rutils.wall_bounce(v, w) for v, w in zip(vs, segs)])
    out["pi_domain"] = rutils.pi_domain(x)
    z = rs.normal(size=x.shape)
    np.random.normal = lambda loc=0.0, scale=1.0, size=None: z * scale
    try:
        out["ou"] = rutils.ornstein_uhlenbeck(0.01, x, drift=0.5, noise_scale=0.3, coherence_time=0.7)
    finally:
        np.random.normal = _patched_normal
    out["ou_z"] = z
    out["n2r"] = rutils.normal_to_rayleigh(x / 3, 0.08)
    speeds = np.array([0.0, 0.01, 0.05, 0.08, 0.3, 2.0])
    out["speeds"] = speeds
    out["r2n"] = np.array([rutils.rayleigh_to_normal(float(v), 0.08) for v in speeds])
    for k, norm in (("d", None), ("1", 1), ("2p5", 2.5)):
        out["gauss_" + k] = rutils.gaussian(th, 0.4, 0.3, norm)
        out["vm_" + k] = rutils.von_mises(th, 0.4, 0.3, norm)
    for name in ("linear", "sigmoid", "relu", "tanh", "retanh", "softmax"):
        oa = {"max_fr": 3, "min_fr": 0.5, "mid_x": 0.2, "width_x": 1.5} if name == "sigmoid" else {"gain": 2.0, "threshold": 0.3}
        for tag, args in (("dflt", {}), ("args", oa)):
            for deriv in (False, True):
                out[f"act_{name}_{tag}_{int(deriv)}"] = np.asarray(rutils.activate(x / 3, name, deriv, dict(args)), dtype=float)
    # BoundaryVectorCells.boundary_vector_preference_function on (l_a, l_b) pairs, the exact zeros and ones included
    lam = rs.randn(40, 7, 2) * 1.5
    lam[0, 0], lam[0, 1], lam[0, 2], lam[0, 3] = [0.0, 0.5], [0.0, 2.0], [0.5, 0.0], [0.5, 1.0]
    np.random.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bv = BoundaryVectorCells(Agent(Environment()), params={"n": 3})
        out["pref"] = bv.boundary_vector_preference_function(lam)
    out["pref_lam"] = lam
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)


def comb_walls(n=60):
    """bench.py's wall-heavy room (cfg3_64w): n interior segments, the teeth of two interleaved combs with doorways."""
    walls, teeth = [], n // 2
    for i in range(teeth):
        x = (i + 1) / (teeth + 1)
        if i % 2 == 0:
            walls += [[[x, 0.0], [x, 0.3]], [[x, 0.4], [x, 0.7]]]
        else:
            walls += [[[x, 1.0], [x, 0.7]], [[x, 0.6], [x, 0.3]]]
    return walls[:n]


def make_walls64():
    """A room at RIAB_MAX_WALLS (60 interior segments + the box's four): motion steps — every position has
    several walls within the repel distance, collisions are frequent —, boundary vector cells over 64 walls, line-of-sight
    PlaceCells over 60 internal walls."""
    _section("64 walls")
    comb = comb_walls(60)
    motion_records("comb60_dt10ms", {"walls": comb}, {"dt": 0.01}, 16, 250, seed=31)
    motion_records("comb60_fast", {"walls": comb}, {"dt": 0.05, "speed_mean": 0.3, "thigmotaxis": 0.8}, 16, 150, seed=32)
    np.random.seed(33)
    out = {"walls": np.array(comb, float)}
    rs = np.random.RandomState(34)
    pos = np.stack((rs.uniform(0, 1, 128), rs.uniform(0, 1, 128)), -1)
    pos[:16, 0] = np.round(pos[:16, 0] * 31) / 31 + rs.choice([-1e-3, 1e-3], 16)   # next to a tooth
    pos[:, 0] = np.clip(pos[:, 0], 1e-3, 1 - 1e-3)
    pos = f32exact(pos)
    out["pos"] = pos
    Env = Environment({"walls": comb})
    Ag = Agent(Env)
    assert len(Env.walls) == 64
    B = BoundaryVectorCells(Ag, {"n": 24, "min_fr": 0.0, "max_fr": 1.0})
    for k in ["tuning_distances", "tuning_angles", "sigma_distances", "sigma_angles", "cell_fr_norm"]:
        out[f"bvc_{k}"] = np.array(getattr(B, k), float)
    out["bvc_rates"] = B.get_state(evaluate_at=None, pos=pos)
    PCs = PlaceCells(Ag, {"n": 40, "widths": 0.12, "wall_geometry": "line_of_sight"})
    PCs.place_cell_centres = f32exact(PCs.place_cell_centres)
    out["pc_los_centres"] = PCs.place_cell_centres
    out["pc_los_rates"] = PCs.get_state(evaluate_at=None, pos=pos)
    out["ref_walls"] = np.array(Env.walls, float)
    out["vectors_from_walls"] = np.array([Env.vectors_from_walls(p) for p in pos[:32]])
    np.savez_compressed(os.path.join(HERE, "walls64.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["motion", "rates", "update", "imported", "feedforward", "ovc", "avc", "velocity", "env", "random_spatial", "task", "task_world", "polygon", "cfg1", "stats", "helpers", "walls64"]
    if "--out" in which:  # write somewhere else (tools/check_golden.py regenerates into a temporary directory)
        HERE = which[which.index("--out") + 1]
        which = [w for i, w in enumerate(which) if w != "--out" and (i == 0 or which[i - 1] != "--out")] or \
            ["motion", "rates", "update", "imported", "feedforward", "ovc", "avc", "velocity", "env", "random_spatial", "task",
             "task_world", "polygon", "cfg1", "stats", "helpers", "walls64"]
        os.makedirs(HERE, exist_ok=True)
    if "helpers" in which:
        make_helpers()
    if "walls64" in which:
        make_walls64()
    if "polygon" in which:
        make_polygon()
    if "cfg1" in which:
        make_cfg1()
    if "stats" in which:
        make_stats()
    if "task" in which:
        make_task()
    if "task_world" in which:
        make_task_world()
    if "random_spatial" in which:
        make_random_spatial()
    if "env" in which:
        make_env_queries()
    if "velocity" in which:
        make_velocity()
    if "avc" in which:
        make_avc()
    if "ovc" in which:
        make_ovc()
    if "feedforward" in which:
        make_feedforward()
    if "imported" in which:
        make_imported()
    if "motion" in which:
        make_motion()
    if "rates" in which:
        make_rates()
    if "update" in which:
        make_update_and_init()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f}: {os.path.getsize(os.path.join(HERE, f)) / 1024:.0f} KiB")

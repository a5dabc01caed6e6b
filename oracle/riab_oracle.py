"""CPU oracle: a float64 NumPy restatement of RatInABox's per-step hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import this module; the product
package `ratinabox_amd` never does (it fails loudly if the HIP library is missing).

Every function restates, batched over agents/positions, the arithmetic of one
reference function and cites it (paths relative to /root/reference).  The
restatement is pinned against the reference itself: `tests/golden/make_golden.py`
imports the reference in the build container, drives it with captured noise and
stores input/output vectors under `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks this module against those vectors (float64, rtol ~1e-12).  The reference's
own test-suite holds no numeric assertion on this path (tests/test_agent.py and
tests/test_neurons.py are empty), so these generated vectors are the pin.

Conventions
-----------
* positions `pos` are `(P, 2)` float64, firing rates are `(n_cells, P)` like the
  reference's `get_state(evaluate_at=None, pos=...)` (Neurons.py:943-949).
* walls are `(N_w, 2, 2)` in `Environment.walls` order (Environment.py:128-163).
* the reference's random geometric jitter (utils.py:64-69, 143-144) is NOT
  applied (the golden vectors are generated with it patched to zero); divisions
  by zero follow IEEE semantics exactly like the zero-jitter reference.
"""
from __future__ import annotations

import numpy as np
from scipy import special as _sp

TWO_PI = 2 * np.pi


# --------------------------------------------------------------------------- #
# small helpers (utils.py)
# --------------------------------------------------------------------------- #
def get_angle(vec):
    """utils.get_angle (utils.py:231-273) for direction vectors `(..., 2)`:
    `mod(arctan2(y, x + 1e-6), 2*pi)`."""
    vec = np.asarray(vec, dtype=np.float64)
    return np.mod(np.arctan2(vec[..., 1], vec[..., 0] + 1e-6), TWO_PI)


def pi_domain(x):
    """utils.pi_domain (utils.py:331-341): `x mod 2pi`, minus 2pi where > pi."""
    x = np.asarray(x, dtype=np.float64) % TWO_PI
    return np.where(x > np.pi, -TWO_PI + x, x)


def ou_increment(x, dt, drift, noise_scale, coherence_time, z):
    """utils.ornstein_uhlenbeck (utils.py:347-368) with the standard-normal draw
    `z` made explicit: the reference draws `normal(scale=dt)` = `dt*z`."""
    sigma = np.sqrt((2 * noise_scale**2) / (coherence_time * dt))
    theta = 1 / coherence_time
    return theta * (drift - x) * dt + sigma * (dt * z)


def rayleigh_to_normal(x, sigma):
    """utils.rayleigh_to_normal (utils.py:416-421); stats.norm.ppf == special.ndtri."""
    u = 1 - np.exp(-(x**2) / (2 * sigma**2))
    u = np.minimum(np.maximum(1e-6, u), 1 - 1e-6)
    return _sp.ndtri(u)


def normal_to_rayleigh(x, sigma):
    """utils.normal_to_rayleigh (utils.py:409-413); stats.norm.cdf == special.ndtr."""
    u = _sp.ndtr(x)
    return sigma * np.sqrt(-2 * np.log(1 - u))


def gaussian(x, mu, sigma):
    """utils.gaussian(..., norm=1) (utils.py:424-438)."""
    return np.exp(-((x - mu) ** 2) / (2 * sigma**2))


def von_mises(theta, mu, sigma):
    """utils.von_mises(..., norm=1) (utils.py:441-457): the reference evaluates
    `exp(k cos) * (1/exp(k))`; restated as `exp(k (cos - 1))` (identical up to
    rounding, and does not overflow for small sigma — SURVEY App. C-13)."""
    kappa = 1 / (sigma**2)
    return np.exp(kappa * (np.cos(theta - mu) - 1.0))


# --------------------------------------------------------------------------- #
# geometry (utils.py:30-184)
# --------------------------------------------------------------------------- #
def segment_intercepts(seg_a, seg_b):
    """utils.vector_intercepts (utils.py:30-118) without the 1e-9 jitter.

    seg_a `(Na,2,2)`, seg_b `(Nb,2,2)` -> `(l_a, l_b)` each `(Na,Nb)`: the line
    parameters of the intersection of the infinite lines through each pair."""
    seg_a = np.asarray(seg_a, dtype=np.float64).reshape(-1, 2, 2)
    seg_b = np.asarray(seg_b, dtype=np.float64).reshape(-1, 2, 2)
    d0 = seg_b[None, :, 0, :] - seg_a[:, None, 0, :]  # (Na,Nb,2)
    sa = (seg_a[:, 1, :] - seg_a[:, 0, :])[:, None, :]  # (Na,1,2)
    sb = (seg_b[:, 1, :] - seg_b[:, 0, :])[None, :, :]  # (1,Nb,2)
    sa_p = np.stack((-sa[..., 1], sa[..., 0]), axis=-1)
    sb_p = np.stack((-sb[..., 1], sb[..., 0]), axis=-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        l_a = (d0[..., 0] * sb_p[..., 0] + d0[..., 1] * sb_p[..., 1]) / (
            sa[..., 0] * sb_p[..., 0] + sa[..., 1] * sb_p[..., 1]
        )
        l_b = ((-d0[..., 0]) * sa_p[..., 0] + (-d0[..., 1]) * sa_p[..., 1]) / (
            sb[..., 0] * sa_p[..., 0] + sb[..., 1] * sa_p[..., 1]
        )
    return l_a, l_b


def segments_collide(seg_a, seg_b):
    """`return_collisions=True` branch of utils.vector_intercepts (utils.py:99-106):
    both parameters strictly inside (0, 1)."""
    l_a, l_b = segment_intercepts(seg_a, seg_b)
    return (l_a > 0) & (l_a < 1) & (l_b > 0) & (l_b < 1)


def shortest_vectors_from_walls(pos, walls):
    """utils.shortest_vectors_from_points_to_lines (utils.py:121-184) without the
    1e-6 jitter.  pos `(P,2)`, walls `(Nw,2,2)` -> `(P,Nw,2)` vectors from the
    nearest point of each wall segment to each position."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    walls = np.asarray(walls, dtype=np.float64).reshape(-1, 2, 2)
    d = pos[:, None, :] - walls[None, :, 0, :]
    s = walls[:, 1, :] - walls[:, 0, :]
    l_v = (d[..., 0] * s[:, 0] + d[..., 1] * s[:, 1]) / (s[:, 0] * s[:, 0] + s[:, 1] * s[:, 1])
    l_v = np.where(l_v > 1, 1.0, l_v)
    l_v = np.where(l_v < 0, 0.0, l_v)
    return pos[:, None, :] - (walls[None, :, 0, :] + l_v[..., None] * s[None, :, :])


# --------------------------------------------------------------------------- #
# Environment queries (Environment.py)
# --------------------------------------------------------------------------- #
class EnvSpec:
    """The subset of `Environment` state the hot path reads (Environment.py:65-191): a 2D box
    `[0, aspect*scale] x [0, scale]` (solid or periodic) or a simple polygon `boundary` (solid), `walls` in
    reference order (Environment.py:128-163): the boundary's edges first when solid — edge i runs from corner
    i+1 to corner i —, then the user's walls, then the edges of the `holes`."""

    def __init__(self, scale=1.0, aspect=1.0, boundary_conditions="solid", walls=(), boundary=None, holes=()):
        self.scale = float(scale)
        self.aspect = float(aspect)
        self.boundary_conditions = boundary_conditions
        self.is_rectangular = boundary is None
        b = [[0, 0], [aspect * scale, 0], [aspect * scale, scale], [0, scale]] if boundary is None else \
            np.asarray(boundary, dtype=np.float64).reshape(-1, 2).tolist()
        self.boundary = np.asarray(b, dtype=np.float64)
        self.holes = [np.asarray(h, dtype=np.float64).reshape(-1, 2) for h in holes]
        user = np.asarray(walls, dtype=np.float64).reshape(-1, 2, 2)
        if boundary_conditions == "solid":
            # Environment.py:137-144: wall i runs from b[i+1] to b[i]
            nb = len(b)
            bw = np.array([[b[(i + 1) % nb], b[i]] for i in range(nb)], dtype=np.float64)
            self.walls = np.vstack((bw, user))
        else:
            assert boundary is None, "periodic boundary conditions need the rectangular box"
            self.walls = user
        for h in self.holes:  # Environment.py:154-161
            k = len(h)
            self.walls = np.vstack((self.walls, np.array([[h[(i + 1) % k], h[i]] for i in range(k)])))
        self.extent = np.array([self.boundary[:, 0].min(), self.boundary[:, 0].max(), self.boundary[:, 1].min(),
                                self.boundary[:, 1].max()])

    @property
    def periodic(self):
        return self.boundary_conditions == "periodic"


def env_vectors_between(env, pos1, pos2):
    """Environment.get_vectors_between___accounting_for_environment
    (Environment.py:657-675): pairwise `pos1[i] - pos2[j]`, wrapped by `scale`
    when periodic.  -> `(N1,N2,2)`."""
    v = np.asarray(pos1, dtype=np.float64).reshape(-1, 1, 2) - np.asarray(pos2, dtype=np.float64).reshape(1, -1, 2)
    if env.periodic:
        flip = np.abs(v) > (env.scale / 2)
        v = np.where(flip, -np.sign(v) * (env.scale - np.abs(v)), v)
    return v


def env_distances(env, pos1, pos2, wall_geometry="euclidean"):
    """Environment.get_distances_between___accounting_for_environment
    (Environment.py:677-779) for 2D: euclidean / line_of_sight / geodesic."""
    pos1 = np.asarray(pos1, dtype=np.float64).reshape(-1, 2)
    pos2 = np.asarray(pos2, dtype=np.float64).reshape(-1, 2)
    vec = env_vectors_between(env, pos1, pos2)
    dist = np.sqrt(vec[..., 0] ** 2 + vec[..., 1] ** 2)
    if wall_geometry == "euclidean":
        return dist
    segs = np.stack(
        (np.repeat(pos1[:, None, :], len(pos2), 1), np.repeat(pos2[None, :, :], len(pos1), 0)), axis=-2
    ).reshape(-1, 2, 2)
    if wall_geometry == "line_of_sight":
        internal = env.walls[4:]  # Environment.py:715-717
        if len(internal):
            blocked = segments_collide(segs, internal).sum(axis=-1) != 0
            dist = np.where(blocked.reshape(dist.shape), 1000.0, dist)
        return dist
    if wall_geometry == "geodesic":
        assert len(env.walls) <= 5  # Environment.py:736-739
        if len(env.walls) == 4:
            return dist
        wall = env.walls[4]
        via = []
        for e in wall:  # Environment.py:746-753: only endpoints strictly inside the env
            if env_is_inside(env, e[None])[0]:
                d1 = np.sqrt(((pos1 - e) ** 2).sum(-1))[:, None]
                d2 = np.sqrt(((e - pos2) ** 2).sum(-1))[None, :]
                via.append(d1 + d2)
        blocked = segments_collide(segs, wall[None]).reshape(dist.shape)
        if via:
            dist = np.where(blocked, np.amin(np.array(via), axis=0), dist)
        return dist
    raise ValueError(wall_geometry)


def polygon_contains(corners, pos):
    """`shapely.Polygon(corners).contains(Point(p))` for `(P, 2)` points: the STRICT interior (a point on an
    edge is not contained) by the even-odd rule — what Environment.check_if_position_is_in_environment calls
    for the boundary and for every hole (Environment.py:808-816)."""
    c = np.asarray(corners, dtype=np.float64).reshape(-1, 2)
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    px, py = pos[:, 0][:, None], pos[:, 1][:, None]
    a, b = c, np.roll(c, -1, axis=0)
    ax, ay, bx, by = a[None, :, 0], a[None, :, 1], b[None, :, 0], b[None, :, 1]
    cross = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
    on_edge = ((cross == 0) & (np.minimum(ax, bx) <= px) & (px <= np.maximum(ax, bx)) &
               (np.minimum(ay, by) <= py) & (py <= np.maximum(ay, by))).any(axis=1)
    straddle = (ay > py) != (by > py)
    with np.errstate(divide="ignore", invalid="ignore"):
        x_cross = ax + (py - ay) * (bx - ax) / (by - ay)
    odd = (np.count_nonzero(straddle & (px < x_cross), axis=1) % 2) == 1
    return odd & ~on_edge


def env_is_inside(env, pos):
    """Environment.check_if_position_is_in_environment (Environment.py:781-818): strictly inside the boundary
    (the rectangle, or the polygon) and not strictly inside any hole."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    if env.is_rectangular:
        e = env.extent
        inside = (pos[:, 0] > e[0]) & (pos[:, 0] < e[1]) & (pos[:, 1] > e[2]) & (pos[:, 1] < e[3])
    else:
        inside = polygon_contains(env.boundary, pos)
    for h in env.holes:
        inside = inside & ~polygon_contains(h, pos)
    return inside


def env_needs_resample(env, pos):
    """Which positions take the RESAMPLE branch of apply_boundary_conditions (Environment.py:886-893): outside a
    polygonal boundary, or inside the bounding box of a rectangular environment but not in it (i.e. in a hole)."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    e = env.extent
    in_box = (pos[:, 0] > e[0]) & (pos[:, 0] < e[1]) & (pos[:, 1] > e[2]) & (pos[:, 1] < e[3])
    return ~env_is_inside(env, pos) & (in_box | (not env.is_rectangular))


def env_apply_boundary_conditions(env, pos, resample_pos=None):
    """Environment.apply_boundary_conditions (Environment.py:855-894) for positions that are NOT inside: outside
    a rectangular box, solid -> clamp to [min+0.01, max-0.01], periodic -> modulo the extent; in a hole / outside
    a polygon -> `resample_pos` (the reference draws `sample_positions(n=1, method="random")` from np.random: the
    accepted draws are an input here)."""
    pos = np.array(pos, dtype=np.float64).reshape(-1, 2)
    e = env.extent
    if env.periodic:
        out = np.stack((pos[:, 0] % e[1], pos[:, 1] % e[3]), axis=-1)
    else:
        x = np.minimum(np.maximum(pos[:, 0], e[0] + 0.01), e[1] - 0.01)
        y = np.minimum(np.maximum(pos[:, 1], e[2] + 0.01), e[3] - 0.01)
        out = np.stack((x, y), axis=-1)
    rs = env_needs_resample(env, pos)
    if rs.any():
        assert resample_pos is not None, "positions in a hole / outside the polygon need resample positions"
        out = np.where(rs[:, None], np.asarray(resample_pos, dtype=np.float64).reshape(-1, 2), out)
    return out


def resample_draws(seed, step, agent_ids, env, max_attempts=64):
    """The production-mode resample (riab_agent_kernel.h): per attempt one Philox call keyed by
    (step lo, step hi ^ attempt << 24, agent id, TAG_MOTION ^ 2), two 24-bit uniforms over the extent, until the
    position is inside.  -> `(B, 2)` accepted positions."""
    agent_ids = np.asarray(agent_ids, dtype=np.uint64)
    out = np.zeros((len(agent_ids), 2))
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    e = env.extent
    for i, aid in enumerate(agent_ids):
        for attempt in range(max_attempts):
            w = philox4x32_10(np.uint32(step & 0xFFFFFFFF), np.uint32(((step >> 32) & 0xFFFFFFFF) ^ (attempt << 24)),
                              np.uint32(aid), np.uint32(TAG_MOTION ^ 2), k0, k1)
            u0 = float(np.float32(int(w[0]) >> 8) * np.float32(2.0 ** -24))
            u1 = float(np.float32(int(w[1]) >> 8) * np.float32(2.0 ** -24))
            out[i] = (e[0] + u0 * (e[1] - e[0]), e[2] + u1 * (e[3] - e[2]))
            if env_is_inside(env, out[i][None])[0]:
                break
    return out


# --------------------------------------------------------------------------- #
# Agent.update (Agent.py:160-242)
# --------------------------------------------------------------------------- #
DEFAULT_MOTION = dict(  # Agent.default_params (Agent.py:68-84)
    speed_coherence_time=0.7,
    speed_mean=0.08,
    speed_std=0.08,
    rotational_velocity_coherence_time=0.08,
    rotational_velocity_std=120 * (np.pi / 180),
    head_direction_smoothing_timescale=0.15,
    thigmotaxis=0.5,
    wall_repel_distance=0.1,
    wall_repel_strength=1.0,
)

MAX_BOUNCES = 16  # the reference loops `while True` (Agent.py:426); bounded here and on the GPU


def init_state(env, n_agents, speed_mean, rng):
    """Agent.initialise_position_and_velocity (Agent.py:523-535) + Agent.__init__
    (Agent.py:128-141), batched: uniform position, uniform heading."""
    pos = np.stack(
        (rng.uniform(env.extent[0], env.extent[1], n_agents), rng.uniform(env.extent[2], env.extent[3], n_agents)),
        axis=-1,
    )
    direction = rng.uniform(0, TWO_PI, n_agents)
    vel = speed_mean * np.stack((np.cos(direction), np.sin(direction)), axis=-1)
    return dict(
        pos=pos,
        velocity=vel.copy(),
        rotational_velocity=np.zeros(n_agents),
        measured_velocity=vel.copy(),
        measured_rotational_velocity=np.zeros(n_agents),
        head_direction=vel / np.linalg.norm(vel, axis=-1, keepdims=True),
        distance_travelled=np.zeros(n_agents),
        distance_to_closest_wall=np.full(n_agents, np.inf),
    )


def agent_step(env, state, dt, z_rot, z_speed, params=None, drift_velocity=None,
               drift_to_random_strength_ratio=1.0, z_zero=None, kwargs=None, forced_pos=None, resample_pos=None):
    """One `Agent.update()` (Agent.py:160-242, random-motion branch, 2D) for B
    independent agents.  `state` is a dict of `(B,...)` float64 arrays (see
    `init_state`); `z_rot`, `z_speed` `(B,)` are the two standard-normal draws
    the reference takes per update.  `params` are the Agent's
    attributes, `kwargs` the per-call overrides `Agent.update(**kwargs)` forwards
    to the sub-steps (Agent.py:280-285, 353-355) — note the reference reads some
    quantities from the attribute even when a kwarg is given (speed_std==0 switch
    :310, wall spring speed :375, bounce speed :439, drift tau :340).  Returns a
    new state dict plus `n_bounces (B,)` and `bc_applied (B,)` diagnostics.
    `forced_pos (B,2)`: the imported / forced-trajectory branches (Agent.py:229-238): the motion
    model is skipped, the agent is put at `forced_pos` and velocity / rotational velocity are
    overwritten by the measured ones.  `resample_pos (B,2)`: where an agent that ends the step in a hole / outside
    a polygonal boundary is put (the reference's np.random draw, Environment.py:886-893)."""
    p = dict(DEFAULT_MOTION)
    if params:
        p.update(params)
    kw = dict(kwargs or {})
    rotational_velocity_drift = kw.get("rotational_velocity_drift", 0.0)
    pos = np.array(state["pos"], dtype=np.float64)
    vel = np.array(state["velocity"], dtype=np.float64)
    rot = np.array(state["rotational_velocity"], dtype=np.float64)
    prev_mv = np.array(state["measured_velocity"], dtype=np.float64)
    hd = np.array(state["head_direction"], dtype=np.float64)
    dist_trav = np.array(state["distance_travelled"], dtype=np.float64)
    dclose = np.array(state["distance_to_closest_wall"], dtype=np.float64)
    B = pos.shape[0]
    prev_pos = pos.copy()
    speed_mean = p["speed_mean"]

    if forced_pos is not None:
        return _finish_step(env, p, dt, np.array(forced_pos, dtype=np.float64).reshape(B, 2), prev_pos, vel, rot,
                            prev_mv, hd, dist_trav, dclose, np.zeros(B, dtype=np.int32), np.zeros(B, dtype=bool),
                            z_zero, overwrite=True)

    # -- _stochastic_velocity_update (Agent.py:268-312)
    rot = rot + ou_increment(rot, dt, rotational_velocity_drift,
                             kw.get("rotational_velocity_std", p["rotational_velocity_std"]),
                             kw.get("rotational_velocity_coherence_time", p["rotational_velocity_coherence_time"]),
                             z_rot)
    dtheta = rot * dt
    c, s = np.cos(dtheta), np.sin(dtheta)
    vel = np.stack((c * vel[:, 0] + (-s) * vel[:, 1], s * vel[:, 0] + c * vel[:, 1]), axis=-1)
    speed = np.sqrt(vel[:, 0] * vel[:, 0] + vel[:, 1] * vel[:, 1])
    zero = speed == 0
    vel = np.where(zero[:, None], np.array([1e-8, 0.0]), vel)
    speed = np.where(zero, 1e-8, speed)
    sm_kw = kw.get("speed_mean", speed_mean)
    nv = rayleigh_to_normal(speed, sm_kw)
    nv = nv + ou_increment(nv, dt, 0.0, 1.0, kw.get("speed_coherence_time", p["speed_coherence_time"]), z_speed)
    speed_new = normal_to_rayleigh(nv, sm_kw)
    if p["speed_std"] == 0:
        speed_new = np.full(B, float(sm_kw))
    vel = (speed_new / speed)[:, None] * vel

    # -- _drift_velocity_update (Agent.py:324-341)
    if drift_velocity is not None:
        dv = np.asarray(drift_velocity, dtype=np.float64).reshape(-1, 2)
        tau = p["speed_coherence_time"] / drift_to_random_strength_ratio
        vel = vel + (1 / tau) * (dv - vel) * dt

    # -- _wall_velocity_update (Agent.py:343-415)
    walls = env.walls
    wall_repel_strength = kw.get("wall_repel_strength", p["wall_repel_strength"])
    if wall_repel_strength != 0.0 and len(walls) > 0:
        vw = shortest_vectors_from_walls(pos, walls)  # (B,Nw,2)
        x = np.sqrt(vw[..., 0] ** 2 + vw[..., 1] ** 2)
        dclose = x.min(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            nvec = vw / x[..., None]
        d = kw.get("wall_repel_distance", p["wall_repel_distance"])
        v0 = wall_repel_strength * speed_mean
        k = v0**2 / d**2
        near = x <= d
        with np.errstate(invalid="ignore"):
            acc = np.where(near, k * (d - x), 0.0)
            spd = np.where(near, v0 * (1 - np.sqrt(1 - (d - x) ** 2 / d**2)), 0.0)
        acc_sum = np.zeros((B, 2))
        spd_sum = np.zeros((B, 2))
        for w in range(len(walls)):  # sequential sum over walls, like ndarray.sum(axis=0)
            acc_sum = acc_sum + acc[:, w, None] * nvec[:, w, :]
            spd_sum = spd_sum + spd[:, w, None] * nvec[:, w, :]
        g = kw.get("thigmotaxis", p["thigmotaxis"])
        vel = vel + 3 * ((1 - g) ** 2) * (acc_sum * dt)
        pos = pos + 6 * (g**2) * (spd_sum * dt)

    # -- propose (Agent.py:216)
    pos = pos + vel * dt

    # -- _check_and_handle_wall_collisions (Agent.py:423-441)
    n_bounces = np.zeros(B, dtype=np.int32)
    if len(walls) > 0:
        active = np.ones(B, dtype=bool)
        for _ in range(MAX_BOUNCES):
            idx = np.nonzero(active)[0]
            if len(idx) == 0:
                break
            for i in idx:
                step = np.array([prev_pos[i], pos[i]])
                hits = segments_collide(walls, step[None]).reshape(-1)
                if not hits.any():
                    active[i] = False
                    continue
                w = walls[np.argmax(hits)]  # first colliding wall = lowest index
                vel[i] = wall_bounce(vel[i], w)
                vel[i] = (0.5 * speed_mean / np.sqrt(vel[i, 0] ** 2 + vel[i, 1] ** 2)) * vel[i]
                pos[i] = prev_pos[i] + vel[i] * dt
                n_bounces[i] += 1

    # -- boundary safety net (Agent.py:221-222)
    outside = ~env_is_inside(env, pos)
    pos_before_bc = pos.copy()
    if outside.any():
        pos = np.where(outside[:, None], env_apply_boundary_conditions(env, pos, resample_pos), pos)

    out = _finish_step(env, p, dt, pos, prev_pos, vel, rot, prev_mv, hd, dist_trav, dclose, n_bounces, outside,
                       z_zero, overwrite=False)
    out["pos_before_bc"] = pos_before_bc  # (diagnostic: which branch of apply_boundary_conditions a step took)
    return out


def _finish_step(env, p, dt, pos, prev_pos, vel, rot, prev_mv, hd, dist_trav, dclose, n_bounces, outside, z_zero,
                 overwrite):
    """Common tail of Agent.update (Agent.py:224-242): measured velocities, head direction,
    distance travelled."""
    B = pos.shape[0]
    # -- _measure_velocity_of_step_taken (Agent.py:444-472)
    d_pos = pos - prev_pos
    if env.periodic:
        flip = np.abs(d_pos) > (env.scale / 2)
        d_pos = np.where(flip, -np.sign(d_pos) * (env.scale - np.abs(d_pos)), d_pos)
    mv = d_pos / dt
    mv_norm = np.sqrt(mv[:, 0] ** 2 + mv[:, 1] ** 2)
    still = mv_norm == 0
    if still.any():
        zz = np.zeros((B, 2)) if z_zero is None else np.asarray(z_zero, dtype=np.float64).reshape(B, 2)
        mv = np.where(still[:, None], 1e-8 * zz, mv)
        mv_norm = np.sqrt(mv[:, 0] ** 2 + mv[:, 1] ** 2)
    mrv = pi_domain(get_angle(mv) - get_angle(prev_mv)) / dt
    if overwrite:  # overwrite_velocity=True (Agent.py:461-462, 469-470)
        vel, rot = mv.copy(), mrv.copy()

    # -- _update_head_direction (Agent.py:474-500)
    tau_h = p["head_direction_smoothing_timescale"]
    imm = mv / mv_norm[:, None]
    if tau_h <= dt:
        hd = imm
    else:
        hd = hd * (1 - dt / tau_h) + dt / tau_h * imm
        hd = hd / np.sqrt(hd[:, 0] ** 2 + hd[:, 1] ** 2)[:, None]

    # -- _update_distance_travelled (Agent.py:502-507)
    dist_trav = dist_trav + np.sqrt(d_pos[:, 0] ** 2 + d_pos[:, 1] ** 2)

    return dict(
        pos=pos,
        velocity=vel,
        rotational_velocity=rot,
        measured_velocity=mv,
        measured_rotational_velocity=mrv,
        head_direction=hd,
        distance_travelled=dist_trav,
        distance_to_closest_wall=dclose,
        n_bounces=n_bounces,
        bc_applied=outside,
    )


def wall_bounce(v, wall):
    """utils.wall_bounce (utils.py:304-328) for one velocity `(2,)` and wall `(2,2)`."""
    par = wall[1] - wall[0]
    perp = np.array([-par[1], par[0]])
    if perp[0] * v[0] + perp[1] * v[1] <= 0:
        perp = -perp
    if par[0] * v[0] + par[1] * v[1] <= 0:
        par = -par
    par = par / np.sqrt(par[0] ** 2 + par[1] ** 2)
    perp = perp / np.sqrt(perp[0] ** 2 + perp[1] ** 2)
    return par * (v[0] * par[0] + v[1] * par[1]) - perp * (v[0] * perp[0] + v[1] * perp[1])


# --------------------------------------------------------------------------- #
# Neurons.get_state (Neurons.py)
# --------------------------------------------------------------------------- #
def place_cells(env, pos, centres, widths, description="gaussian", wall_geometry="euclidean",
                min_fr=0.0, max_fr=1.0, widths_scalar=None):
    """PlaceCells.get_state (Neurons.py:936-981) -> `(n, P)`."""
    centres = np.asarray(centres, dtype=np.float64).reshape(-1, 2)
    w = (np.asarray(widths, dtype=np.float64) * np.ones(len(centres)))[:, None]
    dist = env_distances(env, centres, pos, wall_geometry)
    if description == "gaussian":
        fr = np.exp(-(dist**2) / (2 * (w**2)))
    elif description == "gaussian_threshold":
        fr = np.maximum(np.exp(-(dist**2) / (2 * (w**2))) - np.exp(-1 / 2), 0) / (1 - np.exp(-1 / 2))
    elif description == "diff_of_gaussians":
        ratio = 1.5
        fr = np.exp(-(dist**2) / (2 * (w**2))) - (1 / ratio**2) * np.exp(-(dist**2) / (2 * ((ratio * w) ** 2)))
        fr = fr * (ratio**2 / (ratio**2 - 1))
    elif description == "one_hot":
        closest = np.argmin(np.abs(dist), axis=0)
        fr = np.eye(len(centres))[closest].T
    elif description == "top_hat":
        ws = w[0, 0] if widths_scalar is None else widths_scalar  # Neurons.py:976 uses the scalar `widths`
        fr = 1.0 * (dist < ws)
    else:
        raise ValueError(description)
    return fr * (max_fr - min_fr) + min_fr


def grid_cell_w(orientations):
    """GridCells.__init__ wave-vector construction (Neurons.py:1154-1161) -> `(n,3,2)`."""
    def rot(v, th):
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        return R @ v
    w = []
    for th in np.asarray(orientations, dtype=np.float64):
        w1 = rot(np.array([1, 0]), th)
        w.append(np.array([w1, rot(w1, np.pi / 3), rot(w1, 2 * np.pi / 3)]))
    return np.array(w)


def grid_cells(pos, gridscales, phase_offsets, w, description="rectified_cosines",
               width_ratio=4 / (3 * np.sqrt(3)), min_fr=0.0, max_fr=1.0):
    """GridCells.get_state, 2D (Neurons.py:1172-1236) -> `(n, P)`."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    gs = np.asarray(gridscales, dtype=np.float64)
    origin = gs.reshape(-1, 1) * np.asarray(phase_offsets, dtype=np.float64) / (2 * np.pi)
    vecs = origin[:, None, :] - pos[None, :, :]  # utils.get_vectors_between(origin, pos)
    k = ((2 * np.pi) / gs)[:, None]
    phi = [k * (vecs[..., 0] * w[:, i, 0][:, None] + vecs[..., 1] * w[:, i, 1][:, None]) for i in range(3)]
    if description == "rectified_cosines":
        fr = (1 / 3) * (np.cos(phi[0]) + np.cos(phi[1]) + np.cos(phi[2]))
        f0 = (1 / 3) * (2 * np.cos(np.sqrt(3) * np.pi * width_ratio / 2) + 1)
        fr = (fr - f0) / (1 - f0)
        fr = np.where(fr < 0, 0.0, fr)
    elif description == "shifted_cosines":
        fr = (2 / 3) * ((1 / 3) * (np.cos(phi[0]) + np.cos(phi[1]) + np.cos(phi[2])) + (1 / 2))
    else:
        raise ValueError(description)
    return fr * (max_fr - min_fr) + min_fr


def bvc_test_angles(dtheta=2):
    """BoundaryVectorCells.__init__ (Neurons.py:1584-1596): K = int(360/dtheta)
    angles `[0] + [2*pi*i*dtheta/360 for i in range(K-1)]` (0 deg duplicated, the
    last angle missing — SURVEY App. C-2) and unit directions `R(angle)(1,0)`."""
    K = int(360 / dtheta)
    angles = [0.0] + [2 * np.pi * i * dtheta / 360 for i in range(K - 1)]
    angles = np.array(angles)
    dirs = np.stack((np.cos(angles) * 1 + (-np.sin(angles)) * 0, np.sin(angles) * 1 + np.cos(angles) * 0), axis=-1)
    dirs[0] = np.array([1.0, 0.0])
    return angles, dirs


def bvc_fr_norm(test_angles, sigma_angles):
    """`cell_fr_norm` (Neurons.py:1598-1604): sum over test angles of von_mises(theta; 0, sigma)."""
    return von_mises(test_angles.reshape(1, -1), 0.0, np.asarray(sigma_angles).reshape(-1, 1)).sum(axis=1)


def bvc_ray_distances(pos, walls, test_dirs):
    """Ray stage of BoundaryVectorCells.get_state (Neurons.py:1655-1684, 1746-1778):
    for each position and test direction the line parameter (= distance, rays are
    unit length) to the first wall hit.  -> `(P, K)`."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    P, K = len(pos), len(test_dirs)
    segs = np.empty((P, K, 2, 2))
    segs[:, :, 0, :] = pos[:, None, :]
    segs[:, :, 1, :] = pos[:, None, :] + test_dirs[None, :, :]
    l_a, l_b = segment_intercepts(segs.reshape(-1, 2, 2), walls)
    l_a = l_a.reshape(P, K, -1)
    l_b = l_b.reshape(P, K, -1)
    # boundary_vector_preference_function: np.piecewise, later conditions overwrite earlier
    pref = np.zeros_like(l_a)
    with np.errstate(divide="ignore", invalid="ignore"):
        pref = np.where(l_a > 0, 1 / l_a, pref)
    pref = np.where(l_a < 0, -1.0, pref)
    pref = np.where(l_b < 0, -1.0, pref)
    pref = np.where(l_b > 1, -1.0, pref)
    first = np.argmax(pref, axis=-1)[..., None]
    return np.take_along_axis(l_a, first, axis=-1)[..., 0]


def bvc(pos, walls, tuning_distances, tuning_angles, sigma_distances, sigma_angles, dtheta=2,
        head_direction=None, min_fr=0.0, max_fr=1.0):
    """BoundaryVectorCells.get_state (Neurons.py:1617-1744) -> `(n, P)`.
    `head_direction` `(P,2)` switches to the egocentric frame (test angles are
    shifted by the head bearing, the rays themselves stay allocentric)."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    angles, dirs = bvc_test_angles(dtheta)
    d = bvc_ray_distances(pos, walls, dirs)  # (P,K)
    th = np.broadcast_to(angles[None, :], d.shape)
    if head_direction is not None:
        hdir = np.asarray(head_direction, dtype=np.float64).reshape(-1, 2)
        th = th - get_angle(hdir)[:, None]
    mu_d = np.asarray(tuning_distances, dtype=np.float64)[:, None, None]
    sg_d = np.asarray(sigma_distances, dtype=np.float64)[:, None, None]
    mu_a = np.asarray(tuning_angles, dtype=np.float64)[:, None, None]
    sg_a = np.asarray(sigma_angles, dtype=np.float64)[:, None, None]
    out = np.empty((mu_d.shape[0], len(pos)))
    norm = bvc_fr_norm(angles, np.asarray(sigma_angles, dtype=np.float64))
    chunk = max(1, int(2_000_000 // max(1, d.shape[1] * mu_d.shape[0])))
    for s in range(0, len(pos), chunk):
        dd = d[None, s:s + chunk, :]
        tt = th[None, s:s + chunk, :]
        g = gaussian(dd, mu_d, sg_d) * von_mises(tt, mu_a, sg_a)
        out[:, s:s + chunk] = g.sum(axis=-1)
    out = out / norm[:, None]
    return out * (max_fr - min_fr) + min_fr


def head_direction_cells(head_direction, n, angular_spread_degrees=45.0, min_fr=0.0, max_fr=1.0):
    """HeadDirectionCells.get_state, 2D (Neurons.py:2403-2409, 2466-2483) -> `(n, P)`."""
    hd = np.asarray(head_direction, dtype=np.float64).reshape(-1, 2)
    pref = np.linspace(0, 2 * np.pi, n + 1)[:-1]
    sig = angular_spread_degrees * np.pi / 180
    fr = von_mises(get_angle(hd)[None, :], pref[:, None], sig)
    return fr * (max_fr - min_fr) + min_fr


def velocity_cells(velocity, n, one_sigma_speed, angular_spread_degrees=45.0, min_fr=0.0, max_fr=1.0,
                   scale_velocity=None):
    """VelocityCells.get_state (Neurons.py:2577-2583): HeadDirectionCells tuned to `velocity / |velocity|`
    (Neurons.py:2446-2461), scaled to [min_fr, max_fr], THEN multiplied by `|v| / one_sigma_speed` where v
    is the agent's velocity (`scale_velocity`; the reference uses Agent.velocity for the scale even when a
    `velocity=` kwarg gives the direction, :2581) -> `(n, P)`."""
    v = np.asarray(velocity, dtype=np.float64).reshape(-1, 2)
    fr = head_direction_cells(v / np.linalg.norm(v, axis=-1, keepdims=True), n, angular_spread_degrees, min_fr, max_fr)
    sv = v if scale_velocity is None else np.asarray(scale_velocity, dtype=np.float64).reshape(-1, 2)
    return fr * (np.linalg.norm(sv, axis=-1) / one_sigma_speed)[None, :]


def speed_cell(vel, one_sigma_speed, min_fr=0.0, max_fr=1.0):
    """SpeedCell.get_state (Neurons.py:2632-2651) -> `(1, P)`."""
    v = np.asarray(vel, dtype=np.float64).reshape(-1, 2)
    return (np.linalg.norm(v, axis=-1) / one_sigma_speed * (max_fr - min_fr) + min_fr)[None, :]


def random_spatial_neurons(env, pos, X, targets, lengthscale, wall_geometry="euclidean"):
    """RandomSpatialNeurons.get_state (Neurons.py:2916-2942) with kernel() (:2944-2956): the
    kernel-weighted local average of the anchor targets `(M, n)` -> `(n, P)`."""
    d = env_distances(env, pos, X, wall_geometry)
    k = np.exp(-(d ** 2) / (2 * lengthscale ** 2))
    with np.errstate(invalid="ignore", divide="ignore"):
        k = k / np.sum(k, axis=1, keepdims=True)
        return (k @ np.asarray(targets, dtype=np.float64)).T


def object_vector_cells(env, pos, objects, object_types, tuning_distances, tuning_angles, sigma_distances,
                        sigma_angles, tuning_types, walls_occlude=True, head_direction=None, min_fr=0.0, max_fr=1.0):
    """ObjectVectorCells.get_state (Neurons.py:1991-2116) -> `(n, P)`.  `head_direction (P,2)`
    switches to the egocentric frame."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    objects = np.asarray(objects, dtype=np.float64).reshape(-1, 2)
    geom = "line_of_sight" if walls_occlude else "euclidean"
    dist = env_distances(env, pos, objects, geom)           # (P, M)
    vec = -1 * env_vectors_between(env, pos, objects)        # object - position, (P, M, 2)
    bearing = get_angle(vec.reshape(-1, 2)).reshape(dist.shape)
    if head_direction is not None:
        bearing = bearing - get_angle(np.asarray(head_direction, dtype=np.float64).reshape(-1, 2))[:, None]
    mu_d = np.asarray(tuning_distances, dtype=np.float64)[None, None, :]
    sg_d = np.asarray(sigma_distances, dtype=np.float64)[None, None, :]
    mu_a = np.asarray(tuning_angles, dtype=np.float64)[None, None, :]
    sg_a = np.asarray(sigma_angles, dtype=np.float64)[None, None, :]
    fr = gaussian(dist[:, :, None], mu_d, sg_d) * von_mises(bearing[:, :, None], mu_a, sg_a)  # (P, M, n)
    mask = (np.asarray(object_types)[:, None] == np.asarray(tuning_types)[None, :]).astype(int)[None]
    fr = (fr * mask).sum(axis=1).T
    return fr * (max_fr - min_fr) + min_fr


def agent_vector_cells(env, pos, other_pos, tuning_distances, tuning_angles, sigma_distances, sigma_angles,
                       walls_occlude=True, head_direction=None, min_fr=0.0, max_fr=1.0):
    """AgentVectorCells.get_state (Neurons.py:2204-2320) -> `(n, P)`: ObjectVectorCells with one object,
    the other agent, whose position `other_pos (P,2)` is given per observer position (the reference
    has one observer and one other agent per call)."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
    other = np.broadcast_to(np.asarray(other_pos, dtype=np.float64).reshape(-1, 2), pos.shape)
    n = len(np.asarray(tuning_distances))
    out = np.empty((n, len(pos)))
    for i in range(len(pos)):
        hd = None if head_direction is None else np.asarray(head_direction, dtype=np.float64).reshape(-1, 2)[i:i + 1]
        out[:, i] = object_vector_cells(env, pos[i:i + 1], other[i:i + 1], [0], tuning_distances, tuning_angles,
                                        sigma_distances, sigma_angles, np.zeros(n, dtype=int), walls_occlude, hd,
                                        min_fr, max_fr)[:, 0]
    return out


def activate(x, spec):
    """utils.activate (utils.py:919-1026) for the named activations -> (f(x), df/dx)."""
    name = spec.get("activation", "sigmoid")
    x = np.asarray(x, dtype=np.float64)
    if name == "linear":
        return x, np.ones(x.shape)
    if name == "sigmoid":
        d = {"max_fr": 1, "min_fr": 0, "mid_x": 1, "width_x": 2}
        d.update(spec)
        beta = np.log((1 - 0.05) / 0.05) / (0.5 * d["width_x"])
        f = ((d["max_fr"] - d["min_fr"]) / (1 + np.exp(-beta * (x - d["mid_x"])))) + d["min_fr"]
        return f, beta * (f - d["min_fr"]) * (1 - (f - d["min_fr"]) / (d["max_fr"] - d["min_fr"]))
    d = {"gain": 1, "threshold": 0}
    d.update(spec)
    g, th = d["gain"], d["threshold"]
    if name == "relu":
        return g * np.maximum(0, x - th), g * ((x - th) > 0)
    if name == "tanh":  # the reference's derivative ignores the threshold (utils.py:998)
        return g * np.tanh(x - th), g * (1 - np.tanh(x) ** 2)
    if name == "retanh":
        return g * np.maximum(0, np.tanh(x - th)), g * (1 - np.tanh(x) ** 2) * ((x - th) > 0)
    if name == "softmax":
        return g * np.log(1 + np.exp(x - th)), g / (1 + np.exp(-(x - th)))
    raise ValueError(name)


def feedforward(inputs, weights, biases, spec):
    """FeedForwardLayer.get_state (Neurons.py:2797-2847): inputs list of `(n_in, P)` rates,
    weights list of `(n, n_in)` -> (rates `(n, P)`, activation derivative `(n, P)`)."""
    V = np.zeros((weights[0].shape[0], np.asarray(inputs[0]).shape[1]))
    for w, I in zip(weights, inputs):
        V = V + np.matmul(np.asarray(w, dtype=np.float64), np.asarray(I, dtype=np.float64))
    V = V + np.asarray(biases, dtype=np.float64).reshape(-1, 1)
    return activate(V, spec)


# --------------------------------------------------------------------------- #
# Neurons.update noise + spikes (Neurons.py:145-171, 681-687)
# --------------------------------------------------------------------------- #
def spikes_ref(rates, u, dt):
    """Neurons.save_to_history (Neurons.py:682-684), float64: `u < dt * firingrate`."""
    return np.asarray(u) < (dt * np.asarray(rates))


def spikes_f32(rates32, u32, dt):
    """The product's exactly-specified spike rule: one fp32 multiply, one fp32
    compare.  Bit-exact target for the HIP epilogue."""
    r = np.asarray(rates32, dtype=np.float32)
    return np.asarray(u32, dtype=np.float32) < (np.float32(dt) * r)


# --------------------------------------------------------------------------- #
# Counter-based RNG used by the product in production mode (new functionality:
# the reference uses the global MT19937 stream).  Philox4x32-10,
# Salmon et al. 2011; integer part bit-exact, restated here so tests can
# regenerate the device's draws on the host.
# --------------------------------------------------------------------------- #
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
TAG_MOTION = 0x4D4F5449
TAG_SPIKES = 0x53504B00


SPIKE_ROUNDS = 7   # the spike streams' Philox4x32-7 (the motion / noise / task streams: -10); csrc/riab_device.h


def philox4x32_10(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32-`rounds` (Salmon et al., SC'11; 7 rounds is the smallest count the authors found to pass
    BigCrush, 10 their default).  Counters are uint32 arrays (broadcastable),
    key two python ints.  Returns four uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0 = PHILOX_M0 * c[0]
        p1 = PHILOX_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return tuple(x.astype(np.uint32) for x in c)


def motion_normals(seed, step, agent_ids):
    """The device's per-(step, agent) motion draws: Box-Muller on the top 24 bits of two
    Philox words.  The
    device evaluates log2 / sin / cos with the fp32 hardware approximations, so this
    float64 restatement agrees to ~1e-6 (the Philox words themselves are bit-exact);
    trajectories are compared through the `z_out` record of the kernel.
    Returns z_rot, z_speed, z_zero0, z_zero1 (float64)."""
    agent_ids = np.asarray(agent_ids, dtype=np.uint64)
    pair = step >> 1  # one Philox call serves two steps: words (0,1) on even steps, (2,3) on odd ones
    x = philox4x32_10(pair & 0xFFFFFFFF, (pair >> 32) & 0xFFFFFFFF, agent_ids, TAG_MOTION,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    y = philox4x32_10(step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF, agent_ids, TAG_MOTION ^ 1,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)  # the zero-displacement branch's stream
    def bm(a, b):
        u1 = ((a >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0**-24
        u2 = (b >> np.uint32(8)).astype(np.float64) * 2.0**-24
        r = np.sqrt(-2.0 * np.log(u1))
        return r * np.cos(TWO_PI * u2), r * np.sin(TWO_PI * u2)
    z0, z1 = bm(x[2], x[3]) if (step & 1) else bm(x[0], x[1])
    z2, z3 = bm(y[0], y[1])
    return z0, z1, z2, z3


def spike_uniforms(seed, step, pop_id, n_cells, n_agents, agent_id0=0):
    """The device's per-(step, cell, agent) fp32 uniforms in [0,1): one Philox4x32-7 call
    per (cell, group of 4 consecutive global agent ids); word j -> agent 4g+j;
    `u = (word >> 8) * 2^-24`.  -> `(n_cells, n_agents)` float32.
    (Round 6: seven rounds instead of ten — the spike epilogue is what makes a store-bound rate kernel VALU-bound, and
    the generator is two thirds of it.  The rule is this build's own specification — the reference draws from NumPy's
    global MT19937 stream, which no batched kernel reproduces; the reference-pinned spike test feeds explicit uniforms.)"""
    assert agent_id0 % 4 == 0 and n_agents % 4 == 0
    g = (np.arange(n_agents // 4, dtype=np.uint64) + np.uint64(agent_id0 // 4))[None, :]
    cell = np.arange(n_cells, dtype=np.uint64)[:, None]
    xs = philox4x32_10(step & 0xFFFFFFFF, cell, g, TAG_SPIKES | (pop_id & 0xFF),
                       seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, rounds=SPIKE_ROUNDS)
    u = np.stack([(x >> np.uint32(8)).astype(np.float32) * np.float32(2.0**-24) for x in xs], axis=-1)
    return u.reshape(n_cells, n_agents)


# --------------------------------------------------------------------------------------------------
# TaskEnvironment bookkeeping (reference contribs/TaskEnvironment.py), one single-agent replica
# --------------------------------------------------------------------------------------------------
DECAY_CONSTANT, DECAY_LINEAR, DECAY_EXPONENTIAL, DECAY_NONE = 0, 1, 2, 3
GOAL_TIME_ELAPSED = -2
# no_reward_default (TaskEnvironment.py:949-951): (init_state, dt, expire_clock, preset, knob)
PAD_REWARD = (0.0, 0.01, 0.1, DECAY_NONE, 0.0)


def reward_delta(preset, knob, state):
    """Reward.get_delta without external drive (TaskEnvironment.py:823-832) for the decay presets
    of :732-737 (`partial(preset, *knobs)`: the knob is the first argument, the state the second)."""
    if preset == DECAY_CONSTANT:
        return -knob
    if preset == DECAY_LINEAR:
        return -(knob * state)
    if preset == DECAY_EXPONENTIAL:
        return -(knob * np.exp(state))
    return -0.0


class TaskLane:
    """One agent's view of a TaskEnvironment in which it is the only agent: goal list, reward
    cache and episode counters, stepped exactly as TaskEnvironment.step does after Agent.update
    (TaskEnvironment.py:410-449).  `goals`: rows (x, y, radius, reward init_state, reward dt,
    reward expire_clock, decay preset, decay knob)."""

    def __init__(self, env, goals, goalorder="nonsequential", terminate_delay=0.0, default_reward_level=0.0):
        self.env = env
        self.goals = np.asarray(goals, dtype=np.float64).reshape(-1, 8)
        self.sequential = goalorder == "sequential"
        self.terminate_delay = float(terminate_delay)
        self.default_level = float(default_reward_level)
        self.goal_list = []          # pool indices (GOAL_TIME_ELAPSED = the termination-delay goal)
        self.pad_start = 0.0
        self.delayed = False
        self.rewards = []            # [state, expire_clock, source goal], in append order
        self.episode = 0
        self.ep_start = 0.0
        self.started = False
        self.any_ended = False
        self.finished = []           # (episode, start, end, duration)
        self.late_completions = 0

    def _reward_template(self, src):
        return PAD_REWARD if src == GOAL_TIME_ELAPSED else tuple(self.goals[src, 3:8])

    def _met(self, src, pos, t_env):
        if src == GOAL_TIME_ELAPSED:  # TimeElapsedGoal.check (:1271-1278)
            return t_env - self.pad_start >= self.terminate_delay
        g = self.goals[src]           # SpatialGoal._in_goal_radius (:1319-1332)
        d = env_distances(self.env, np.asarray(pos, float).reshape(1, 2), g[None, 0:2], "line_of_sight")
        return bool((d < g[2]).all())

    def _award(self, src):            # RewardCache.append (:902-911): a copy of the goal's reward
        tpl = self._reward_template(src)
        self.rewards.append([float(tpl[0]), float(tpl[2]), src])

    def _check_pass(self, pos, t_env):
        """GoalCache.check(remove_finished=True) (:1076-1152) for the lane; returns goals consumed."""
        done = 0
        if not self.goal_list:
            return 0
        if self.sequential:           # `this` is always the head: pop() rewinds the marker (:1161-1163)
            if self._met(self.goal_list[0], pos, t_env):
                self._award(self.goal_list[0])
                del self.goal_list[0]
                done = 1
            return done
        g = 0
        while g < len(self.goal_list):
            if self._met(self.goal_list[g], pos, t_env):
                self._award(self.goal_list[g])
                del self.goal_list[g]
                done += 1
            g += 1                    # also after a pop (:1141): the goal that slid into slot g waits
        return done

    def _rewards_update(self):
        """RewardCache.update (:913-927) with the remove-while-iterating skip."""
        i = 0
        while i < len(self.rewards):
            state, expire, src = self.rewards[i]
            tpl = self._reward_template(src)
            rdt = float(tpl[1])
            state = state + reward_delta(int(tpl[3]), float(tpl[4]), state) * rdt
            expire = expire - rdt
            if expire <= 0:
                del self.rewards[i]
            else:
                self.rewards[i] = [state, expire, src]
            i += 1

    def step(self, pos, t_env):
        """-> (reward total, terminal).  `t_env` is the clock after `self.t += self.dt`."""
        self._rewards_update()
        self._check_pass(pos, t_env)
        terminal = len(self.goal_list) == 0
        if terminal and self.terminate_delay and not self.delayed:  # :421-434
            self.delayed = True
            self.pad_start = t_env
            self.goal_list.append(GOAL_TIME_ELAPSED)
            self._check_pass(pos, t_env)
            terminal = len(self.goal_list) == 0
        late = self._check_pass(pos, t_env)                            # :438
        terminal_last = len(self.goal_list) == 0
        if late and terminal_last and not terminal:
            self.late_completions += 1
        total = 0.0
        for r in self.rewards:        # python sum(), left to right from 0 (:933)
            total = total + r[0]
        total = total + self.default_level
        return total, terminal_last

    def reset(self, t_env, selected):
        """TaskEnvironment.reset (:307-351) with the goal selection given (`selected` pool indices)."""
        zero = False
        if self.started:
            duration = t_env - self.ep_start
            zero = duration == 0
            if not zero:
                self.any_ended = True
                self.finished.append((self.episode, self.ep_start, t_env, duration))
        if not zero:
            self.episode += 1
        self.started = True
        self.ep_start = t_env if self.any_ended else 0.0
        self.goal_list = [int(s) for s in selected]
        self.delayed = False


def world_check_pass(goal_list, met, sequential):
    """One GoalCache.check(remove_finished=True) (:1076-1152) over SEVERAL agents with agentmode="interact"
    (GoalCache.pop :1165-1172: a goal is popped from every agent's list, so all lists stay equal — one shared list).
    `goal_list`: the shared list (modified in place); `met(agent, entry) -> bool`: Goal.check for that agent.
    Agents take their turns in `agent_names` order against the list AS THE EARLIER AGENTS OF THE PASS LEFT IT.
    Returns [(agent, entry), ...] in award order."""
    awards = []
    n_agents = met.n_agents
    for a in range(n_agents):
        if len(goal_list) == 0:           # :1102 / :1126
            continue
        if sequential:                    # `this` is the head for everybody: pop() rewinds every agent's marker (:1167-1172)
            if met(a, goal_list[0]):
                awards.append((a, goal_list[0]))
                del goal_list[0]
            continue
        g = 0
        while g < len(goal_list):
            if met(a, goal_list[g]):
                awards.append((a, goal_list[g]))
                del goal_list[g]
            g += 1                        # also after a pop (:1141): the goal that slid into slot g is left to the later agents
    return awards


class _Met:
    def __init__(self, n_agents, fn):
        self.n_agents = n_agents
        self.fn = fn

    def __call__(self, a, entry):
        return self.fn(a, entry)


class TaskWorld:
    """A TaskEnvironment with `n_agents` agents in ONE world, agentmode="interact" (the reference's default, :1030):
    one clock, one episode, one shared goal list; every agent keeps its own reward cache.  Stepped exactly as
    TaskEnvironment.step does after the agents' updates (:410-449).  `goals`: rows as for TaskLane."""

    def __init__(self, env, goals, n_agents, goalorder="nonsequential", terminate_delay=0.0, default_reward_level=0.0):
        self.lanes = [TaskLane(env, goals, goalorder, terminate_delay, default_reward_level) for _ in range(n_agents)]
        self.n_agents = n_agents
        self.sequential = goalorder == "sequential"
        self.terminate_delay = float(terminate_delay)
        self.goal_list = []
        self.pad_start = 0.0
        self.delayed = False
        self.episode = 0
        self.ep_start = 0.0
        self.started = False
        self.any_ended = False
        self.finished = []
        self.late_completions = 0

    def _pass(self, pos, t_env):
        def met(a, entry):
            if entry == GOAL_TIME_ELAPSED:    # TimeElapsedGoal.check (:1271-1278): whoever looks
                return t_env - self.pad_start >= self.terminate_delay
            return bool(self._inside[a, entry])
        awards = world_check_pass(self.goal_list, _Met(self.n_agents, met), self.sequential)
        for a, entry in awards:           # _is_terminal_state (:283-285): appended in award order
            self.lanes[a]._award(entry)
        return len(awards)

    def step(self, pos, t_env):
        """-> (reward totals [n_agents], terminal).  `pos` [n_agents, 2] after the agents' updates."""
        pos = np.asarray(pos, float).reshape(self.n_agents, 2)
        # SpatialGoal._in_goal_radius (:1319-1332) of every agent for every goal of the pool, once per step (the passes ask
        # about the same goals at the same positions)
        G = self.lanes[0].goals
        self._inside = env_distances(self.lanes[0].env, pos, G[:, 0:2], "line_of_sight") < G[None, :, 2]
        for L in self.lanes:              # RewardCache.update of every agent (:406-408)
            L._rewards_update()
        self._pass(pos, t_env)
        terminal = len(self.goal_list) == 0
        if terminal and self.terminate_delay and not self.delayed:   # :421-434: the pad goal joins every agent's list
            self.delayed = True
            self.pad_start = t_env
            self.goal_list.append(GOAL_TIME_ELAPSED)
            self._pass(pos, t_env)
            terminal = len(self.goal_list) == 0
        late = self._pass(pos, t_env)     # :438
        terminal_last = len(self.goal_list) == 0
        if late and terminal_last and not terminal:
            self.late_completions += 1
        totals = []
        for L in self.lanes:
            total = 0.0
            for r in L.rewards:
                total = total + r[0]
            totals.append(total + L.default_level)
        return np.array(totals), terminal_last

    def reset(self, t_env, selected):
        """TaskEnvironment.reset (:307-351): one episode table, one goal selection for everybody."""
        zero = False
        if self.started:
            duration = t_env - self.ep_start
            zero = duration == 0
            if not zero:
                self.any_ended = True
                self.finished.append((self.episode, self.ep_start, t_env, duration))
        if not zero:
            self.episode += 1
        self.started = True
        self.ep_start = t_env if self.any_ended else 0.0
        self.goal_list = [int(s) for s in selected]
        self.delayed = False


def task_reset_draws(seed, counter, lane_id, n_pool, n_select):
    """The production goal selection of riab_task_reset: uniform sample without replacement driven
    by Philox — draw i picks the j-th goal still in the pool (ascending order),
    j = floor(w_i * (n_pool - i) / 2^32), w_i = word i%4 of block 1 + i//4."""
    TAG = 0x5441534B
    remaining = list(range(n_pool))
    out = []
    words = None
    for i in range(min(n_select, n_pool)):
        if i % 4 == 0:
            words = philox4x32_10(counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, lane_id & 0xFFFFFFFF,
                                  (TAG + 1 + i // 4) & 0xFFFFFFFF, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        w = int(words[i % 4])
        j = (w * (n_pool - i)) >> 32
        out.append(remaining.pop(j))
    return out


def task_teleport_draw(seed, counter, lane_id, env):
    """Teleport position of riab_task_reset (block 0 of the same Philox stream)."""
    TAG = 0x5441534B
    w = philox4x32_10(counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, lane_id & 0xFFFFFFFF, TAG,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    e = env.extent
    cx, cy = 0.5 * (e[0] + e[1]), 0.5 * (e[2] + e[3])
    half = 0.45 * np.sqrt((e[1] - e[0]) * (e[3] - e[2]))
    ux, uy = (int(w[0]) + 0.5) * 2.0 ** -32, (int(w[1]) + 0.5) * 2.0 ** -32
    return np.array([cx + (2 * ux - 1) * half, cy + (2 * uy - 1) * half])

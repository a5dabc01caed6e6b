"""Host-side helpers of the batched hot path: the `default_params` protocol,
init-time parameter samplers and a few scalar geometry helpers used while
building device tables.  Nothing here runs per step.

Reference interfaces mirrored (paths relative to the reference checkout):
`utils.update_class_params / collect_all_params / check_params` (utils.py:804-916),
`utils.distribution_sampler` (utils.py:460-538), `utils.create_random_assembly`
(utils.py:1124-1220), `utils.rotate / get_angle / get_rayleigh_*` (utils.py:231-301, 395-406).
Samplers draw from the global `np.random` state in the same order as the
reference so that a seeded script builds identical cell tables (tests/golden:
update_init.npz)."""
import inspect
import warnings

import numpy as np


# --------------------------------------------------------------------------- #
# default_params protocol
# --------------------------------------------------------------------------- #
def collect_all_params(obj_class, keys_only=False, dict_name="default_params"):
    """Merge the `default_params` class attributes from the root of the class
    hierarchy down to `obj_class` (children override parents)."""
    if not inspect.isclass(obj_class):
        raise ValueError("obj_class must be a class object.")
    chain = [c for c in reversed(obj_class.__mro__) if dict_name in c.__dict__]
    if dict_name not in obj_class.__dict__:
        warnings.warn(f"{obj_class.__name__} has no class attribute '{dict_name}'; nothing to collect.")
        return [] if keys_only else {}
    merged = {}
    for c in chain:
        merged.update(getattr(c, dict_name))
    return sorted(merged.keys()) if keys_only else merged


def update_class_params(obj, params, get_all_defaults=False):
    """Set every (key, value) of `params` as an attribute of `obj`; with
    `get_all_defaults` the inherited defaults are filled in first."""
    if get_all_defaults:
        merged = collect_all_params(obj.__class__)
        merged.update(params)
        params = merged
    for k, v in params.items():
        setattr(obj, k, v)


def check_params(obj, param_keys):
    """Warn about keys that no class in the hierarchy declares in `default_params`."""
    if inspect.isclass(obj):
        raise ValueError("Obj must be an instance of a class, not a class object.")
    cls = obj.__class__
    if "default_params" not in cls.__dict__:
        warnings.warn(f"{cls} has no 'default_params'; cannot check parameter keys.")
        return
    known = collect_all_params(cls, keys_only=True)
    unexpected = [k for k in param_keys if k not in known]
    if unexpected:
        names = ", ".join(f"'{k}'" for k in unexpected)
        warnings.warn(
            f"Found {len(unexpected)} unexpected params key(s) while initializing {cls.__name__} object: {names}.\n"
            f"If you intended to set this parameter, ignore this message. To see all default parameters for this "
            f"class call {cls.__name__}.get_all_default_params().")
    return unexpected


# --------------------------------------------------------------------------- #
# scalar geometry used at init time
# --------------------------------------------------------------------------- #
def rotate(vector, theta):
    """Rotate a 2-vector anticlockwise by `theta` radians."""
    c, s = np.cos(theta), np.sin(theta)
    return np.matmul(np.array([[c, -s], [s, c]]), vector)


def get_angle(vec, is_array=False):
    """Angle of direction vectors `(..., 2)`, anticlockwise from the x-axis, in
    [0, 2pi): `mod(arctan2(y, x + 1e-6), 2pi)` (the 1e-6 is the reference's).
    A segment `(2, 2)` = [start, end] counts as its direction; `is_array=True` says the
    first axis is a list (of vectors `(N, 2)` or segments `(N, 2, 2)`), which only matters
    for telling two vectors from one segment (reference utils.get_angle, utils.py:231-273)."""
    vec = np.asarray(vec, dtype=float)
    one = vec[0] if is_array else vec
    if one.shape == (2, 2):
        vec = vec[..., 1, :] - vec[..., 0, :]
    return np.mod(np.arctan2(vec[..., 1], vec[..., 0] + 1e-6), 2 * np.pi)


def get_rayleigh_sigma(mean):
    return mean / np.sqrt(np.pi / 2)


def get_rayleigh_mean(sigma):
    return sigma * np.sqrt(np.pi / 2)


# --------------------------------------------------------------------------- #
# init-time samplers
# --------------------------------------------------------------------------- #
def distribution_sampler(distribution_name="uniform", distribution_parameters=(1,), shape=(10,)):
    """Sample an array of `shape` from a named distribution:
    uniform (low, high)|(p -> 0.5p..1.5p), rayleigh (scale), normal (loc, scale),
    logarithmic (low, high), delta (value), modules (v1, v2, ...), truncnorm
    (low, high, loc, scale)."""
    if isinstance(distribution_parameters, list):
        distribution_parameters = tuple(distribution_parameters)
    elif not isinstance(distribution_parameters, tuple):
        distribution_parameters = (distribution_parameters,)
    p = distribution_parameters
    if distribution_name == "uniform":
        low, high = (0.5 * p[0], 1.5 * p[0]) if len(p) == 1 else (p[0], p[1])
        return np.random.uniform(low, high, size=shape)
    if distribution_name == "rayleigh":
        return np.random.rayleigh(scale=p[0], size=shape)
    if distribution_name == "normal":
        return np.random.normal(loc=p[0], scale=p[1], size=shape)
    if distribution_name == "logarithmic":
        assert len(shape) == 1, "Logarithmic distribution only works for 1D arrays"
        return np.logspace(np.log10(p[0]), np.log10(p[1]), num=shape[0], base=10)
    if distribution_name == "delta":
        return p[0] * np.ones(shape)
    if distribution_name == "modules":
        assert len(shape) == 1, "Modules distribution only works for 1D arrays"
        per = shape[0] // len(p)
        out = p[-1] * np.ones(shape)  # the remainder joins the last module
        for i, val in enumerate(p):
            out[i * per:(i + 1) * per] = val
        return out
    if distribution_name == "truncnorm":
        from scipy import stats
        lower, upper, mu, sigma = p[0], p[1], p[2], p[3]
        return stats.truncnorm.rvs((lower - mu) / sigma, (upper - mu) / sigma, scale=sigma, loc=mu, size=shape)
    raise ValueError("This distribution is not recognised")


def create_random_assembly(tuning_distance_distribution="uniform", tuning_distance=(0.02, 0.3),
                           tuning_angle_distribution="uniform", tuning_angle=(0.0, 360.0),
                           sigma_angle_distribution="uniform", sigma_angle=(10, 30),
                           sigma_distance_distribution="diverging", sigma_distance=(0.08, 12), n=10, **kwargs):
    """Tuning parameters of a random population of vector cells: returns
    (tuning_distance, tuning_angle [rad], sigma_distance, sigma_angle [rad]).
    Lists/arrays are taken verbatim (and fix n); tuples parameterise a
    distribution.  "diverging" sigma_distance = xi + tuning_distance/beta."""
    given = [p for p in (tuning_distance, tuning_angle, sigma_distance, sigma_angle)
             if isinstance(p, (list, np.ndarray))]
    if len(given) == 1:
        n = len(given[0])
    elif len(given) > 1:
        lengths = {len(p) for p in given}
        assert len(lengths) == 1, "If more than one parameter is passed as a list, they must all have the same length"
        n = lengths.pop()

    def draw(value, dist):
        if isinstance(value, (list, np.ndarray)):
            return np.array(value)
        if isinstance(value, tuple):
            return distribution_sampler(dist, value, shape=(n,))
        return value

    tuning_distance = np.abs(draw(tuning_distance, tuning_distance_distribution))
    if isinstance(sigma_distance, tuple) and sigma_distance_distribution == "diverging":
        sigma_distance = sigma_distance[0] + tuning_distance / sigma_distance[1]
    else:
        sigma_distance = draw(sigma_distance, sigma_distance_distribution)
    tuning_angle = draw(tuning_angle, tuning_angle_distribution)
    sigma_angle = draw(sigma_angle, sigma_angle_distribution)
    tuning_angle = np.asarray(tuning_angle, dtype=float) * (np.pi / 180)
    sigma_angle = np.asarray(sigma_angle, dtype=float) * (np.pi / 180)
    return tuning_distance, tuning_angle, sigma_distance, sigma_angle


# --------------------------------------------------------------------------- #
# egocentric field-of-view manifolds (reference utils.py:1033-1121)
# --------------------------------------------------------------------------- #
def _fov_row_angles(first, last, step):
    """Angles (radians) of one concentric row: cells every `step` from `first + step/2` up to
    `last` on the right of the heading, mirrored on the left."""
    right = np.arange(first + step / 2, last, step)
    return np.concatenate((-right[::-1], right))


def create_uniform_radial_assembly(distance_range=[0.0, 0.2], angle_range=[0, 90], spatial_resolution=0.04,
                                   **kwargs):
    """Concentric rows of receptive fields of constant size `spatial_resolution` tiling the
    field of view: returns lists (mu_d, mu_theta [rad], sigma_d, sigma_theta [rad])."""
    lo, hi = (a * np.pi / 180 for a in angle_range)
    mu_d, mu_theta, sigma_d, sigma_theta = [], [], [], []
    for radius in np.arange(max(0.01, distance_range[0]), distance_range[1], spatial_resolution):
        for theta in _fov_row_angles(lo, hi, spatial_resolution / radius):
            mu_d.append(radius)
            mu_theta.append(theta)
            sigma_d.append(spatial_resolution)
            sigma_theta.append(spatial_resolution / radius)
    return mu_d, mu_theta, sigma_d, sigma_theta


def create_diverging_radial_assembly(distance_range=[0.01, 0.2], angle_range=[0, 90], spatial_resolution=0.04,
                                     beta=5, **kwargs):
    """As the uniform assembly but the field size grows with radius (Hartley et al. 2000):
    size = xi + radius/beta with xi fixed by the innermost row having size `spatial_resolution`;
    successive rows just touch: r_next = (2 r + size + xi) / (2 - 1/beta)."""
    lo, hi = (a * np.pi / 180 for a in angle_range)
    mu_d, mu_theta, sigma_d, sigma_theta = [], [], [], []
    radius = max(0.01, distance_range[0])
    xi = spatial_resolution - radius / beta
    while radius < distance_range[1]:
        size = xi + radius / beta
        step = size / radius
        if step / 2 > hi:
            right = np.array([lo + step / 2])  # at least one cell per row
            thetas = np.concatenate((-right[::-1], right))
        else:
            thetas = _fov_row_angles(lo, hi, step)
        for theta in thetas:
            mu_d.append(radius)
            mu_theta.append(theta)
            sigma_d.append(size)
            sigma_theta.append(size / radius)
        radius = (2 * radius + size + xi) / (2 - 1 / beta)
    return mu_d, mu_theta, sigma_d, sigma_theta


def polygon_area(corners):
    """Area of a simple polygon given by its corners (shoelace); what the reference asks shapely for when it
    discounts the holes in `Environment.sample_positions` (reference Environment.py:605-606)."""
    c = np.asarray(corners, dtype=float).reshape(-1, 2)
    x, y = c[:, 0], c[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def polygon_contains(corners, point):
    """True if `point` lies STRICTLY inside the simple polygon `corners` — a point on an edge or a corner is not
    contained, the semantics of the `shapely.Polygon.contains` the reference calls
    (reference Environment.py:808-816).  Even-odd rule on the ray towards +x; host-side, one point (the init-time
    samplers); the kernels carry their own form (csrc/riab_device.h: polygon_contains_strict)."""
    c = np.asarray(corners, dtype=float).reshape(-1, 2)
    px, py = float(point[0]), float(point[1])
    if not (np.isfinite(px) and np.isfinite(py)):
        return False
    a, b = c, np.roll(c, -1, axis=0)
    ax, ay, bx, by = a[:, 0], a[:, 1], b[:, 0], b[:, 1]
    cross = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
    on_edge = (cross == 0) & (np.minimum(ax, bx) <= px) & (px <= np.maximum(ax, bx)) & \
        (np.minimum(ay, by) <= py) & (py <= np.maximum(ay, by))
    if on_edge.any():
        return False
    straddle = (ay > py) != (by > py)
    with np.errstate(divide="ignore", invalid="ignore"):
        x_cross = ax + (py - ay) * (bx - ax) / (by - ay)
    return bool(np.count_nonzero(straddle & (px < x_cross)) % 2)


# --------------------------------------------------------------------------- #
# the reference's public geometry / statistics helpers, for user code that calls them
# --------------------------------------------------------------------------- #
# NumPy, host side, for scripts written against `ratinabox.utils`.  None of them is on the accelerated path: the kernels
# (csrc/) carry their own arithmetic and nothing in Agent / Neurons / Environment calls these.  Deterministic: the
# reference perturbs the inputs of `vector_intercepts` / `shortest_vectors_from_points_to_lines` by N(0, 1e-9 / 1e-6)
# to dodge exact degeneracies (utils.py:62-67, 150-151); these do not.
def get_perpendicular(a=None):
    """`[x, y] -> [-y, x]` (reference utils.get_perpendicular, utils.py:17-27)."""
    a = np.asarray(a, dtype=float)
    return np.stack((-a[..., 1], a[..., 0]), axis=-1)


def _cross2(u, v):
    return u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]


def vector_intercepts(vector_list_a, vector_list_b, return_collisions=False):
    """Where the LINES through two lists of segments `(N_a, 2, 2)`, `(N_b, 2, 2)` meet, as the pair of line parameters
    `(l_a, l_b)` of every (a, b): shape `(N_a, N_b, 2)`; the segments themselves cross iff both lie strictly in (0, 1).
    `return_collisions=True`: that boolean `(N_a, N_b)` instead; `"as_well"`: both (reference utils.vector_intercepts,
    utils.py:30-118).  With a = p0 + l_a sa, b = q0 + l_b sb and d0 = q0 - p0: l_a = (d0 x sb) / (sa x sb),
    l_b = (d0 x sa) / (sa x sb); parallel segments give inf / nan as in the reference."""
    a = np.asarray(vector_list_a, dtype=float)
    b = np.asarray(vector_list_b, dtype=float)
    assert a.shape[-2:] == (2, 2) and b.shape[-2:] == (2, 2), "vector_list_a and vector_list_b must be shape (_,2,2), _ is optional"
    a, b = a.reshape(-1, 2, 2), b.reshape(-1, 2, 2)
    sa = (a[:, 1] - a[:, 0])[:, None, :]
    sb = (b[:, 1] - b[:, 0])[None, :, :]
    d0 = b[None, :, 0, :] - a[:, None, 0, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        den = _cross2(sa, sb)
        intercepts = np.stack((_cross2(d0, sb) / den, _cross2(d0, sa) / den), axis=-1)
    hit = ((intercepts > 0) & (intercepts < 1)).all(-1)
    if return_collisions is True:
        return hit
    if return_collisions == "as_well":
        return intercepts, hit
    return intercepts


def shortest_vectors_from_points_to_lines(positions, vectors):
    """For positions `(N_p, 2)` and segments `(N_v, 2, 2)`: the shortest vector FROM each segment TO each position,
    `(N_p, N_v, 2)` — the foot of the perpendicular clamped to the segment's ends (reference utils.py:121-184)."""
    p = np.asarray(positions, dtype=float).reshape(-1, 2)
    v = np.asarray(vectors, dtype=float).reshape(-1, 2, 2)
    s = v[:, 1] - v[:, 0]
    d = p[:, None, :] - v[None, :, 0, :]
    lam = np.clip((d * s).sum(-1) / (s * s).sum(-1), 0.0, 1.0)
    return d - lam[..., None] * s


def get_line_segments_between(pos1, pos2):
    """Pairwise segments `[pos1[i], pos2[j]]`: `(N, M, 2, D)` (reference utils.py:187-200)."""
    pos1, pos2 = np.asarray(pos1, dtype=float), np.asarray(pos2, dtype=float)
    a = pos1.reshape(-1, 1, pos1.shape[-1])
    b = pos2.reshape(1, -1, pos2.shape[-1])
    a, b = np.broadcast_arrays(a, b)
    return np.stack((a, b), axis=-2)


def get_vectors_between(pos1=None, pos2=None, line_segments=None):
    """Pairwise `pos1[i] - pos2[j]`, `(N, M, D)` (reference utils.py:203-214: FROM pos2 TO pos1)."""
    if line_segments is None:
        line_segments = get_line_segments_between(pos1, pos2)
    line_segments = np.asarray(line_segments, dtype=float)
    return line_segments[..., 0, :] - line_segments[..., 1, :]


def get_distances_between(pos1=None, pos2=None, vectors=None):
    """Pairwise Euclidean distances `(N, M)` (reference utils.py:217-228)."""
    if vectors is None:
        vectors = get_vectors_between(pos1, pos2)
    return np.linalg.norm(np.asarray(vectors, dtype=float), axis=-1)


def get_bearing(segment, is_array=False):
    """Bearing of a direction vector / segment: clockwise from North (+y), in [0, 2pi) = `mod(pi/2 - get_angle, 2pi)`
    (reference utils.py:276-289)."""
    return np.mod(np.pi / 2 - get_angle(segment, is_array=is_array), 2 * np.pi)


def wall_bounce(current_velocity, wall):
    """Velocity after reflecting off `wall` = [start, end]: the component along the wall is kept, the one across it
    changes sign (reference utils.wall_bounce, utils.py:304-328)."""
    v = np.asarray(current_velocity, dtype=float)
    wall = np.asarray(wall, dtype=float)
    along = wall[1] - wall[0]
    along = along / np.linalg.norm(along)
    across = get_perpendicular(along)
    return along * np.dot(v, along) - across * np.dot(v, across)


def pi_domain(x):
    """Angles recast onto (-pi, pi] (reference utils.pi_domain, utils.py:331-341)."""
    x = np.mod(np.asarray(x, dtype=float), 2 * np.pi)
    return np.where(x > np.pi, x - 2 * np.pi, x)


def ornstein_uhlenbeck(dt, x, drift=0.0, noise_scale=0.2, coherence_time=5.0):
    """Increment `dx` of an Ornstein-Uhlenbeck process in `x` over `dt`: `(drift - x) dt / tau + sigma N(0, dt)` with
    `sigma = sqrt(2 noise_scale^2 / (tau dt))`, one normal per element from the global NumPy stream (reference
    utils.ornstein_uhlenbeck, utils.py:347-368; the motion kernel's form of it: csrc/riab_agent_kernel.h `ou_step`)."""
    x = np.asarray(x, dtype=float)
    tau = coherence_time * np.ones_like(x)
    sigma = np.sqrt(2 * (noise_scale * np.ones_like(x)) ** 2 / (tau * dt))
    return (drift * np.ones_like(x) - x) * dt / tau + sigma * np.random.normal(size=x.shape, scale=dt)


def normal_to_rayleigh(x, sigma=1):
    """N(0, 1) variate -> Rayleigh(sigma) variate through the two CDFs (reference utils.py:409-413)."""
    from scipy import stats
    return sigma * np.sqrt(-2 * np.log(1 - stats.norm.cdf(x)))


def rayleigh_to_normal(x, sigma=1):
    """Rayleigh(sigma) variate -> N(0, 1) variate; the uniform in between is clipped to [1e-6, 1 - 1e-6] (reference
    utils.py:416-421, which takes scalars; arrays work here)."""
    from scipy import stats
    u = np.clip(1 - np.exp(-np.asarray(x, dtype=float) ** 2 / (2 * sigma ** 2)), 1e-6, 1 - 1e-6)
    return stats.norm.ppf(u)


def gaussian(x, mu, sigma, norm=None):
    """`norm * exp(-(x - mu)^2 / 2 sigma^2)`; `norm` = the peak value, default the density's 1 / sqrt(2 pi sigma^2)
    (reference utils.gaussian, utils.py:424-438: a `norm` of 0 or None selects the default)."""
    peak = norm or 1 / np.sqrt(2 * np.pi * np.asarray(sigma, dtype=float) ** 2)
    return peak * np.exp(-((x - mu) ** 2) / (2 * sigma ** 2))


def von_mises(theta, mu, sigma, norm=None):
    """`norm * exp(kappa (cos(theta - mu) - 1))`, `kappa = 1 / sigma^2`; `norm` = the value at the centre, default the
    density's `exp(kappa) / (2 pi I0(kappa))` (reference utils.von_mises, utils.py:441-457)."""
    kappa = 1 / np.asarray(sigma, dtype=float) ** 2
    if not norm:
        from scipy import special
        norm = np.exp(kappa) / (2 * np.pi * special.i0(kappa))
    return np.exp(kappa * np.cos(theta - mu)) * (norm / np.exp(kappa))


_ACTIVATION_DEFAULTS = {"sigmoid": {"max_fr": 1, "min_fr": 0, "mid_x": 1, "width_x": 2}}


def activate(x, activation="sigmoid", deriv=False, other_args={}):
    """The reference's activation functions and their derivatives (utils.activate, utils.py:919-1026), NumPy:
    "linear", "sigmoid" (max_fr, min_fr, mid_x, width_x = distance between the 5 % and 95 % points), "relu", "tanh",
    "retanh", "softmax" (a softplus; gain, threshold each), or `other_args["function"](x, deriv=...)`;
    `other_args["activation"]` overrides `activation`.  (FeedForwardLayer evaluates the same presets on the device:
    csrc/riab_ff.hip.)  The derivatives of "tanh" / "retanh" ignore the threshold inside the tanh, as the
    reference's do."""
    if "function" in other_args:
        return other_args["function"](x, deriv=deriv)
    name = other_args.get("activation", activation)
    assert name in ("linear", "sigmoid", "relu", "tanh", "retanh", "softmax")
    x = np.asarray(x, dtype=float)
    args = dict(_ACTIVATION_DEFAULTS.get(name, {"gain": 1, "threshold": 0}))
    args.update(other_args)
    if name == "linear":
        return np.ones(x.shape) if deriv else x
    if name == "sigmoid":
        lo, hi = args["min_fr"], args["max_fr"]
        beta = np.log(0.95 / 0.05) / (0.5 * args["width_x"])
        f = (hi - lo) / (1 + np.exp(-beta * (x - args["mid_x"]))) + lo
        return beta * (f - lo) * (1 - (f - lo) / (hi - lo)) if deriv else f
    g, z = args["gain"], x - args["threshold"]
    if name == "relu":
        return g * (z > 0) if deriv else g * np.maximum(0, z)
    if name == "tanh":
        return g * (1 - np.tanh(x) ** 2) if deriv else g * np.tanh(z)
    if name == "retanh":
        return g * (1 - np.tanh(x) ** 2) * (z > 0) if deriv else g * np.maximum(0, np.tanh(z))
    return g / (1 + np.exp(-z)) if deriv else g * np.log(1 + np.exp(z))

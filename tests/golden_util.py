"""Helpers shared by the golden-vector tests (loading fixtures, rebuilding the
oracle's EnvSpec / state dicts from a motion_*.npz record)."""
import os

import numpy as np

from oracle import riab_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MOTION_FILES = sorted(f for f in os.listdir(GOLDEN) if f.startswith("motion_") and f.endswith(".npz"))

# layout of a flattened state row written by make_golden.py
PRE_SLICES = dict(pos=slice(0, 2), velocity=slice(2, 4), rotational_velocity=4, measured_velocity=slice(5, 7),
                  head_direction=slice(7, 9), distance_travelled=9)
POST_EXTRA = dict(measured_rotational_velocity=10, distance_to_closest_wall=11)


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def shape_from(g, prefix="env_"):
    """(boundary corners or None, list of hole corner arrays) stored by make_golden.py."""
    b = g[prefix + "boundary"]
    holes, k = [], 0
    for n in g[prefix + "hole_sizes"]:
        holes.append(g[prefix + "holes"][k:k + int(n)])
        k += int(n)
    return (b if len(b) else None), holes


def env_from(g):
    boundary, holes = shape_from(g)
    return orc.EnvSpec(scale=float(g["env_scale"]), aspect=float(g["env_aspect"]),
                       boundary_conditions=str(g["env_bc"]), walls=g["user_walls"], boundary=boundary, holes=holes)


def product_env_params(g, prefix="env_"):
    """The polygon / holes entries of the product Environment's params for a golden file."""
    boundary, holes = shape_from(g, prefix)
    out = {}
    if boundary is not None:
        out["boundary"] = boundary.tolist()
    if holes:
        out["holes"] = [h.tolist() for h in holes]
    return out


def params_from(g):
    p = {str(k): float(v) for k, v in zip(g["params_keys"], g["params_vals"])}
    dt = p.pop("dt", float(g["dt"]))
    kw = {str(k): float(v) for k, v in zip(g["kw_keys"], g["kw_vals"])}
    return p, kw, dt


def state_from_rows(rows):
    rows = np.atleast_2d(rows)
    B = rows.shape[0]
    st = {k: rows[:, s].copy() for k, s in PRE_SLICES.items()}
    st["measured_rotational_velocity"] = np.zeros(B)
    st["distance_to_closest_wall"] = np.full(B, np.inf)
    return st

FF_ACTS = {"linear": {"activation": "linear"},
           "sigmoid": {"activation": "sigmoid", "max_fr": 5, "min_fr": 0.5, "mid_x": 0.2, "width_x": 1.5},
           "relu": {"activation": "relu", "gain": 2.0, "threshold": 0.1},
           "tanh": {"activation": "tanh", "gain": 1.5, "threshold": -0.2},
           "retanh": {"activation": "retanh", "gain": 1.2, "threshold": 0.05},
           "softmax": {"activation": "softmax", "gain": 0.7, "threshold": 0.3}}

TASK_FILES = sorted(f for f in os.listdir(GOLDEN) if f.startswith("task_") and f.endswith(".npz"))
TASKWORLD_FILES = sorted(f for f in os.listdir(GOLDEN) if f.startswith("taskworld_") and f.endswith(".npz") and "list_logic" not in f)

"""GPU tests of SURVEY.md §8 row a6 — polygonal boundaries and holes (reference Environment.py:71-73, 128-163,
781-818, 855-894): the device inside test, apply_boundary_conditions, the resample branch of the motion kernel in
parity mode (the reference's replacement positions handed in) and in production mode (Philox rejection sampling,
bit-exact against the oracle's restatement), PlaceCells under the wall geometries of a polygonal room.  Single steps
and rollouts in an L-shaped room and in a box with two holes run with every other motion golden in
tests/test_gpu_parity.py (motion_lroom_dt20ms.npz, motion_box_holes_dt20ms.npz).

STAND-IN: the reference decides "inside the environment" with shapely, which this image does not have; the goldens used
here (polygon.npz, motion_lroom_*, motion_box_holes_*) were generated with the build's own strict point-in-polygon
(oracle/ref_shims/shapely: even-odd crossings + an exact on-edge test).  For rectangles — all SURVEY 8 asks — strict
interior is unambiguous; for points exactly ON a polygon's or hole's edge these fixtures pin the shim's semantics, not
shapely's own."""
import numpy as np
import pytest
import torch

from oracle import riab_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def riab():
    assert torch.cuda.is_available(), "these tests need the GPU"
    import ratinabox_amd
    return ratinabox_amd


def _env(riab, g, tag):
    return riab.Environment(dict(gu.product_env_params(g, tag + "_"), walls=g[f"{tag}_user_walls"].tolist()))


@pytest.mark.parametrize("tag", ["lroom", "holes", "both"])
def test_inside_test_and_boundary_conditions_vs_reference(riab, tag):
    g = gu.load("polygon.npz")
    env = _env(riab, g, tag)
    pts = g[f"{tag}_points"]
    assert np.array_equal(env.positions_in_environment(pts), g[f"{tag}_inside"])   # bit-exact, edge points included
    np.random.seed(5)  # the generator's seed: the host draws the replacements in the reference's order
    out = env.apply_boundary_conditions(pts)
    np.testing.assert_array_equal(out, g[f"{tag}_bc_out"])
    assert env.positions_in_environment(out).all()


def test_place_cells_in_polygonal_rooms_vs_reference(riab):
    g = gu.load("polygon.npz")
    env = _env(riab, g, "lroom")
    ag = riab.Agent(env)
    for geom in ("euclidean", "line_of_sight"):
        pcs = riab.PlaceCells(ag, {"place_cell_centres": g[f"pc_{geom}_centres"], "widths": 0.15, "wall_geometry": geom,
                                   "description": "gaussian_threshold"})
        got = pcs.get_state(evaluate_at=None, pos=g["pc_pos"])
        ref = g[f"pc_{geom}_rates"]
        assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-5)  # thresholded rates pass through zero
        assert (ref > 0).mean() > 0.03
    quad = riab.Environment({"boundary": g["quad_boundary"].tolist(), "walls": g["quad_user_walls"].tolist()})
    pcs = riab.PlaceCells(riab.Agent(quad), {"place_cell_centres": g["quad_centres"], "widths": 0.2,
                                             "wall_geometry": "geodesic"})
    np.testing.assert_allclose(pcs.get_state(evaluate_at=None, pos=g["quad_pos"]), g["quad_rates"], rtol=1e-5, atol=1e-30)


def test_production_resample_is_the_oracles_philox_rejection_sampling(riab):
    """Production mode (no replacement positions given): agents put into a hole / outside the polygon are moved to
    the first Philox candidate that is inside — bit-exactly the oracle's restatement of the draw — and every
    agent is inside afterwards, in a long run too."""
    g = gu.load("polygon.npz")
    env = _env(riab, g, "both")
    np.random.seed(2)
    B = 256
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.02, "seed": 31, "speed_mean": 0.01})
    boundary, holes = gu.shape_from(g, "both_")
    oenv = orc.EnvSpec(walls=g["both_user_walls"], boundary=boundary, holes=holes)
    pos = np.array(ag.pos)
    bad = np.zeros(B, dtype=bool)
    bad[::3] = True
    pos[::6] = holes[0].mean(axis=0) + 0.01 * (np.random.rand(len(pos[::6]), 2) - 0.5)      # in the hole
    pos[3::6] = [2.45, 0.05] + 0.01 * np.random.rand(len(pos[3::6]), 2)                      # in the box, off the polygon
    ag.pos = pos
    assert not orc.env_is_inside(oenv, pos[bad]).any() and orc.env_is_inside(oenv, pos[~bad]).all()
    ag.update()
    new = np.array(ag.pos)
    # (a 0.2 mm step cannot carry an agent out of the 1 cm patches they were put in: they are all resampled)
    want = orc.resample_draws(31, 0, np.arange(B), oenv)
    np.testing.assert_array_equal(new[bad], want[bad])
    assert env.positions_in_environment(new).all()
    assert ag.diagnostics["boundary_conditions"] == int(bad.sum()) and ag.diagnostics["bounce_saturations"] == 0
    ag.speed_mean = 0.3
    ag.simulate(3000)
    assert env.positions_in_environment(ag.pos).all()
    traj = ag.get_history_tensor()[-3000:, :2].permute(0, 2, 1).reshape(-1, 2).cpu().numpy()
    assert env.positions_in_environment(traj[::37]).all()


@pytest.mark.parametrize("case", ["open", "walls", "narrow", "thigmotaxis"])
def test_box_fast_path_equals_general_wall_arithmetic(riab, case):
    """The trajectory kernel's box fast path (a solid rectangular room: the four boundary edges as coordinate
    differences, csrc/riab_agent_kernel.h) against the general point-to-segment arithmetic: the SAME room handed over
    as a polygonal boundary has the same wall table but takes the general path.  Agents that never needed the
    boundary safety net (clamp in the box, re-sampling in the polygon) must agree to rounding."""
    scale, aspect = (0.19, 1.0) if case == "narrow" else (0.8, 1.5)
    walls = [[[0.5, 0.0], [0.5, 0.45]], [[0.9, 0.8], [0.9, 0.4]]] if case == "walls" else []
    corners = [[0, 0], [aspect * scale, 0], [aspect * scale, scale], [0, scale]]
    box = riab.Environment({"scale": scale, "aspect": aspect, "walls": walls})
    poly = riab.Environment({"boundary": corners, "walls": walls})
    assert np.array_equal(box.walls, poly.walls) and box.is_rectangular and not poly.is_rectangular
    params = {"n_agents": 512, "dt": 0.02, "seed": 7, "agent_id0": 0}
    if case == "thigmotaxis":
        params.update(thigmotaxis=0.95, wall_repel_distance=0.2, wall_repel_strength=2.0, speed_mean=0.2)
    np.random.seed(3)
    a = riab.Agent(box, dict(params))
    b = riab.Agent(poly, dict(params))
    b.state_tensor.copy_(a.state_tensor)
    T = 600
    ta = a.simulate(T, neurons=[])
    tb = b.simulate(T, neurons=[])
    torch.cuda.synchronize()
    sa, sb = a.state_tensor.cpu().numpy()[:, :512], b.state_tensor.cpu().numpy()[:, :512]
    # lanes whose two runs stayed on the common code path: no boundary-condition event in either
    moved = (ta[:, :2, :512] - tb[:, :2, :512]).abs().amax(dim=(0, 1)).cpu().numpy()
    same = moved < 1e-3
    assert same.mean() > 0.97, same.mean()   # (an escape through a corner is rare; those lanes diverge by design)
    np.testing.assert_allclose(sa[:, same], sb[:, same], rtol=0, atol=2e-9)
    assert a.diagnostics["bounce_saturations"] == 0
    if case == "narrow":   # narrower than twice the repel distance: the box itself takes the general path, bit for bit
        assert np.array_equal(sa[:, same], sb[:, same])

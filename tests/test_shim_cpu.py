"""The `shapely` stand-in behind the polygon / hole fixtures (oracle/ref_shims/shapely: strict even-odd point-in-polygon
with an exact on-edge test) against an INDEPENDENT implementation: matplotlib.path.Path.contains_points.

DESIGN.md 5: polygon.npz, motion_box_holes_* and motion_lroom_* record the reference running on this stand-in.  Off the
edges the two implementations must agree on every point; ON an edge or a corner shapely's `Polygon.contains` (which the
reference calls, Environment.py:808-816) says "outside", and so does the stand-in; matplotlib has no such rule — those
cases are LISTED, not asserted."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(ROOT, "oracle", "ref_shims")

# every polygon tests/golden/make_golden.py builds an Environment from (its L_ROOM, HOLE_A, HOLE_B, make_polygon's "both"),
# and the unit box
POLYGONS = {
    "box": [[0, 0], [1, 0], [1, 1], [0, 1]],
    "l_room": [[0, 0], [1, 0], [1, 0.5], [0.5, 0.5], [0.5, 1], [0, 1]],
    "hole_a": [[0.35, 0.35], [0.65, 0.35], [0.65, 0.65], [0.35, 0.65]],
    "hole_b": [[0.1, 0.7], [0.25, 0.7], [0.22, 0.9], [0.12, 0.88]],
    "both_boundary": [[0, 0], [2, 0], [2.5, 1], [1, 1.5], [-0.2, 1]],
    "both_hole": [[0.5, 0.4], [1.0, 0.4], [0.7, 0.8]],
}


@pytest.fixture(scope="module")
def shim():
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "shapely" or k.startswith("shapely.")}
    sys.path.insert(0, SHIMS)
    try:
        import shapely
        assert os.path.dirname(os.path.abspath(shapely.__file__)).startswith(SHIMS)
        yield shapely
    finally:
        sys.path.remove(SHIMS)
        for k in [k for k in sys.modules if k == "shapely" or k.startswith("shapely.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_generator_polygons_are_the_ones_checked_here():
    """(the list above is kept by hand: it must contain the generator's polygons)"""
    src = open(os.path.join(ROOT, "tests", "golden", "make_golden.py")).read()
    for name in ("l_room", "hole_a", "hole_b"):
        text = repr(POLYGONS[name]).replace(" ", "")
        assert text in src.replace(" ", ""), name
    assert repr(POLYGONS["both_boundary"]).replace(" ", "") in src.replace(" ", "")
    assert repr(POLYGONS["both_hole"]).replace(" ", "") in src.replace(" ", "")


@pytest.mark.parametrize("name", sorted(POLYGONS))
def test_stand_in_agrees_with_matplotlib_off_the_edges(shim, name):
    from matplotlib.path import Path
    poly = np.array(POLYGONS[name], float)
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    lo, hi = poly.min(0) - 0.3, poly.max(0) + 0.3
    pts = rng.uniform(lo, hi, size=(4000, 2))
    # ... and points close to (but not on) the edges and corners, where a sloppy crossing rule goes wrong first
    a, b = poly, np.roll(poly, -1, axis=0)
    t = rng.uniform(0, 1, size=(len(poly), 60, 1))
    on = a[:, None, :] + t * (b - a)[:, None, :]
    normal = np.stack(((b - a)[:, 1], -(b - a)[:, 0]), -1)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    near = (on + rng.choice([-1, 1], size=t.shape) * rng.uniform(1e-9, 1e-3, size=t.shape) * normal[:, None, :]).reshape(-1, 2)
    corners = (poly[:, None, :] + rng.uniform(-1e-6, 1e-6, size=(len(poly), 20, 2))).reshape(-1, 2)
    pts = np.concatenate((pts, near, corners))
    P = shim.Polygon(poly)
    ours = np.array([P.contains(shim.Point(p)) for p in pts])
    # exact arithmetic says whether a point is ON an edge; leave those to the listing test
    on_edge = np.array([P._on_boundary(float(p[0]), float(p[1])) for p in pts])
    theirs = Path(poly, closed=False).contains_points(pts, radius=0.0)
    keep = ~on_edge
    assert keep.sum() > 4000
    bad = np.nonzero(ours[keep] != theirs[keep])[0]
    assert bad.size == 0, (name, pts[keep][bad[:5]], ours[keep][bad[:5]])
    assert 0.02 < ours.mean() < 0.98          # (both answers occur)


def test_on_edge_points_are_outside_for_the_stand_in_and_listed_for_matplotlib(shim, capsys):
    """shapely: `contains` is the strict interior — a point on the boundary is not contained.  The stand-in implements
    that with an exact test; matplotlib's answer on the same points is whatever its crossing rule gives: printed."""
    from matplotlib.path import Path
    lines = []
    for name in sorted(POLYGONS):
        poly = np.array(POLYGONS[name], float)
        P = shim.Polygon(poly)
        a, b = poly, np.roll(poly, -1, axis=0)
        mids = 0.5 * (a + b)
        # points representable exactly on axis-aligned edges, the midpoints of the others (on the edge only if exact
        # arithmetic says so), and the corners themselves
        pts = np.concatenate((poly, mids))
        exact = np.array([P._on_boundary(float(p[0]), float(p[1])) for p in pts])
        assert exact[:len(poly)].all(), name                       # every corner is on the boundary
        ours = np.array([P.contains(shim.Point(p)) for p in pts])
        assert not ours[exact].any(), name                         # ... and on the boundary means outside
        theirs = Path(poly, closed=False).contains_points(pts, radius=0.0)
        differ = np.nonzero(exact & theirs)[0]
        lines.append(f"{name}: {int(exact.sum())} of {len(pts)} corner / midpoint samples lie exactly on the boundary; "
                     f"matplotlib calls {len(differ)} of them inside: {[tuple(round(float(v), 3) for v in pts[i]) for i in differ[:6]]}")
    with capsys.disabled():
        print("\n[on-edge points: stand-in says outside (shapely's rule); matplotlib]\n  " + "\n  ".join(lines))

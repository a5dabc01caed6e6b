"""Minimal stand-in for `shapely`, used ONLY to import the reference here.

TEST INFRASTRUCTURE (oracle side). The upstream package does `import shapely`
at module import (reference ratinabox/Environment.py:8) but shapely is not
installed in this image and cannot be installed (no network).  The only use of
shapely on the accelerated path is the strict point-in-polygon test in
`Environment.check_if_position_is_in_environment` (reference
ratinabox/Environment.py:808-816); everything else is holes/plots.

This module provides exactly the surface the reference touches:
`Point`, `Polygon(coords).contains(Point)` (strict interior, boundary points
are outside, like shapely), `.area`, `MultiPolygon`, `geometry.Polygon`.
It is never imported by the product package and never travels to the GPU box
as part of a code path (tests/golden/*.npz are data generated with it).
"""
import sys
import types

import numpy as np


class Point:
    def __init__(self, *xy):
        if len(xy) == 1:
            xy = xy[0]
        self.xy = np.asarray(xy, dtype=float).reshape(-1)
        self.x, self.y = float(self.xy[0]), float(self.xy[1])


class Polygon:
    def __init__(self, coords):
        self.coords = np.asarray(coords, dtype=float).reshape(-1, 2)

    @property
    def area(self):
        x, y = self.coords[:, 0], self.coords[:, 1]
        return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))

    def _on_boundary(self, px, py):
        a = self.coords
        b = np.roll(self.coords, -1, axis=0)
        for (ax, ay), (bx, by) in zip(a, b):
            cross = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
            if cross != 0.0:
                continue
            if min(ax, bx) <= px <= max(ax, bx) and min(ay, by) <= py <= max(ay, by):
                return True
        return False

    def contains(self, point):
        px, py = point.x, point.y
        if not (np.isfinite(px) and np.isfinite(py)):
            return False
        if self._on_boundary(px, py):
            return False
        inside = False
        a = self.coords
        b = np.roll(self.coords, -1, axis=0)
        for (ax, ay), (bx, by) in zip(a, b):
            if (ay > py) != (by > py):
                x_cross = ax + (py - ay) * (bx - ax) / (by - ay)
                if px < x_cross:
                    inside = not inside
        return inside

    def difference(self, other):  # plotting only
        return self


class MultiPolygon:  # plotting only
    geoms = ()


geometry = types.ModuleType("shapely.geometry")
geometry.Polygon = Polygon
geometry.Point = Point
geometry.MultiPolygon = MultiPolygon
sys.modules["shapely.geometry"] = geometry

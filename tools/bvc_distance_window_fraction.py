"""How many of the terms the BVC kernel issues could be skipped by DISTANCE if the 64 positions of a tile were neighbours
in space (VERDICT r4 #4b): agents sorted once per chunk into a coarse grid (cell side `side`), a tile =
64 agents of one grid cell, a (4-cell group, 4 directions) block of the accumulate stage skipped when EVERY one of its
64 x 16 terms has a radial factor below 2^-20 (|d - mu| > 5.27 sigma: less than 1e-6 of the term's peak) — on top of the
direction windows.  Default tunings, K = 180, 4096 agents uniform in the room.  Prints the skippable fraction of the
issued terms for the open box and for cfg 3's nine-wall maze, for tiles of neighbours and for today's unsorted tiles."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import riab_oracle as orc  # noqa: E402

rng = np.random.default_rng(1)
K, n, B = 180, 256, 4096
ang, dirs = orc.bvc_test_angles(2)
mu_d = rng.uniform(0.05, 0.3, n)
sg_d = 0.08 + mu_d / 12
mu_t = rng.uniform(0, 2 * np.pi, n)
sg_t = np.deg2rad(rng.uniform(10, 30, n))
LOG2E = np.log2(np.e)
vm = LOG2E * (1 / sg_t ** 2)[:, None] * (np.cos(ang[None, :] - mu_t[:, None]) - 1)
w = np.exp2(vm)
idx = np.argsort(w, axis=1)
dropped = np.cumsum(np.take_along_axis(w, idx, axis=1), axis=1) <= 1e-6 * w.sum(axis=1)[:, None]
keep = np.ones((n, K), dtype=bool)
np.put_along_axis(keep, idx, ~dropped, axis=1)
band = np.minimum(keep.sum(1) // 24, 7)
order = np.lexsort((np.mod(mu_t, 2 * np.pi), band))
mu_d, sg_d, keep = mu_d[order], sg_d[order], keep[order]
# direction windows per group of four, whole quads of directions (as the kernel has them)
issued = np.zeros((n // 4, K // 4), dtype=bool)
for g in range(n // 4):
    u = keep[4 * g:4 * g + 4].any(0)
    issued[g] = u.reshape(K // 4, 4).any(1)   # (an upper bound on the kernel's contiguous window: good enough here)
MAZE = [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]
for name, interior in (("open box", []), ("cfg 3 maze", MAZE)):
    env = orc.EnvSpec(walls=interior)
    pos = rng.uniform(0.01, 0.99, (B, 2))
    d = orc.bvc_ray_distances(pos, env.walls, dirs)            # (B, K)
    far = np.abs(d[:, None, :] - mu_d[None, :, None]) > 5.27 * sg_d[None, :, None]   # (B, n, K): term negligible
    for label, side in (("neighbours, 1/8 m grid", 1 / 8), ("neighbours, 1/16 m grid", 1 / 16), ("unsorted (today)", None)):
        if side is None:
            tiles = [np.arange(t, t + 64) for t in range(0, B, 64)]
        else:
            cell = (np.floor(pos[:, 0] / side) * 1000 + np.floor(pos[:, 1] / side)).astype(int)
            o = np.argsort(cell, kind="stable")
            tiles = [o[t:t + 64] for t in range(0, B, 64)]     # (tiles of 64 consecutive agents in grid-cell order)
        skip = tot = 0
        for t in tiles:
            f = far[t].all(axis=0)                              # (n, K): negligible for all 64 positions
            blk = f.reshape(n // 4, 4, K // 4, 4).all(axis=(1, 3))   # per (group, direction quad)
            tot += issued.sum()
            skip += (blk & issued).sum()
        print(f"{name:11s} {label:24s}: {skip / tot:.3f} of the issued (group, direction-quad) blocks are skippable by distance")

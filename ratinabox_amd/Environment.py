"""`Environment` — host-side mirror of the reference's Environment class for the
batched hot path (reference ratinabox/Environment.py:20-895).

Same constructor / parameter dictionary / attributes that `Agent` and `Neurons`
read: `walls (N_w,2,2)` in reference order, `extent`, `scale`, `aspect`, `D`,
`dimensionality`, `boundary_conditions`, `Agents`, `flattened_discrete_coords`,
`add_wall`, `sample_positions`, `discretise_environment`.  The geometry QUERIES
of the reference (wall collisions, vectors from walls, distances accounting for
walls, inside test, boundary conditions: Environment.py:657-894) are not Python
here: they are inlined into the HIP kernels, which read the device copy of the
wall table kept by `device_tables()`.

In scope: 2D environments — rectangular boxes (solid or periodic) and simple-polygon
boundaries (solid), with interior walls and polygonal holes (reference
Environment.py:71-73, 128-163).  1D environments are outside the accelerated path and
raise NotImplementedError."""
import copy
import warnings

import numpy as np

from . import utils


class Environment:
    default_params = {
        "dimensionality": "2D",
        "boundary_conditions": "solid",  # "solid" or "periodic"
        "scale": 1,        # metres
        "aspect": 1,       # width / height of the rectangular box
        "dx": 0.01,        # discretisation used by evaluate_at="all"
        "boundary": None,  # corners [[x0,y0],[x1,y1],...] of a simple polygon bounding the environment (None: the box)
        "walls": [],       # interior walls [[x0,y0],[x1,y1]]
        "holes": [],       # corners [[[x0,y0],...],...] of polygonal holes inside the environment
        "objects": [],     # object positions [[x0,y0],...] (all type 0); or use add_object
    }

    def __init__(self, params={}):
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        utils.update_class_params(self, self.params, get_all_defaults=True)
        utils.check_params(self, params.keys())

        self.Agents = []
        self.agents_dict = {}
        if self.dimensionality != "2D":
            raise NotImplementedError("ratinabox_amd accelerates 2D environments only")
        self._device_cache = {}
        if self.boundary_conditions not in ("solid", "periodic"):
            raise ValueError("boundary_conditions must be 'solid' or 'periodic'")
        self.D = 2
        self.is_rectangular = self.boundary is None
        if self.is_rectangular:
            b = [[0, 0], [self.aspect * self.scale, 0], [self.aspect * self.scale, self.scale], [0, self.scale]]
        else:
            b = np.asarray(self.boundary, dtype=float).reshape(-1, 2).tolist()
            assert len(b) >= 3, "a polygonal boundary needs at least 3 corners"
            if self.boundary_conditions == "periodic":
                # The reference's warning and what it announces (Environment.py:130-136).  The reference itself then
                # rewrites only params["boundary_conditions"]: its attribute stays "periodic" and the polygon gets no
                # boundary walls — an accident; here the announced change is carried out.
                warnings.warn("Periodic boundary conditions are only allowed in rectangual environments. Changing "
                              "boundary conditions to 'solid'.")
                self.params["boundary_conditions"] = "solid"
                self.boundary_conditions = "solid"
        self.boundary = b
        self.walls = np.array(self.walls, dtype=float).reshape(-1, 2, 2)
        self._wall_is_hole = [False] * len(self.walls)
        self._n_boundary = 0
        if self.boundary_conditions == "solid":
            # reference order (Environment.py:137-144): boundary wall i runs from corner i+1 to corner i
            nb = len(b)
            boundary_walls = np.array([[b[(i + 1) % nb], b[i]] for i in range(nb)], dtype=float)
            self.walls = np.vstack((boundary_walls, self.walls))
            self._wall_is_hole = [False] * nb + self._wall_is_hole
            self._n_boundary = nb
        # holes: their edges follow the walls (Environment.py:146-162)
        holes, self.holes = list(self.holes), []
        self.holes_polygons = []   # (corner arrays; the reference keeps shapely polygons)
        self.has_holes = False
        if len(holes) > 0:
            assert np.array(holes, dtype=object).ndim >= 1 and all(np.ndim(h) == 2 for h in holes), \
                "Incorrect dimensionality for holes list. It must be a list of lists of coordinates"
            for h in holes:
                self.add_hole(h)
        self.boundary_polygon = np.asarray(b, dtype=float)
        left, right = min(c[0] for c in b), max(c[0] for c in b)
        bottom, top = min(c[1] for c in b), max(c[1] for c in b)
        left, right, bottom, top = float(left), float(right), float(bottom), float(top)
        self.centre = np.array([(left + right) / 2, (top + bottom) / 2])
        self.extent = np.array([left, right, bottom, top])
        # objects seen by ObjectVectorCells (Environment.py:166-176)
        self.passed_in_objects = copy.deepcopy(self.objects)
        self.objects = {"objects": np.empty((0, self.D)), "object_types": np.empty(0, int)}
        self.n_object_types = 0
        for o in self.passed_in_objects:
            self.add_object(o, type=0)
        self.discrete_coords = self.discretise_environment(dx=self.dx)
        self.flattened_discrete_coords = self.discrete_coords.reshape(-1, self.discrete_coords.shape[-1])
        self._device_cache = {}
        self.query_device = "cuda"  # where the geometry queries (get_distances_between... etc.) run

    @classmethod
    def get_all_default_params(cls, verbose=False):
        all_params = utils.collect_all_params(cls, dict_name="default_params")
        if verbose:
            import pprint
            pprint.pprint(all_params)
        return all_params

    # -- agents registry (Environment.py:250-330) ---------------------------------------
    def add_agent(self, agent=None):
        assert agent is not None, "agent must be an Agent"
        if agent.name in self.agents_dict:
            name = f"agent_{len(self.Agents)}"
            if name in self.agents_dict:
                raise ValueError(f"Agents named {agent.name} and {name} already exist; choose a unique name")
            warnings.warn(f"An agent with the name {agent.name} already exists. Renaming to {name}")
            agent.name = name
        self.Agents.append(agent)
        self.agents_dict[agent.name] = agent

    def agent_lookup(self, agent_names=None):
        """Agents by name: a name or a list of names -> list of Agents (None -> None); unknown names
        raise ValueError (Environment.py:220-276)."""
        if agent_names is None:
            return None
        if isinstance(agent_names, str):
            agent_names = [agent_names]
        return [self._agent_lookup(name) for name in agent_names]

    def _agent_lookup(self, agent_name):
        if agent_name is None:
            return None
        if agent_name in self.agents_dict:
            return self.agents_dict[agent_name]
        for agent in self.Agents:
            if agent.name == agent_name:
                self.agents_dict[agent_name] = agent
                return agent
        raise ValueError("Agent name not found in Environment.agents list. Make sure the there no typos. agent name is "
                         "case sensitive")

    def remove_agent(self, agent=None):
        if isinstance(agent, str):
            agent = self._agent_lookup(agent)
        if agent is None:
            return None
        self.Agents.remove(agent)
        self.agents_dict.pop(agent.name)

    # -- geometry edits -------------------------------------------------------------------
    def add_wall(self, wall):
        """Append one wall [[x1,y1],[x2,y2]] (Environment.py:330-342)."""
        wall = np.asarray(wall, dtype=float).reshape(1, 2, 2)
        self.walls = wall if len(self.walls) == 0 else np.concatenate((self.walls, wall), axis=0)
        self._wall_is_hole.append(False)
        self._device_cache.clear()

    def add_hole(self, hole):
        """Add a polygonal hole [[x1,y1],[x2,y2],...] (>= 3 corners): its edges are appended to `walls`
        (edge i runs from corner i+1 to corner i) and its interior stops being part of the environment
        (reference Environment.py:344-364)."""
        hole = np.asarray(hole, dtype=float).reshape(-1, 2)
        assert len(hole) >= 3, "holes must have at least 3 corners"
        self.holes.append(hole.tolist())
        self.has_holes = True
        k = len(hole)
        hole_walls = np.array([[hole[(i + 1) % k], hole[i]] for i in range(k)], dtype=float)
        self.walls = hole_walls if len(self.walls) == 0 else np.vstack((self.walls, hole_walls))
        self._wall_is_hole.extend([True] * k)
        self.holes_polygons.append(hole)
        self._device_cache.clear()

    def add_object(self, object, type="new"):
        """Add an object at `object` (x, y).  type: "new" (a new type id), "same" (the last
        object's type) or an existing / the next integer type (Environment.py:366-395)."""
        object = np.array(object, dtype=float).reshape(1, -1)
        assert object.shape[1] == self.D
        if type == "new":
            type = self.n_object_types
        elif type == "same":
            type = 0 if len(self.objects["object_types"]) == 0 else self.objects["object_types"][-1]
        else:
            assert type <= self.n_object_types, (f"Newly added object must be one of the existing types or the next "
                                                  f"one along ({self.n_object_types}), not {type}")
        self.objects["objects"] = np.append(self.objects["objects"], object, axis=0)
        self.objects["object_types"] = np.append(self.objects["object_types"], np.array([type], int), axis=0)
        self.n_object_types = len(np.unique(self.objects["object_types"]))

    # -- sampling (Environment.py:560-633) ------------------------------------------------
    def sample_positions(self, n=10, method="uniform_jitter"):
        """n positions in the box: "random", "uniform" (a grid, x fastest) or
        "uniform_jitter"; draws from the global np.random state in the reference's order."""
        ex = self.extent
        constrained = (not self.is_rectangular) or self.has_holes
        if method == "random":
            positions = np.zeros((n, 2))
            positions[:, 0] = np.random.uniform(ex[0], ex[1], size=n)
            positions[:, 1] = np.random.uniform(ex[2], ex[3], size=n)
            if constrained:
                # positions outside the polygon / inside a hole are redrawn until they pass, in the reference's
                # (recursive) draw order (Environment.py:593-600)
                for i in range(n):
                    if not self.check_if_position_is_in_environment(positions[i]):
                        positions[i] = self.sample_positions(n=1, method="random").reshape(-1)
            return positions
        if method[:7] == "uniform":
            area = (ex[1] - ex[0]) * (ex[3] - ex[2])
            if self.has_holes:
                area -= sum(utils.polygon_area(h) for h in self.holes)
            delta = np.sqrt(area / n)
            x = np.linspace(ex[0] + delta / 2, ex[1] - delta / 2, int((ex[1] - ex[0]) / delta))
            y = np.linspace(ex[2] + delta / 2, ex[3] - delta / 2, int((ex[3] - ex[2]) / delta))
            positions = np.array(np.meshgrid(x, y)).reshape(2, -1).T
            if constrained:  # grid points that are not legal positions are dropped (and made up for below)
                keep = [i for i, p in enumerate(positions) if self.check_if_position_is_in_environment(p)]
                positions = positions[keep]
            n_uniform = positions.shape[0]
            if method[7:] == "_jitter":
                positions = positions + np.random.uniform(-0.45 * delta, 0.45 * delta, positions.shape)
            n_remaining = n - n_uniform
            if n_remaining > 0:
                pick = np.random.choice(range(len(positions)), n_remaining, replace=True)
                extra = np.array([positions[i] for i in pick])
                delta /= 2
                extra = extra + np.random.uniform(-0.45 * delta, 0.45 * delta, extra.shape)
                positions = np.vstack((positions, extra))
            return positions
        raise ValueError(f"unknown sampling method {method}")

    def discretise_environment(self, dx=None):
        """(Ny, Nx, 2) grid of positions, y descending (Environment.py:635-655)."""
        dx = self.dx if dx is None else dx
        minx, maxx, miny, maxy = [float(v) for v in self.extent]
        self.x_array = np.arange(minx + dx / 2, maxx, dx)
        self.y_array = np.arange(miny + dx / 2, maxy, dx)[::-1]
        xm, ym = np.meshgrid(self.x_array, self.y_array)
        return np.stack((xm, ym), axis=-1)

    def check_if_position_is_in_environment(self, pos):
        """True if `pos` is strictly inside the boundary and not strictly inside a hole (reference
        Environment.py:781-818; host-side, one position: what the init-time samplers call per point.  The
        per-step test runs in the motion kernel, the batched one in `apply_boundary_conditions`)."""
        pos = np.asarray(pos, dtype=float).reshape(-1)
        if self.is_rectangular:
            e = self.extent
            inside = bool((pos[0] > e[0]) and (pos[0] < e[1]) and (pos[1] > e[2]) and (pos[1] < e[3]))
        else:
            inside = utils.polygon_contains(self.boundary_polygon, pos)
        if inside and self.has_holes:
            inside = not any(utils.polygon_contains(h, pos) for h in self.holes_polygons)
        return inside

    # -- geometry queries on device (the stand-alone forms of what the kernels inline) ----------
    def _query_rows(self, pos):
        """(P,2) host positions -> device float64 rows x [P], y [P]."""
        import torch
        a = np.ascontiguousarray(np.asarray(pos, dtype=np.float64).reshape(-1, 2).T)
        t = torch.from_numpy(a).to(self.query_device)
        return t[0], t[1]

    def get_vectors_between___accounting_for_environment(self, pos1=None, pos2=None, line_segments=None):
        """Pairwise vectors `pos1[i] - pos2[j]` `(N1, N2, 2)`, wrapped round a periodic box
        (reference Environment.py:657-675).  `line_segments (N1, N2, 2, 2)` as in the reference:
        [..., 0, :] = pos1[i], [..., 1, :] = pos2[j]."""
        return self._pairwise(pos1, pos2, line_segments, "euclidean", want_dist=False)

    def get_distances_between___accounting_for_environment(self, pos1, pos2, wall_geometry="euclidean",
                                                           return_vectors=False):
        """Pairwise distances `(N1, N2)` under the wall geometry ("euclidean", "line_of_sight": 1000
        where an internal wall blocks the view, "geodesic": round the single internal wall)
        (reference Environment.py:677-779)."""
        if wall_geometry != "euclidean":
            assert self.boundary_conditions == "solid", f"{wall_geometry} geometry is only possible with solid boundaries"
        if wall_geometry == "geodesic":
            assert len(self.walls) <= 5, "unfortunately geodesic geometry is only defined in closed rooms with one additional wall"
        d, v = self._pairwise(pos1, pos2, None, wall_geometry, want_dist=True)
        return (d, v) if return_vectors else d

    def _pairwise(self, pos1, pos2, line_segments, wall_geometry, want_dist):
        import torch
        from . import _lib as L
        if line_segments is not None:
            seg = np.asarray(line_segments, dtype=np.float64)
            pos1, pos2 = seg[:, 0, 0, :], seg[0, :, 1, :]
        x1, y1 = self._query_rows(pos1)
        x2, y2 = self._query_rows(pos2)
        n1, n2 = x1.numel(), x2.numel()
        env, _keep = self.device_tables(self.query_device)
        vec = torch.empty((2, n1, n2), dtype=torch.float64, device=x1.device)
        dist = torch.empty((n1, n2), dtype=torch.float64, device=x1.device) if want_dist else None
        rc = L.lib.riab_env_pairwise(env, L.ptr(x1), L.ptr(y1), n1, L.ptr(x2), L.ptr(y2), n2, L.GEOMETRIES[wall_geometry],
                                     L.ptr(dist), L.ptr(vec[0]), L.ptr(vec[1]), L.current_stream())
        L.check(rc, "riab_env_pairwise")
        v = vec.permute(1, 2, 0).cpu().numpy()
        return (dist.cpu().numpy(), v) if want_dist else v

    def vectors_from_walls(self, pos):
        """Shortest vectors from every wall to `pos`: `(N_walls, 2)` for one position (reference
        Environment.py:843-853), `(P, N_walls, 2)` for `(P, 2)` positions (batched extension)."""
        import torch
        from . import _lib as L
        single = np.ndim(pos) == 1
        x, y = self._query_rows(pos)
        env, _keep = self.device_tables(self.query_device)
        out = torch.empty((max(int(env.n_walls), 1), 2, x.numel()), dtype=torch.float64, device=x.device)
        L.check(L.lib.riab_env_vectors_from_walls(env, L.ptr(x), L.ptr(y), x.numel(), L.ptr(out), L.current_stream()),
                "riab_env_vectors_from_walls")
        v = out[:int(env.n_walls)].permute(2, 0, 1).cpu().numpy()
        return v[0] if single else v

    def check_wall_collisions(self, proposed_step):
        """(walls, collisions): does the step `[[x0, y0], [x1, y1]]` strictly cross each wall?
        `(N_walls,)` bools for one step (reference Environment.py:820-841), `(P, N_walls)` for
        `(P, 2, 2)` steps (batched extension)."""
        import torch
        from . import _lib as L
        if self.walls is None or len(self.walls) == 0:
            return (None, None)
        step = np.asarray(proposed_step, dtype=np.float64)
        single = step.ndim == 2
        step = step.reshape(-1, 2, 2)
        x0, y0 = self._query_rows(step[:, 0])
        x1, y1 = self._query_rows(step[:, 1])
        env, _keep = self.device_tables(self.query_device)
        out = torch.empty((int(env.n_walls), x0.numel()), dtype=torch.uint8, device=x0.device)
        L.check(L.lib.riab_env_check_wall_collisions(env, L.ptr(x0), L.ptr(y0), L.ptr(x1), L.ptr(y1), x0.numel(),
                                                     L.ptr(out), L.current_stream()), "riab_env_check_wall_collisions")
        hit = out.t().cpu().numpy().astype(bool)
        return (self.walls, hit[0] if single else hit)

    def apply_boundary_conditions(self, pos):
        """Positions inside the environment are returned unchanged.  Outside a rectangular box: clamped 1 cm
        inside it (solid) or wrapped (periodic).  Inside a hole, or outside a polygonal boundary: replaced by
        `sample_positions(n=1, method="random")` — drawn from np.random here on the host, in the reference's
        order (reference Environment.py:855-894).  `(2,)` or `(P, 2)`."""
        import torch
        from . import _lib as L
        single = np.ndim(pos) == 1
        x, y = self._query_rows(pos)
        xy = torch.stack((x, y)).contiguous()
        flags = torch.empty(x.numel(), dtype=torch.uint8, device=xy.device)
        env, _keep = self.device_tables(self.query_device)
        L.check(L.lib.riab_env_boundary_conditions(env, L.ptr(xy[0]), L.ptr(xy[1]), x.numel(), L.ptr(flags), 1,
                                                   L.current_stream()), "riab_env_boundary_conditions")
        out = xy.t().cpu().numpy()
        for i in np.flatnonzero(flags.cpu().numpy() == 2):  # (2: the kernel leaves the random draw to the caller)
            out[i] = self.sample_positions(n=1, method="random").reshape(-1)
        return out[0] if single else out

    def positions_in_environment(self, pos):
        """check_if_position_is_in_environment for `(P, 2)` positions at once, on the device -> `(P,)` bools
        (batched extension)."""
        import torch
        from . import _lib as L
        x, y = self._query_rows(pos)
        flags = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
        env, _keep = self.device_tables(self.query_device)
        L.check(L.lib.riab_env_boundary_conditions(env, L.ptr(x), L.ptr(y), x.numel(), L.ptr(flags), 0,
                                                   L.current_stream()), "riab_env_boundary_conditions")
        return flags.cpu().numpy() == 1

    # -- device tables --------------------------------------------------------------------
    # walls from which the motion step gets a broad phase (include/riab_hip.h RiabMotion.wall_grid): any room with an interior
    # wall.  (Built for rooms of dozens of walls — 10.1 -> 3.8 us per step at 64 —; at the nine walls of BASELINE configs[2]
    # the throughput kernel neither gains nor loses (196-198 M either way), the latency-bound one-launch step gains 2 us
    # (closed loop 52.7 -> 50.7 us per step): from 13 walls until that was measured.)
    WALL_GRID_FROM = 5

    def wall_grid(self, device, wd, lmax=0.02):
        """The broad phase of the motion step for wall-heavy rooms (include/riab_hip.h: RiabMotion.wall_grid): per cell of
        a 16 x 16 grid over the extent, [0] the walls that can be the nearest wall of, or lie within `wd` (the wall repel
        distance) of, a point of the cell, [1] the walls within `lmax` of the cell — the only ones a step of at most that
        length can cross.  Conservative supersets from float64 NumPy: a cell is treated as the disc around its centre
        that contains it (+ 1e-6), `dist(centre, wall) -/+ radius` bound the distance of its points from below / above.
        -> (device int64 tensor [G * G, 2], G, wd, lmax), or None for rooms with fewer than WALL_GRID_FROM walls; cached
        on the wall table and wd."""
        import torch
        from . import _lib
        walls = np.asarray(self.walls, dtype=np.float64).reshape(-1, 4)
        if len(walls) < self.WALL_GRID_FROM or len(walls) > _lib.MAX_WALLS or _lib.env("RIAB_NO_WALL_GRID") == "1":
            return None   # (RIAB_NO_WALL_GRID=1: A/B and the tests' comparator — every wall, every step)
        slot = ("wall_grid", str(device))
        key = (walls.tobytes(), tuple(float(e) for e in self.extent), float(wd), float(lmax))
        hit = self._device_cache.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        G = _lib.WALL_GRID_MAX
        e0, e1, e2, e3 = (float(e) for e in self.extent)
        cw, ch = (e1 - e0) / G, (e3 - e2) / G
        cx = e0 + (np.arange(G) + 0.5) * cw
        cy = e2 + (np.arange(G) + 0.5) * ch
        C = np.stack(np.meshgrid(cx, cy, indexing="xy"), -1).reshape(-1, 2)            # cell iy * G + ix
        a, s_ = walls[:, :2], walls[:, 2:] - walls[:, :2]
        ss = np.maximum((s_ ** 2).sum(1), 1e-300)
        lam = np.clip(((C[:, None, :] - a[None]) * s_[None]).sum(-1) / ss[None], 0.0, 1.0)
        d = np.linalg.norm(C[:, None, :] - (a[None] + lam[..., None] * s_[None]), axis=-1)   # [cells, walls]
        r = 0.5 * np.hypot(cw, ch) * (1 + 1e-6) + 1e-6 * max(e1 - e0, e3 - e2)
        upper = (d + r).min(1, keepdims=True)                       # some wall is at most this far from every point of the cell
        near = (d - r) <= np.maximum(upper, wd * (1 + 1e-5)) + 1e-9
        coll = (d - r) <= lmax + 1e-9
        bits = (1 << np.arange(len(walls), dtype=np.uint64))
        tab = np.stack(((near * bits).sum(1, dtype=np.uint64), (coll * bits).sum(1, dtype=np.uint64)), -1)
        out = (torch.from_numpy(tab.view(np.int64).copy()).to(device), G, float(wd), float(lmax))
        self._device_cache[slot] = (key, out)
        return out

    def device_tables(self, device):
        """(RiabEnv struct, walls tensor): device copy of the wall table in the
        layout include/riab_hip.h documents.  Rebuilt when `walls` changed."""
        import torch
        from . import _lib
        slot = ("env", str(device))  # ("cuda" and "cuda:0" name the same device but are cached separately)
        hit = self._device_cache.get(slot)
        w_now = self.walls
        # the wall table as bytes: catches in-place edits of Environment.walls at ~1 us per call
        w_bytes = w_now.tobytes() if type(w_now) is np.ndarray and w_now.dtype == np.float64 else \
            np.asarray(w_now, dtype=np.float64).tobytes()
        if len(self._wall_is_hole) != len(w_now):
            if any(self._wall_is_hole):
                raise ValueError("Environment.walls was replaced in an environment with holes: use add_wall / add_hole, "
                                 "which keep track of which walls are hole edges")
            self._wall_is_hole = [False] * len(np.asarray(w_now).reshape(-1, 4))
        shape_key = (self.is_rectangular, self._n_boundary, self._wall_is_hole)
        if hit is not None:
            (src, bc, sc, asp, shp), env, wt = hit
            # fast path (every step): same geometry, wall array unchanged
            if src == w_bytes and bc == self.boundary_conditions and sc == self.scale and asp == self.aspect and \
                    shp[0] == shape_key[0] and shp[1] == shape_key[1] and shp[2] == shape_key[2]:
                return env, wt
        walls = np.ascontiguousarray(np.asarray(self.walls, dtype=np.float64).reshape(-1, 4))
        key = (w_bytes, self.boundary_conditions, self.scale, self.aspect,
               (shape_key[0], shape_key[1], list(shape_key[2])))
        if len(walls) > _lib.MAX_WALLS:
            raise ValueError(f"at most {_lib.MAX_WALLS} walls are supported on device, got {len(walls)}")
        wt = torch.from_numpy(walls if len(walls) else np.zeros((1, 4))).to(device)
        env = _lib.RiabEnv()
        for i in range(4):
            env.extent[i] = float(self.extent[i])
        env.scale = float(self.scale)
        env.periodic = 1 if self.boundary_conditions == "periodic" else 0
        env.n_walls = int(len(walls))
        env.walls = wt.data_ptr()
        env.polygon = 0 if self.is_rectangular else 1
        env.n_boundary = int(self._n_boundary)
        env.hole_mask = sum(1 << k for k, h in enumerate(self._wall_is_hole) if h)
        self._device_cache[slot] = (key, env, wt)
        return env, wt

    def op_env_args(self, device):
        """(walls tensor or None, env list, periodic) for the registered operators (ops.py: torch.ops.riab.*)."""
        env, wt = self.device_tables(device)
        e = [float(x) for x in self.extent] + [float(self.scale)]
        if env.polygon or env.hole_mask:
            mask = int(env.hole_mask)
            e += [float(env.polygon), float(env.n_boundary), float(mask & 0xFFFFFFFF), float(mask >> 32)]
        return (wt if env.n_walls else None), e, bool(env.periodic)

    def plot_environment(self, *a, **k):
        raise NotImplementedError("plotting is outside the accelerated path; use the reference package for figures")

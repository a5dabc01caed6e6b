"""`Agent` — batched, device-resident drop-in for the reference's Agent on the
random-motion path (reference ratinabox/Agent.py:17-1102).

One `Agent` object holds `n_agents` INDEPENDENT agents (new: the reference has one
agent per object and loops a Python list, README.md:236-241).  The public surface
is the reference's: `Agent(Environment, params)`, `update(dt, drift_velocity,
drift_to_random_strength_ratio, **kwargs)`, attributes `pos, velocity,
rotational_velocity, measured_velocity, measured_rotational_velocity,
head_direction, distance_travelled, distance_to_closest_wall, t, dt, history,
Neurons`, `get_history_arrays()`, `reset_history()`.  With `n_agents == 1` the
attribute shapes are the reference's (`pos (2,)`, ...); otherwise they carry a
leading agent axis (`pos (B,2)`, `history["pos"] (T,B,2)`).

All per-step arithmetic runs in the HIP kernel `riab_agent_step`
(csrc/riab_agent_kernel.h); this class owns the state tensor `[12, B]` (float64, HBM),
the trajectory history chunks `[T, 8, B]` (float32, HBM) and resolves parameters.
`simulate(T)` is the fused path: T steps per launch on one stream while the
firing-rate kernels of the previous chunk run on another."""
import copy
import operator
import os

import numpy as np
import torch

from . import _lib, utils
from ._history import DeviceHistory, HistoryView
from .plan import AutoStepper as _AutoStepper

_L = _lib


_NO_KWARGS = {}
_PLAIN = (int, float, str, bool, type(None), np.float64, np.float32, np.int64, np.int32, np.bool_)
# what a repeated plain simulate() depends on besides the populations and the geometry, read with ONE attrgetter call
_AGENT_FAST_ATTRS = ("save_history", "seed", "agent_id0", "dt", "_time_rate_kernel", "_timed_population",
                     "use_imported_trajectory", "DIRECT_NATIVE_CALL", "rotational_velocity_std",
                     "rotational_velocity_coherence_time", "speed_coherence_time", "speed_mean", "speed_std",
                     "wall_repel_strength", "wall_repel_distance", "thigmotaxis", "head_direction_smoothing_timescale")
_ENV_FAST_ATTRS = ("boundary_conditions", "scale", "aspect", "is_rectangular", "_n_boundary")
_agent_fast = operator.attrgetter(*_AGENT_FAST_ATTRS)
_env_fast = operator.attrgetter(*_ENV_FAST_ATTRS)
_ENV_RAW = getattr(os.environ, "_data", None)   # (os.environ's own mapping: bytes keys on POSIX; None elsewhere)
_dispatch_depth = torch._C._len_torch_dispatch_stack
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
from time import perf_counter as _perf_counter  # noqa: E402


class Agent:
    default_params = {
        "name": None,
        "dt": 0.05,
        "speed_coherence_time": 0.7,
        "speed_mean": 0.08,
        "speed_std": 0.08,
        "rotational_velocity_coherence_time": 0.08,
        "rotational_velocity_std": (120 * (np.pi / 180)),
        "head_direction_smoothing_timescale": 0.15,
        "thigmotaxis": 0.5,
        "wall_repel_distance": 0.1,
        "wall_repel_strength": 1.0,
        "save_history": True,
        # --- batched extension (not in the reference) ---
        "n_agents": 1,        # independent agents held by this object
        "device": "cuda",     # torch device of the state / history tensors
        "seed": 0,            # Philox key of the in-kernel noise (production mode)
        "agent_id0": 0,       # global id of agent 0 (multi-GPU shards; keys the RNG)
    }

    # simulate(): True = call riab_simulate through ctypes (the default: ~3 us less host time per call, which is 3 % of a
    # 20-step run); False = through torch.ops.riab.simulate_, which a dispatch mode / tracer sees — taken automatically
    # while a TorchDispatchMode is active.  User code that wants the run inside a compiled function calls the operator
    # itself (Agent.simulate_args).
    DIRECT_NATIVE_CALL = True
    AUTO_AFTER = 4   # plain update() calls in a row before the per-step loop is served from a native plan
    AUTO_AFTER_MAX = 1024  # ... after back-off: a stepper that served fewer than AUTO_KEEP steps before something
    AUTO_KEEP = 4          #     closed it doubles the wait for the next one (a loop that edits a weight every step
                           #     would otherwise rebuild a plan every five steps for nothing)

    def __init__(self, Environment, params={}):
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        utils.update_class_params(self, self.params, get_all_defaults=True)
        utils.check_params(self, params.keys())

        self.Environment = Environment
        self.agent_idx = len(self.Environment.Agents)
        if self.name is None:
            self.name = f"agent_{self.agent_idx}"
        self.Environment.add_agent(agent=self)

        self._explicit_rng = "seed" in params or "agent_id0" in params
        self.n_agents = int(self.n_agents)
        assert self.n_agents >= 1
        for other in self.Environment.Agents[:self.agent_idx]:
            lo, hi = int(other.agent_id0), int(other.agent_id0) + int(other.n_agents)
            if other.rng_seed == self.rng_seed and lo < int(self.agent_id0) + self.n_agents and int(self.agent_id0) < hi:
                import warnings
                warnings.warn(f"{self.name} and {other.name} share the RNG key {self.rng_seed} and overlapping agent ids: "
                              "they will draw IDENTICAL noise (give them different `seed`s or disjoint `agent_id0` ranges)")
        assert self.agent_id0 % 4 == 0, "agent_id0 must be a multiple of 4"
        self._B = self.n_agents
        self._Bp = (self.n_agents + 3) // 4 * 4  # kernels need a multiple of 4 on the agent axis
        self._device = torch.device(self.device)
        # (the raw handle of torch's current stream is asked for by device index: the agent's own device when it names one)
        self._device_index = self._device.index if self._device.index is not None else \
            (torch.cuda.current_device() if self._device.type == "cuda" and torch.cuda.is_available() else 0)

        self.Neurons = []
        self.prev_t = 0
        self.t = 0
        self.average_measured_speed = max(self.speed_mean, self.speed_std)
        self.use_imported_trajectory = False
        self._step_index = 0  # counts update() calls: the RNG counter
        self._times = []
        self._hist = DeviceHistory((_L.HIST_ROWS, self._Bp), torch.float32, self._device)
        self._diag = torch.zeros(4, dtype=torch.int32, device=self._device)
        self._streams = None
        self._scratch_row = None
        self._last_row = None   # newest fp32 history row; None when the state was edited from the host
        self._plan = None       # an active StepPlan (plan.py), if any
        self._auto_streak = 0   # consecutive plain update() calls (plan.AutoStepper engages after _auto_after)
        self._auto_after = self.AUTO_AFTER
        self._auto_enabled = os.environ.get("RIAB_NO_AUTO_PLAN") != "1"
        self._streamer = None   # native handle of the flag-coupled pipeline (created on first use)
        self._snap = None       # what the last plain native simulate() prepared (see _simulate_repeat)
        # simulate() calls served by each engine: "native" = riab_simulate, "plan" = a native step plan (populations that
        # read the float64 state), "chunks" = the Python-driven chunk pipeline (RIAB_NO_NATIVE=1: the tests' comparator)
        self.engine_runs = {"native": 0, "plan": 0, "chunks": 0}
        self._run_cache = None  # (population structs, their array, the RiabSimulate argument block, key) of the last call
        self._ctrl = None       # its control words on the device
        self._pipeline_unchecked = False
        self._host_clock = None       # a list: _simulate_repeat appends (time before, time after) its native call
        self._unchecked_runs = []     # native simulate() calls since the last pipeline check: what a recovery recomputes
        self._unchecked_lost = False  # ... more of them than are kept
        self._recovered_runs = 0      # diagnostics["pipeline_recovered"]
        self._timeouts_recovered = 0
        self._recover_warned = False
        self._step1_timeouts = 0           # waits of the one-launch step that gave up (plan.py: _settle_fused) ...
        self._step1_recovered_steps = 0    # ... and the steps whose fused rows were recomputed
        self._time_rate_kernel = False
        self._timed_population = None
        self._serial_warned = False

        self._state = torch.zeros((_L.STATE_ROWS, self._Bp), dtype=torch.float64, device=self._device)
        self.initialise_position_and_velocity()
        # measured velocity starts as the velocity, head direction as its unit vector (Agent.py:137-141)
        st = self._state
        st[_L.S_MVEL_X:_L.S_MVEL_Y + 1] = st[_L.S_VEL_X:_L.S_VEL_Y + 1]
        st[_L.S_MROT_VEL] = 0.0
        nrm = torch.sqrt(st[_L.S_VEL_X] ** 2 + st[_L.S_VEL_Y] ** 2)
        st[_L.S_HD_X] = st[_L.S_VEL_X] / nrm
        st[_L.S_HD_Y] = st[_L.S_VEL_Y] / nrm
        st[_L.S_DIST] = 0.0
        st[_L.S_DWALL] = float("inf")
        # (the version is read AFTER an active step plan has published its rows: a plan commits rows lazily)
        self.history = HistoryView(("t", "pos", "distance_travelled", "vel", "rot_vel", "head_direction"),
                                   self._materialise_history, lambda: (self._sync_plan(), self._hist.version)[1])

    @classmethod
    def get_all_default_params(cls, verbose=False):
        all_params = utils.collect_all_params(cls, dict_name="default_params")
        if verbose:
            import pprint
            pprint.pprint(all_params)
        return all_params

    @property
    def rng_seed(self):
        """Philox key of this Agent OBJECT's streams (motion noise, neuron noise, spikes).  The reference's agents
        all draw from the global np.random stream and are independent of each other; here every stream is a pure
        function of (key, global agent id, population, step), so a second Agent object of the same Environment
        with the same key and id range replays the first one's noise.  An Agent constructed WITHOUT an explicit
        `seed` / `agent_id0` therefore gets the object's index in `Environment.Agents` folded into its key
        (agent 0 keeps `seed` itself).  Explicit `seed` / `agent_id0` are taken as given — that is how shards of
        one logical population reproduce the same agents on any number of GPUs — and a warning is raised at
        construction when two Agents of an Environment then share a key with overlapping id ranges."""
        if self.agent_idx == 0 or self._explicit_rng:
            return int(self.seed)
        return (int(self.seed) + self.agent_idx * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

    # ---- initial state (Agent.py:523-535) ---------------------------------------------------
    def initialise_position_and_velocity(self):
        """Uniform positions in the box, uniform headings, speed = speed_mean.  Draws
        from np.random per agent in the reference's order (position x, y, then heading)."""
        B = self._B
        pos = np.zeros((self._Bp, 2))
        vel = np.zeros((self._Bp, 2))
        for i in range(B):
            pos[i] = self.Environment.sample_positions(n=1, method="random")[0]
            direction = np.random.uniform(0, 2 * np.pi)
            vel[i] = self.speed_mean * np.array([np.cos(direction), np.sin(direction)])
        pos[B:], vel[B:] = pos[0], vel[0]  # padding agents shadow agent 0
        self._upload(_L.S_POS_X, pos)
        self._upload(_L.S_VEL_X, vel)
        self._state[_L.S_ROT_VEL] = 0.0

    # ---- attribute access -------------------------------------------------------------------
    def _squeeze(self, a):
        return a[0] if self._B == 1 else a

    MAX_UNCHECKED_RUNS = 1024   # native runs remembered between two pipeline checks

    def _check_pipeline(self):
        """After a native simulate(): a wait of the flag-coupled pipeline that gave up (bounded spins: a producer starved
        for about a second — forward progress between two kernels is not something the device promises) leaves rate
        rows unwritten, in that call and in every later one until the flags are cleared.  Checked on the first host
        read that follows, which synchronises anyway.  The trajectory kernel does not wait for anybody outside its own
        workgroup, so its rows and the agent state are complete (verified on the progress words): the rates of every
        run since the last check are then RECOMPUTED from those rows with the populations' ordinary, stream-ordered
        kernels — the same functors on the same fp32 rows: the bits the pipeline would have written — counted in
        diagnostics["pipeline_recovered"], and the caller gets a warning instead of an exception.  What cannot be
        redone raises as before: a trajectory kernel that itself gave up, populations with additive OU noise (their
        noise state has advanced), more unchecked runs than are remembered, a population whose tuning parameters were
        edited between the run and this read.  (Replays of a captured graph are not remembered: their rows are not
        recoverable — a capture runs in strict mode, whose rate stage starts behind a gate and does not give up in practice.)"""
        if not self._pipeline_unchecked:
            return
        self._pipeline_unchecked = False
        w = self._ctrl[:4].cpu()
        runs, self._unchecked_runs = self._unchecked_runs, []
        lost, self._unchecked_lost = self._unchecked_lost, False
        if not (int(w[_L.CTRL_TIMEOUTS]) or int(w[_L.CTRL_ABORT])):
            return
        n = int(w[_L.CTRL_TIMEOUTS])
        # (a rate stage that gave up did not wait for the trajectory kernel's last publication, which is what orders
        # the caller's stream behind that kernel: wait for the whole device before the flags are cleared and the
        # caller goes on — state uploads, the next update(), recycled buffers)
        torch.cuda.synchronize(self._device)
        n_traj = (self._Bp + 63) // 64
        idx = torch.arange(n_traj, device=self._device)
        words = self._ctrl[_L.CTRL_PROGRESS + 32 * (idx // 4) + (idx % 4)].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        complete = bool(runs) and bool((words == (int(self._step_index) & 0xFFFFFFFF)).all())
        self._ctrl[_L.CTRL_TIMEOUTS] = 0
        self._ctrl[_L.CTRL_ABORT] = 0
        why = None
        if not complete:
            why = "the trajectory kernel itself did not finish"
        elif lost:
            why = f"more than {self.MAX_UNCHECKED_RUNS} runs since the last host read"
        elif any(N.noise_std != 0 for r in runs for N in r[5]):
            why = "a population with additive OU noise took part (its noise state has advanced)"
        else:
            for r in runs:
                for N, (tabs, lo, hi) in zip(r[5], r[7]):
                    N._auto_key()   # (brings the content-keyed tables up to date with the host arrays as they are NOW)
                    if N._table_cache.get("t") is not tabs or float(N.min_fr) != lo or float(N.max_fr) != hi:
                        why = (f"the parameters of {type(N).__name__} were edited between that run and this read: its rows "
                               "would be recomputed with other tables than the run used")
        if why is not None:
            raise _L.RiabError(f"the flag-coupled simulate() pipeline was aborted ({n} waits timed out) and its rows cannot "
                               f"be recomputed: {why}; the rows of that run are incomplete (RIAB_NO_NATIVE=1 selects the "
                               "Python-driven chunk pipeline)")
        stream = _L.current_stream()
        for traj_c, traj_s, n_steps, step0, dt, neurons, ats, _tables in runs:
            traj = traj_c[traj_s:traj_s + n_steps]
            outs = [N._rows_views(at, n_steps) for N, at in zip(neurons, ats)]
            self._sim_outs = dict(zip(neurons, outs))   # (FeedForwardLayers read their inputs' rows of the same piece)
            for t0 in range(0, n_steps, 1024):
                tc = min(1024, n_steps - t0)
                for N, out in zip(neurons, outs):
                    N._rates_from_trajectory(traj[t0:t0 + tc], out, t0, tc, step0 + t0, float(dt), stream=stream)
        torch.cuda.synchronize(self._device)
        self._recovered_runs += len(runs)
        self._timeouts_recovered += n
        if not self._recover_warned:
            self._recover_warned = True
            import warnings
            warnings.warn(f"the flag-coupled simulate() pipeline was aborted ({n} waits of its rate stage timed out: something "
                          f"kept the trajectory kernel off the device for about a second); the firing rates of the last "
                          f"{len(runs)} run(s) were recomputed from the complete trajectory with the stream-ordered kernels "
                          "(diagnostics['pipeline_recovered'])", RuntimeWarning)

    def _settle_plan(self):
        """Host reads (the accessors that copy to the host, which waits for the device anyway): has a wait of the
        one-launch step given up since the last look?  (plan.py _settle_fused: recovered, counted.)  The accessors that
        hand out DEVICE tensors do not look — that would synchronise a closed loop every step; the plan does when it
        closes."""
        if self._plan is not None:
            self._plan.settle()

    def _download(self, row, width):
        self._settle_plan()
        self._check_pipeline()
        a = self._state[row:row + width, :self._B].t().contiguous().cpu().numpy()
        return self._squeeze(a if width > 1 else a[:, 0])

    def _upload(self, row, value):
        """Host edit of a state attribute: one value (broadcast to every agent) or one row per agent."""
        if self._plan is not None and self._plan.__class__ is _AutoStepper:
            self._plan.close()  # (the populations must now read the edited float64 state, which the eager path does)
        self._last_row = None  # the fp32 row the rate kernels read no longer mirrors the state
        width = 2 if row in (_L.S_POS_X, _L.S_VEL_X, _L.S_MVEL_X, _L.S_HD_X) else 1
        v = np.asarray(value, dtype=np.float64).reshape(-1, width)
        full = np.broadcast_to(v[:1], (self._Bp, width)).copy()  # (padding lanes repeat agent 0)
        n = min(len(v), self._Bp)
        full[:n] = v[:n]
        self._state[row:row + width] = torch.from_numpy(np.ascontiguousarray(full.T)).to(self._device)

    pos = property(lambda s: s._download(_L.S_POS_X, 2), lambda s, v: s._upload(_L.S_POS_X, v))
    velocity = property(lambda s: s._download(_L.S_VEL_X, 2), lambda s, v: s._upload(_L.S_VEL_X, v))
    rotational_velocity = property(lambda s: s._download(_L.S_ROT_VEL, 1), lambda s, v: s._upload(_L.S_ROT_VEL, v))
    measured_velocity = property(lambda s: s._download(_L.S_MVEL_X, 2), lambda s, v: s._upload(_L.S_MVEL_X, v))
    measured_rotational_velocity = property(lambda s: s._download(_L.S_MROT_VEL, 1),
                                            lambda s, v: s._upload(_L.S_MROT_VEL, v))
    head_direction = property(lambda s: s._download(_L.S_HD_X, 2), lambda s, v: s._upload(_L.S_HD_X, v))
    distance_travelled = property(lambda s: s._download(_L.S_DIST, 1), lambda s, v: s._upload(_L.S_DIST, v))
    distance_to_closest_wall = property(lambda s: s._download(_L.S_DWALL, 1))

    @property
    def state_tensor(self):
        """Device state `[12, B_padded]` float64 (rows: include/riab_hip.h RIAB_S_*)."""
        return self._state

    @property
    def diagnostics(self):
        """Counters accumulated by the kernel: bounces, bounce-loop saturations,
        boundary conditions applied, zero-displacement steps."""
        self._settle_plan()
        d = self._diag.cpu().numpy()
        out = dict(bounces=int(d[0]), bounce_saturations=int(d[1]), boundary_conditions=int(d[2]),
                   zero_displacement=int(d[3]))
        out["step1_timeouts_recovered"] = self._step1_timeouts
        out["step1_recovered_steps"] = self._step1_recovered_steps
        if self._ctrl is not None:
            w = self._ctrl[:4].cpu()
            out["pipeline_timeouts"] = int(w[_L.CTRL_TIMEOUTS])  # waits of the flag-coupled pipeline that gave up: must be 0
            out["pipeline_recovered"] = self._recovered_runs     # runs whose rates were recomputed after such a wait
            out["pipeline_timeouts_recovered"] = self._timeouts_recovered
            # simulate() calls (>= 8 steps) whose trajectory kernel had FINISHED before the firing-rate stage began: the
            # two kernels are meant to run side by side on two hardware queues; results are right, the call is slower
            # (minus the calls whose second launch the HOST issued late — a descheduled thread: such a call finds every row
            # published too, without any queue being shared; riab_streamer_info 7)
            late = int(_L.lib.riab_streamer_info(self._streamer, 7)) if self._streamer is not None else 0
            out["pipeline_serialised"] = max(0, int(w[_L.CTRL_SERIALISED]) - max(0, late))
            # (a lone count is a hiccup of the device — about one call in a thousand starts its second kernel that late on
            # the shared boxes of this pool; queue sharing shows on EVERY call: warn when it is systematic)
            if out["pipeline_serialised"] >= 3 and out["pipeline_serialised"] * 20 >= self.engine_runs["native"] \
                    and not self._serial_warned:
                self._serial_warned = True
                import warnings
                warnings.warn(f"{out['pipeline_serialised']} simulate() call(s) ran their trajectory and firing-rate "
                              "kernels one after the other instead of side by side: the process's HIP streams share a "
                              "hardware queue (GPU_MAX_HW_QUEUES, many streams in flight, an RCCL communicator: "
                              "DESIGN.md 7).  Results are unaffected; short runs are up to 45 % slower.", RuntimeWarning)
        return out

    # ---- parameter resolution ---------------------------------------------------------------
    def _motion(self, dt, has_drift, ratio, kwargs):
        """Resolve one update's parameters into the ABI struct, reproducing which values
        the reference takes from kwargs and which from attributes (Agent.py:280-285,
        310, 340, 353-355, 375, 439, 489)."""
        key = None
        g = kwargs.get
        # (wall-heavy rooms: the broad phase of the step's wall loops, Environment.wall_grid — None below 13 walls)
        grid = self.Environment.wall_grid(self._device, float(g("wall_repel_distance", self.wall_repel_distance))) \
            if self._device.type == "cuda" else None
        if not kwargs:
            key = (dt, has_drift, ratio, self.rotational_velocity_std, self.rotational_velocity_coherence_time,
                   self.speed_coherence_time, self.speed_mean, self.speed_std, self.wall_repel_strength,
                   self.wall_repel_distance, self.thigmotaxis, self.head_direction_smoothing_timescale)
            hit = getattr(self, "_motion_cache", None)
            if hit is not None and hit[0] == key and hit[2] is grid:
                return hit[1]
        m = _L.RiabMotion()
        rot_std = g("rotational_velocity_std", self.rotational_velocity_std)
        rot_tau = g("rotational_velocity_coherence_time", self.rotational_velocity_coherence_time)
        spd_tau = g("speed_coherence_time", self.speed_coherence_time)
        m.dt = float(dt)
        m.rot_theta_kw = 1 / rot_tau
        m.rot_sigma_kw = float(np.sqrt((2 * rot_std ** 2) / (rot_tau * dt)))
        m.rot_drift_kw = float(g("rotational_velocity_drift", 0))
        m.speed_theta_kw = 1 / spd_tau
        m.speed_sigma_kw = float(np.sqrt((2 * 1 ** 2) / (spd_tau * dt)))
        m.speed_mean_kw = float(g("speed_mean", self.speed_mean))
        m.speed_mean = float(self.speed_mean)
        m.speed_std_is_zero = 1 if self.speed_std == 0 else 0
        m.has_drift = 1 if has_drift else 0
        m.drift_theta = 1 / (self.speed_coherence_time / ratio) if has_drift else 0.0
        m.wall_repel_strength_kw = float(g("wall_repel_strength", self.wall_repel_strength))
        m.wall_repel_distance_kw = float(g("wall_repel_distance", self.wall_repel_distance))
        m.thigmotaxis_kw = float(g("thigmotaxis", self.thigmotaxis))
        m.hd_tau = float(self.head_direction_smoothing_timescale)
        if grid is not None:
            m.wall_grid, m.wall_grid_n, m.wall_grid_wd, m.wall_grid_lmax = grid[0].data_ptr(), grid[1], grid[2], grid[3]
            m._keep_grid = grid
        if key is not None:
            self._motion_cache = (key, m, grid)
        return m

    def _motion_key_now(self, dt, has_drift=False, ratio=1):
        """The attribute values `_motion` resolves an update() without per-call overrides from (its cache key)."""
        return (dt, has_drift, ratio, self.rotational_velocity_std, self.rotational_velocity_coherence_time,
                self.speed_coherence_time, self.speed_mean, self.speed_std, self.wall_repel_strength,
                self.wall_repel_distance, self.thigmotaxis, self.head_direction_smoothing_timescale)

    def _as_device_f64(self, x, rows):
        """array-like `(B, rows)` / `(rows,)` / tensor `[rows, B]` -> device float64 `[rows, Bp]`."""
        if torch.is_tensor(x):  # stays on the device: closed-loop callers pass policy outputs directly
            t = x.to(self._device, torch.float64)
            if t.shape == (rows, self._Bp):
                return t.contiguous()
            if t.shape == (rows,):
                return t.reshape(rows, 1).expand(rows, self._Bp).contiguous()
            if t.shape == (self._B, rows):
                t = t.t()
            assert t.shape == (rows, self._B), f"expected ({self._B},{rows}) or ({rows},{self._B}), got {tuple(x.shape)}"
            if self._Bp != self._B:
                t = torch.cat((t, t[:, :1].expand(rows, self._Bp - self._B)), dim=1)
            return t.contiguous()
        a = np.asarray(x, dtype=np.float64)
        a = np.broadcast_to(a.reshape(-1, rows), (self._B, rows)) if a.size == rows else a.reshape(self._B, rows)
        full = np.empty((rows, self._Bp))
        full[:, :self._B] = a.T
        full[:, self._B:] = a[0][:, None]
        return torch.from_numpy(full).to(self._device)

    # ---- one step -----------------------------------------------------------------------------
    def update(self, dt=None, drift_velocity=None, drift_to_random_strength_ratio=1, **kwargs):
        """One motion step for every agent (reference Agent.update, Agent.py:160-242).

        kwargs: the reference's per-call overrides (speed_mean, thigmotaxis, ...), plus
        `noise=` — explicit standard normals `(2, B)` / `(B, 2)` [rotation OU, speed OU]
        instead of the in-kernel Philox draws (parity mode)."""
        # ---- the unchanged reference loop, served natively (plan.AutoStepper) ----
        if (dt is None or dt == self.dt) and not kwargs and self._auto_enabled and \
                (not self.use_imported_trajectory or (self.interpolate and drift_velocity is None)):
            st = self._plan
            if st is not None and st.__class__ is _AutoStepper:
                if st.step_agent(drift_velocity, drift_to_random_strength_ratio):
                    return
                st.close()
            elif st is None:
                self._auto_streak += 1
                if self._auto_streak > self._auto_after and self._device.type == "cuda":
                    try:
                        if _AutoStepper(self).step_agent(drift_velocity, drift_to_random_strength_ratio):
                            return
                    except NotImplementedError:      # a population a plan cannot hold: stay eager
                        self._auto_enabled = False
                        if self._plan is not None:
                            self._plan.close()
        else:
            self._auto_streak = 0
        forced = kwargs.pop("forced_next_position", None)
        if forced is not None:
            # Agent._update_position_to_forced_next_position (Agent.py:244-253): overrides everything else
            forced = np.asarray(forced, dtype=np.float64)
            assert forced.shape in ((2,), (self._B, 2)), "forced_next_position must have shape (2,) or (n_agents, 2)"
            self._advance(1, dt, None, 1, kwargs, forced=self._as_device_f64(forced, 2).unsqueeze(0))
        elif self.use_imported_trajectory:
            self._advance_imported(1, dt, kwargs)
        else:
            self._advance(1, dt, drift_velocity, drift_to_random_strength_ratio, kwargs)

    def _advance(self, T, dt, drift_velocity, ratio, kwargs, hist_view=None, stream=None, z_out=None,
                 forced=None):
        if self._plan is not None:
            self._plan.close()  # eager stepping resumes: the plan's cursors would go stale
        dt = dt or self.dt
        self.dt = dt
        noise = kwargs.pop("noise", None) if "noise" in kwargs else None
        resample = kwargs.pop("resample_positions", None) if "resample_positions" in kwargs else None
        has_drift = drift_velocity is not None
        m = self._motion(dt, has_drift, ratio, kwargs)
        env, _walls = self.Environment.device_tables(self._device)
        drift = self._as_device_f64(drift_velocity, 2) if has_drift else None
        z = self._noise_tensor(noise, T) if noise is not None else None
        rs = self._resample_tensor(resample, T) if resample is not None else None
        if hist_view is None and self.save_history:
            hist_view = self._hist.reserve(T)
        if hist_view is None:
            # not saving: the kernel still leaves the step's fp32 row (what the rate kernels read)
            if self._scratch_row is None or self._scratch_row.shape[0] < T:
                self._scratch_row = torch.empty((T, _L.HIST_ROWS, self._Bp), dtype=torch.float32, device=self._device)
            hist_view = self._scratch_row[:T]
        if stream is None and self._device.type == "cuda":
            # the eager per-step path goes through the registered operator (ops.py): torch.ops.riab.agent_step_
            from . import ops
            w_op, e_op, periodic = self.Environment.op_env_args(self._device)
            torch.ops.riab.agent_step_(self._state, hist_view, self._diag, w_op, e_op, periodic, ops.motion_list(m), drift, z,
                                       z_out, forced, rs, ops.seed_arg(self.rng_seed), int(self._step_index),
                                       int(self.agent_id0), int(T))
        else:
            s = stream if stream is not None else _L.current_stream()
            rc = _L.lib.riab_agent_step(env, m, _L.ptr(self._state), self._Bp, int(self.agent_id0), _L.ptr(drift),
                                        _L.ptr(z), _L.ptr(z_out), _L.ptr(forced), _L.ptr(rs), int(self.rng_seed),
                                        int(self._step_index), int(T),
                                        _L.ptr(hist_view), _L.ptr(self._diag), s)
            _L.check(rc, "riab_agent_step")
        self._keep = (drift, z, _walls, hist_view, forced, rs)  # keep operands alive until the stream is done
        self._last_row = hist_view[T - 1]            # fp32 [8, Bp]: positions / head directions of the newest step
        for _ in range(T):
            self.prev_t = self.t
            self.t += dt
            if self.save_history:
                self._times.append(self.t)
        self._step_index += T
        return hist_view

    # ---- fused path ----------------------------------------------------------------------------
    def simulate(self, n_steps, dt=None, drift_velocity=None, drift_to_random_strength_ratio=1, chunk=128,
                 neurons=None, noise=None, **kwargs):
        """`n_steps` x (Agent.update(); N.update() for N in neurons) without returning to
        Python between steps (new; the open-loop workload of SURVEY §7.3-1).

        One native call (`riab_simulate`): the trajectory kernel publishes its history rows through
        flags in device memory and the firing-rate kernels of the populations consume them while it runs — one gated
        rate kernel for a single store-bound population, every population's kernels per chunk of rows behind a gate
        for any other set; any batch size, explicit `noise=` normals, per-call motion kwargs and imported trajectories
        included.  Populations that read the agent's float64 state (VelocityCells) advance through a native step plan.
        AgentVectorCells and recurrent layers have no open-loop run (NotImplementedError: they advance through
        update()).  The older Python-driven pipeline — trajectory chunks of `chunk` steps on one HIP stream, the rate
        kernels of each finished chunk on a second one — is kept as the comparator of the tests (`RIAB_NO_NATIVE=1`)
        only; `Agent.engine_runs` counts which engine served each call.  Histories land in HBM
        (`save_history=True`) exactly as `n_steps` calls of update() would have left them.  Returns the trajectory
        history tensor of this call `[n_steps, 8, B_padded]` (device; None when populations that read the agent's state
        are advanced through a step plan and the agent keeps no history).  The tensor is returned while the kernels
        still run and is not validated: a wait of the flag-coupled pipeline that gave up (bounded spins: a producer
        starved for about a second) is reported by the next host read, by get_history_tensor() / get_history_tensors()
        / firingrate_tensor, and counted in diagnostics["pipeline_timeouts"]."""
        if self._snap is not None and neurons is None and noise is None and drift_velocity is None and not kwargs and \
                self._plan is None and (dt is None or dt == self._snap["dt"]) and self.dt == self._snap["dt"] and n_steps > 0:
            self._auto_streak = 0
            traj = self._simulate_repeat(int(n_steps))
            if traj is not None:
                return traj
        neurons = list(self.Neurons if neurons is None else neurons)
        if self._plan is not None:
            self._plan.close()  # (before any history row is reserved: a plan's pending rows are not committed yet)
        self._auto_streak = 0
        if int(n_steps) <= 0:   # nothing to do: an empty trajectory, no row reserved, no launch
            assert int(n_steps) == 0, "n_steps must not be negative"
            return torch.empty((0, _L.HIST_ROWS, self._Bp), dtype=torch.float32, device=self._device)
        if any(getattr(N, "_reads_agent_state", False) for N in neurons):
            # VelocityCells / SpeedCell read the float64 velocity STATE, which no history row keeps: such a run is
            # n_steps native plan steps (one motion launch + the rate launches per step, looped in C++)
            if noise is not None or kwargs or self.use_imported_trajectory:
                raise NotImplementedError("simulate() with populations that read the agent's state (VelocityCells, "
                                          "SpeedCell) takes neither explicit noise, per-call motion kwargs nor an "
                                          "imported trajectory: use update()")
            return self._simulate_by_plan(int(n_steps), dt or self.dt, drift_velocity, drift_to_random_strength_ratio,
                                          neurons)
        traj = self._simulate_native(int(n_steps), dt or self.dt, drift_velocity, drift_to_random_strength_ratio, neurons,
                                     noise, kwargs)
        if traj is not None:
            return traj
        self.engine_runs["chunks"] += 1
        if self._streams is None:
            self._streams = self._make_streams()
        s_traj, s_rate = self._streams
        cur = torch.cuda.current_stream(self._device)
        n_steps = int(n_steps)
        dt = dt or self.dt
        if self.save_history:
            traj = self._hist.reserve(n_steps)
        else:
            traj = torch.empty((n_steps, _L.HIST_ROWS, self._Bp), dtype=torch.float32, device=self._device)
        outs = [N._reserve_rows(n_steps, ring=min(chunk, n_steps)) for N in neurons]
        self._sim_outs = dict(zip(neurons, outs))  # FeedForwardLayers read their inputs' rows of the same chunk
        ready = torch.cuda.Event()
        ready.record(cur)
        s_traj.wait_event(ready)
        s_rate.wait_event(ready)
        z_all = None
        if noise is not None:
            z_all = noise
        step0 = self._step_index
        t_start = self.t
        t0 = 0
        for tc in self._chunk_schedule(n_steps, chunk):
            view = traj[t0:t0 + tc]
            kw = dict(kwargs)
            if z_all is not None:
                kw["noise"] = z_all[t0:t0 + tc]
            with torch.cuda.stream(s_traj):
                if self.use_imported_trajectory:
                    self._advance_imported(tc, dt, kw, hist_view=view, stream=_lib.C.c_void_p(s_traj.cuda_stream))
                else:
                    self._advance(tc, dt, drift_velocity, drift_to_random_strength_ratio, kw, hist_view=view,
                                  stream=_lib.C.c_void_p(s_traj.cuda_stream))
                ev = torch.cuda.Event()
                ev.record(s_traj)
            s_rate.wait_event(ev)
            with torch.cuda.stream(s_rate):
                for N, out in zip(neurons, outs):
                    N._rates_from_trajectory(view, out, t0, tc, step0 + t0, float(dt),
                                             stream=_lib.C.c_void_p(s_rate.cuda_stream))
            t0 += tc
        done_a, done_b = torch.cuda.Event(), torch.cuda.Event()
        done_a.record(s_traj)
        done_b.record(s_rate)
        cur.wait_event(done_a)
        cur.wait_event(done_b)
        if self.save_history:
            times = self._times[-n_steps:]
        else:  # (the agent keeps no history, its populations may: the clock values of these steps, `t += dt` repeated)
            t, times = t_start, []
            for _ in range(n_steps):
                t += dt
                times.append(t)
        for N, out in zip(neurons, outs):
            N._finish_rows(out, n_steps, times)
        return traj

    def _simulate_by_plan(self, n_steps, dt, drift_velocity, ratio, neurons):
        """simulate() as `n_steps` steps of a native step plan (plan.StepPlan): the same kernels, arguments and
        counters as `n_steps` x (update(); N.update()), so bit-identical to that loop."""
        from .plan import StepPlan
        plan = StepPlan(self, neurons, capacity=max(1, min(n_steps, 1024)))
        self.engine_runs["plan"] += 1
        done = 0
        while done < n_steps:
            n = min(plan.capacity, n_steps - done)
            plan.step(n, drift_velocity=drift_velocity, drift_to_random_strength_ratio=ratio, dt=dt)
            done += n
        plan.close()
        # (an agent that keeps no history has only its newest row in this mode: nothing to return)
        return self._hist.stack()[len(self._hist) - n_steps:] if self.save_history else None

    def _make_streamer(self):
        """The native handle of the flag-coupled pipeline and its control words.  The words are zeroed ONCE, and the
        zero-fill (an asynchronous kernel on the current stream) is waited for here: the gate and rate kernels run on
        the streamer's own non-blocking stream, which nothing else orders behind that fill — without the wait they
        could read a recycled block's old contents as started counts / progress / abort flags."""
        h = _L.lib.riab_streamer_create()
        if not h:
            raise _L.RiabError("riab_streamer_create failed")
        self._streamer = _L.C.c_void_p(h)
        self._ctrl = torch.zeros(_L.ctrl_words(self._Bp), dtype=torch.int32, device=self._device)
        torch.cuda.current_stream(self._device).synchronize()
        # residency of the two kernels (riab_hip.h "Residency"): the default is RIAB_GATE_RESERVED — a short call of one
        # store-bound population from an idle stream is two launches, the rate kernel in its reserving shape; A/B:
        # RIAB_GATE=always (the one-wave started gate in front of every rate stage: three launches), RIAB_GATE=when_busy
        # or RIAB_GATE_WHEN_BUSY=1 (no gate and no reservation while the stream is idle: this process owns the device)
        gate = (_L.env("RIAB_GATE") or ("when_busy" if _L.env("RIAB_GATE_WHEN_BUSY") == "1" else "")).lower()
        if gate:
            _L.check(_L.lib.riab_streamer_configure(self._streamer, _L.STREAMER_OPT_GATE,
                                                    {"always": _L.GATE_ALWAYS, "when_busy": _L.GATE_WHEN_BUSY,
                                                     "reserved": _L.GATE_RESERVED}[gate]), "riab_streamer_configure")
        if _L.env("RIAB_FORM_STEP_NS"):   # (A/B of the populations / chunk form: the trajectory step the choice compares, in ns)
            _L.check(_L.lib.riab_streamer_configure(self._streamer, _L.STREAMER_OPT_STEP_NS, int(_L.env("RIAB_FORM_STEP_NS"))),
                     "riab_streamer_configure")
        if _L.env("RIAB_SIDE_STREAM"):      # (1: default-priority second stream; 2: the caller's stream — serial, diagnostics)
            _L.check(_L.lib.riab_streamer_configure(self._streamer, _L.STREAMER_OPT_SIDE_STREAM, int(_L.env("RIAB_SIDE_STREAM"))),
                     "riab_streamer_configure")
        if _L.env("RIAB_NO_FUSED") == "1":  # A/B comparisons: always the chunk form of the rate stage
            _L.lib.riab_streamer_configure(self._streamer, _L.STREAMER_OPT_POLL_MAX, 0)
        elif _L.env("RIAB_POLL_MAX"):       # (... or the one-kernel form up to this many steps)
            _L.check(_L.lib.riab_streamer_configure(self._streamer, _L.STREAMER_OPT_POLL_MAX, int(_L.env("RIAB_POLL_MAX"))),
                     "riab_streamer_configure")
        if _L.env("RIAB_HEAD_ROWS"):        # (rows of a long run that the row-following kernel serves: 65535 = all)
            _L.check(_L.lib.riab_streamer_configure(self._streamer, _L.STREAMER_OPT_HEAD_ROWS, int(_L.env("RIAB_HEAD_ROWS"))),
                     "riab_streamer_configure")
        # Everything a later call might have to set up, now (riab_hip.h "Two modes"): the strict mode's own stream, the
        # screening of the default mode's second stream, the timing events.  A call on a stream that is being captured
        # (torch.cuda.graph around Agent.simulate() / torch.ops.riab.simulate_) then runs in the strict mode by itself.
        if not torch.cuda.is_current_stream_capturing():
            _L.check(_L.lib.riab_streamer_warmup(self._streamer, _L.current_stream()), "riab_streamer_warmup")
        if _L.env("RIAB_STRICT") == "1":
            self.pipeline_mode(strict=True)

    def pipeline_mode(self, strict=None):
        """`strict=True`: native simulate() calls of this agent take the mode of riab_simulate that conforms to the C ABI's
        contract (include/riab_hip.h "Two modes": nothing allocated, synchronised, queried or shared; capturable; two
        more launches per call); `False`: the mode tuned for one short call per synchronisation; `None` (the default of
        a new Agent): strict for runs of more than 256 steps, where it costs nothing measurable, the short-call mode below.
        Calls on a stream that is being captured are strict whatever this says.  Same rows either way."""
        if self._streamer is None:
            self._make_streamer()
        _L.check(_L.lib.riab_streamer_configure(self._streamer, _L.STREAMER_OPT_STRICT, 2 if strict is None else 1 if strict else 0),
                 "riab_streamer_configure")

    # ---- the open-loop run as ONE native call (riab_simulate) -----------------------------------------------------------
    def _simulate_native(self, n_steps, dt, drift_velocity, ratio, neurons, noise=None, kwargs=None):
        """`riab_simulate` (csrc/riab_simulate.hip): the trajectory kernel publishes its rows through
        flags in device memory and the firing-rate stage consumes them on the caller's stream while it runs — one
        row-following rate kernel for a single store-bound population (PlaceCells / GridCells / HeadDirectionCells
        without OU noise, whole 256-agent groups), the same kernel for the largest such population followed by the
        others' ordinary kernels when its stores keep pace with the trajectory, every population's ordinary kernel per
        chunk of rows behind a progress gate otherwise (`last_rate_stage_form()`).  Any batch size, explicit `noise=` normals, per-call motion kwargs,
        `resample_positions=` and imported trajectories (the forced-position kernel followed by the populations'
        kernels) are covered: a 20-step run is ~70 us of GPU time, so everything that is not needed to issue the call
        — views, clocks, mirrors — happens AFTER it, while the kernels run.  Populations that do not save their
        history stream through a ring of rows: one native call per ring length (inside a call the kernels of a chunk
        run while the next chunk's rows are produced, so the rows of one call must not alias; calls are ordered by
        the stream).  Returns None — nothing reserved, nothing launched — for what it does not cover: populations
        that cannot be recorded (AgentVectorCells, recurrent FeedForwardLayers), float32 motion; `RIAB_NO_NATIVE=1`
        switches it off (A/B comparisons: the Python-driven chunk pipeline gives identical results)."""
        if n_steps <= 0 or _L.env("RIAB_NO_NATIVE") == "1":
            return None
        for N in neurons:
            if N.Agent is not self:
                return None
        self.dt = dt  # (the populations' OU-noise constants are those of this dt, as in update(dt=...))
        if len(neurons) == 1:
            try:
                structs = [neurons[0]._population()]
            except NotImplementedError:
                return None
        else:
            index, structs = {}, []
            try:
                for N in neurons:
                    structs.append(N._population(index))
                    index[N] = len(index)
            except NotImplementedError:
                return None
        if self._streamer is None:
            self._make_streamer()
        kwargs = dict(kwargs) if kwargs else {}
        resample = kwargs.pop("resample_positions", None)
        has_drift = drift_velocity is not None
        m = self._motion(dt, has_drift, ratio, kwargs)
        env, _walls = self.Environment.device_tables(self._device)
        drift = self._as_device_f64(drift_velocity, 2) if has_drift else None
        Bp, B = self._Bp, self._B
        z = rs = forced = None
        if self.use_imported_trajectory:
            if noise is not None or has_drift or not self.interpolate:
                return None
            # the positions of the coming steps on the clock the loop would have had (repeated `t += dt`)
            ts = np.cumsum(np.concatenate(([self.t + dt], np.full(n_steps - 1, dt))))
            pos = np.broadcast_to(self.pos_interp(ts % max(self.t_interp)), (n_steps, B, 2))
            full = np.empty((n_steps, 2, Bp))
            full[:, :, :B] = np.transpose(pos, (0, 2, 1))
            full[:, :, B:] = full[:, :, :1]
            forced = torch.from_numpy(full).to(self._device)
        if noise is not None:
            z = self._noise_tensor(noise, n_steps)
        if resample is not None:
            rs = self._resample_tensor(resample, n_steps)
        # rows: (tensor, first row) pairs now, views later
        if self.save_history:
            traj_c, traj_s = self._hist.reserve_at(n_steps)
        else:
            traj_c, traj_s = torch.empty((n_steps, _L.HIST_ROWS, Bp), dtype=torch.float32, device=self._device), 0
        ats = [N._reserve_rows_at(n_steps, 128) for N in neurons]   # (rings of 256 rows where rates are not saved)
        piece = n_steps
        for at in ats:
            if at[4] is not None and at[4] < piece:
                piece = at[4]
        npop = len(neurons)
        timed = -1
        if self._time_rate_kernel:
            tp = getattr(self, "_timed_population", None)
            timed = neurons.index(tp) if tp in neurons else 0
        # The array of population structs and the argument block are kept from call to call, with everything that does
        # not change between two calls of a loop already in them (the structs themselves are cached by the populations
        # while their tables do not change; `env` / `m` are cached objects while geometry / parameters do not change).
        key = (env, m, self.agent_id0, self.seed, self.agent_idx, timed, self._time_rate_kernel, drift is None)
        cache = self._run_cache
        if cache is None or cache[3] != key or len(cache[0]) != npop or any(x is not y for x, y in zip(cache[0], structs)):
            arr = (_L.RiabPopulation * npop)()
            for i, pop in enumerate(structs):
                _L.C.memmove(_L.C.byref(arr, i * _L.POP_SIZE), _L.C.byref(pop), _L.POP_SIZE)
            run = _L.RiabSimulate()
            run.pops, run.n_pops = (_L.C.cast(arr, _L.C.POINTER(_L.RiabPopulation)) if npop else None), npop
            run.env, run.motion = _L.C.pointer(env), _L.C.pointer(m)
            run.state, run.B, run.agent_id0 = self._state.data_ptr(), Bp, int(self.agent_id0)
            run.seed = int(self.rng_seed)
            run.diag, run.ctrl, run.timed_pop = self._diag.data_ptr(), self._ctrl.data_ptr(), timed
            run.timing_mode = 1 if self._time_rate_kernel == "events" else 0
            cache = self._run_cache = (list(structs), arr, run, key, _L.C.byref(run))
        arr, run, byref = cache[1], cache[2], cache[4]
        run.watch, run.n_watch = None, 0   # (this road has just checked every table by content itself)
        run.drift = drift.data_ptr() if drift is not None else None
        bases = []
        for N, at in zip(neurons, ats):
            n = int(N.n)
            fr_row, sp_row = (n * Bp * 4, n * Bp) if at[4] is None else (0, 0)
            bases.append((at[0].data_ptr() + at[1] * n * Bp * 4, None if at[2] is None else at[2].data_ptr() + at[3] * n * Bp,
                          fr_row, sp_row))
        traj_row = _L.HIST_ROWS * Bp * 4
        traj_p = traj_c.data_ptr() + traj_s * traj_row
        step = int(self._step_index)
        # the call goes through the registered operator (torch.ops.riab.simulate_: traceable, composes with torch code
        # on the same stream); `Agent.DIRECT_NATIVE_CALL = True` calls the ABI entry point directly (~3 us less host
        # time per call, nothing a tracer can see)
        via_op = not self.DIRECT_NATIVE_CALL or torch._C._len_torch_dispatch_stack() > 0
        if via_op:
            from . import ops  # noqa: F401  (registers torch.ops.riab.*)
            op = torch.ops.riab.simulate_
            handle_i, run_i = self._streamer.value, _L.C.addressof(run)
            op_rates = [at[0] for at in ats]
            op_spikes = [at[2] for at in ats if at[2] is not None]
        call, handle, stream = _L.lib.riab_simulate, self._streamer, _L.current_stream()
        noise_row = 2 * Bp * 8
        t0 = tc = 0
        while t0 < n_steps:
            tc = min(piece, n_steps - t0)
            for i, (fr_p, sp_p, fr_row, sp_row) in enumerate(bases):
                q = arr[i]
                q.rates_base = fr_p + t0 * fr_row
                q.spikes_base = None if sp_p is None else sp_p + t0 * sp_row
                q.capacity_rows = tc
            run.step0, run.T, run.hist = step + t0, tc, traj_p + t0 * traj_row
            run.noise = z.data_ptr() + t0 * noise_row if z is not None else None
            run.forced_pos = forced.data_ptr() + t0 * noise_row if forced is not None else None
            run.resample_pos = rs.data_ptr() + t0 * noise_row if rs is not None else None
            if via_op:
                try:
                    op(self._state, traj_c, op_rates, op_spikes, self._ctrl, self._diag, handle_i, run_i, traj_s + t0,
                       [at[1] + (t0 if at[4] is None else 0) for at in ats],
                       [at[3] + t0 for at in ats if at[2] is not None])
                    rc = 0
                except _L.RiabError as e:
                    rc = getattr(e, "code", _L.EINVAL)
            else:
                rc = call(handle, byref, stream)
            if rc == _L.EUNSUPPORTED and t0 == 0:  # (nothing was launched; the chunk pipeline reserves its own rows)
                if self.save_history:
                    self._hist.unreserve(n_steps)
                for N, at in zip(neurons, ats):
                    if at[4] is None:
                        N._hist_fr.unreserve(n_steps)
                        if at[2] is not None:
                            N._hist_sp.unreserve(n_steps)
                return None
            if rc:
                self._native_failed(rc, t0, tc, dt)
            t0 += tc
        traj = self._after_native(n_steps, dt, traj_c, traj_s, neurons, ats, tc,
                                  (drift, _walls, arr, structs, traj_c, z, rs, forced, env, m))
        # a plain call (Philox noise, no drift, no per-call parameters, full histories, the direct ABI call): the next
        # one with the same arguments takes the short road in simulate() — same checks, a third of the Python
        plain = z is None and rs is None and forced is None and drift is None and not kwargs and not via_op and \
            self.save_history and all(at[4] is None for at in ats)
        self._snap = None
        run.watch, run.n_watch = None, 0
        # (a population that reads another's rows — a FeedForwardLayer — describes itself through the index of the
        # recorded populations, which the short road does not rebuild: such sets take the general road every time)
        if plain and not any(getattr(N, "inputs", None) for N in neurons):
            self._snap = dict(neurons=list(neurons), structs=structs, arr=arr, run=run, byref=byref, env=env, m=m, dt=dt,
                              seed=self.seed, a0=self.agent_id0, timing=self._time_rate_kernel,
                              timed=getattr(self, "_timed_population", None),
                              pops=[(N, int(N.n), at[2] is not None) for N, at in zip(neurons, ats)],
                              fast=self._fast_record(neurons))
        return traj

    def _fast_record(self, neurons):
        """What lets the NEXT plain simulate() skip re-deriving every table key by content (`_simulate_repeat`): for
        populations whose tables come straight from float64 arrays held in attributes (PlaceCells, GridCells,
        HeadDirectionCells), those array OBJECTS (a replaced attribute is seen by identity), a snapshot of their bytes
        (an edit in place — `PCs.place_cell_centres[-1] = ...`, reference tests/test_advanced.py:59 — is seen by the
        library's memcmp against the snapshot, RiabSimulate.watch, before it launches anything), the scalar
        parameters by value; the same for the Environment's wall table.  None when anything does not fit (the repeat
        then re-examines every key the slower way)."""
        if _L.env("RIAB_NO_FAST_REPEAT") == "1":
            return None
        from .Neurons import FAST_REPEAT_TYPES
        keep, pops = [], []

        def watchable(a):
            return type(a) is np.ndarray and a.dtype == np.float64 and a.flags.c_contiguous

        for N in neurons:
            if type(N) not in FAST_REPEAT_TYPES:
                return None
            arrs = [(name, getattr(N, name, None)) for name in N._watch_arrays]
            getter = N._fast_getter()
            sc = getter(N)
            if not all(watchable(a) for _, a in arrs) or not all(type(v) in _PLAIN for v in sc):
                return None
            pops.append((N, arrs, getter, sc))
            keep.extend(a for _, a in arrs)
        Env = self.Environment
        walls = Env.walls
        if not watchable(walls):
            return None
        keep.append(walls)
        snaps = [a.copy() for a in keep]
        watch = (_L.RiabWatch * len(keep))()
        for w, a, c in zip(watch, keep, snaps):
            w.live, w.snapshot, w.bytes = a.ctypes.data, c.ctypes.data, a.nbytes
        ag = _agent_fast(self)
        if not all(type(v) in _PLAIN for v in ag if v is not self._timed_population):
            return None
        # (the argument block holds raw pointers into the populations' device tables and the wall table: snapshot-equal
        # content must imply that those tensors still exist, whatever rebuilt the populations' caches in between)
        tables = [list(getattr(N, "_plan_tables", None) or ()) for N in neurons]
        return dict(pops=pops, walls=walls, env=_env_fast(Env), holes=list(Env._wall_is_hole), ag=ag, watch=watch,
                    n_watch=len(keep), _keep=(keep, snaps, tables, Env.device_tables(self._device)))

    def _after_native(self, n_steps, dt, traj_c, traj_s, neurons, ats, tc, keep):
        """The kernels of a native run are in flight: now the views and the Python-side mirrors (clocks, step index,
        the populations' newest rows)."""
        outs = [N._rows_views(at, n_steps) for N, at in zip(neurons, ats)]
        self.engine_runs["native"] += 1
        self._pipeline_unchecked = True
        # (what _check_pipeline needs to recompute this run's rates should a wait of its rate stage have given up)
        runs = self._unchecked_runs
        if len(runs) >= self.MAX_UNCHECKED_RUNS:
            del runs[0]
            self._unchecked_lost = True
        # (+ what each population's kernels read, as it was for THIS run: its device tables — content-keyed: a new object
        # when a tuning array was edited — and its rate scaling; a recovery with other parameters would not rewrite the
        # bits the run would have written)
        runs.append((traj_c, traj_s, n_steps, self._step_index, dt, neurons, ats,
                     tuple((N._table_cache.get("t"), float(N.min_fr), float(N.max_fr)) for N in neurons)))
        traj = traj_c[traj_s:traj_s + n_steps]
        self._keep = (keep, outs)
        self._last_row = traj[n_steps - 1]
        self._last_fused_units = self._Bp * tc  # agent-steps of the call `last_rate_kernel_ms` refers to (the last piece)
        t, times = self.t, []
        for _ in range(n_steps):  # (the reference's clock: repeated `t += dt`, not t0 + i*dt)
            self.prev_t = t
            t += dt
            times.append(t)
        self.t = t
        if self.save_history:
            self._times.extend(times)
        self._step_index += n_steps
        for N, out in zip(neurons, outs):
            if out["ring"] is not None:
                out["last"] = out["fr"][tc - 1]   # (every piece starts at the ring's first row)
            N._finish_rows(out, n_steps, times)
        return traj

    def _simulate_repeat(self, n_steps):
        """simulate(n_steps) again with what the previous plain native call prepared (`self._snap`): every input of
        that preparation is re-examined, and anything that has changed sends the call down the general road (None).
        Two ways of examining: with a `fast` record (`_fast_record`: PlaceCells / GridCells / HeadDirectionCells) the
        attributes by identity and value here and the CONTENT of their arrays by the library, against snapshots, in the
        same native call that launches (RIAB_ECHANGED: nothing launched, the general road rebuilds the tables);
        otherwise each population's tables by content through its cached descriptor.  What is left is reserving the
        rows, a few fields of the argument block and the call: ~5 / ~9 us of host time instead of ~17 [MI355X host]."""
        sn = self._snap
        Ns = self.Neurons
        pops = sn["pops"]
        fast = sn["fast"]
        if len(Ns) != len(pops) or _dispatch_depth() > 0:
            return None
        if (_ENV_RAW.get(b"RIAB_NO_NATIVE") == b"1") if (_ENV_RAW is not None and os.name != "nt") else (_L.env("RIAB_NO_NATIVE") == "1"):
            return None
        arr, run = sn["arr"], sn["run"]
        if fast is not None:
            if _agent_fast(self) != fast["ag"]:
                return None
            i = 0
            for N, arrs, getter, sc in fast["pops"]:
                if Ns[i] is not N or getter(N) != sc:
                    return None
                for name, a in arrs:
                    if getattr(N, name) is not a:
                        return None
                i += 1
            Env = self.Environment
            if Env.walls is not fast["walls"] or _env_fast(Env) != fast["env"] or Env._wall_is_hole != fast["holes"]:
                return None
            if sn.get("watching") is not fast:   # (the argument block keeps the list from call to call)
                run.watch, run.n_watch = _L.C.addressof(fast["watch"]), fast["n_watch"]
                sn["watching"] = fast
        else:
            if self.use_imported_trajectory or not self.save_history or self.seed != sn["seed"] or \
                    self.agent_id0 != sn["a0"] or self._time_rate_kernel != sn["timing"] or not self.DIRECT_NATIVE_CALL or \
                    getattr(self, "_timed_population", None) is not sn["timed"]:
                return None
            structs = sn["structs"]
            for i, (N, _n, has_sp) in enumerate(pops):
                if Ns[i] is not N or not N.save_history or bool(N.save_spikes) != has_sp or N._population() is not structs[i]:
                    return None
            if self._motion(sn["dt"], False, 1, _NO_KWARGS) is not sn["m"] or \
                    self.Environment.device_tables(self._device)[0] is not sn["env"]:
                return None
            run.watch, run.n_watch = None, 0
            sn["watching"] = None
        dt = sn["dt"]
        Bp = self._Bp
        traj_c, traj_s = self._hist.reserve_at(n_steps)
        ats = []
        i = 0
        for N, n, has_sp in pops:
            fr_c, fr_s = N._hist_fr.reserve_at(n_steps)
            q = arr[i]
            q.rates_base = fr_c.data_ptr() + fr_s * n * Bp * 4
            if has_sp:
                sp_c, sp_s = N._hist_sp.reserve_at(n_steps)
                q.spikes_base = sp_c.data_ptr() + sp_s * n * Bp
            else:
                sp_c, sp_s = None, 0
            q.capacity_rows = n_steps
            ats.append((fr_c, fr_s, sp_c, sp_s, None))
            i += 1
        run.step0, run.T = self._step_index, n_steps
        run.hist = traj_c.data_ptr() + traj_s * (_L.HIST_ROWS * Bp * 4)
        stream = _L.C.c_void_p(_raw_stream(self._device_index)) if _raw_stream is not None else _L.current_stream()
        hc = self._host_clock   # (bench.py: where a short region's host time goes — None unless asked for)
        if hc is not None:
            t_b = _perf_counter()
        rc = _L.lib.riab_simulate(self._streamer, sn["byref"], stream)
        if hc is not None:
            hc.append((t_b, _perf_counter()))
        if rc:
            self._snap = None
            if rc == _L.EUNSUPPORTED or rc == _L.ECHANGED:   # (nothing was launched: give the rows back, the general road decides)
                self._hist.unreserve(n_steps)
                for (N, _n, has_sp) in pops:
                    N._hist_fr.unreserve(n_steps)
                    if has_sp:
                        N._hist_sp.unreserve(n_steps)
                return None
            self._native_failed(rc, 0, n_steps, dt)
        return self._after_native(n_steps, dt, traj_c, traj_s, sn["neurons"], ats, n_steps, sn)

    def simulate_args(self, n_steps, neurons=None):
        """The arguments of `torch.ops.riab.simulate_` for `n_steps` steps of this agent and `neurons` (default: all of
        its populations) into buffers of their own — for user code that wants the open-loop run as an operator inside
        a traced / compiled function:

            a = Ag.simulate_args(64)
            def run_and_reduce(state, hist, rates):
                torch.ops.riab.simulate_(state, hist, rates, a.spikes, a.ctrl, a.diag, a.streamer, a.run, 0, a.rate_rows,
                                         a.spike_rows)
                return rates[0].mean(dim=(0, 2))
            torch.compile(run_and_reduce, fullgraph=True)(a.state, a.hist, a.rates)

        The block describes ONE run (steps `step_index .. step_index + n_steps` of the agent's Philox streams from the
        state tensor's current contents); the host-side clocks and histories of the Agent object are not advanced by
        calls of the operator.  Keep the returned object alive while the operator may run."""
        import types
        neurons = list(self.Neurons if neurons is None else neurons)
        if self._streamer is None:
            self._make_streamer()
        index, structs = {}, []
        for N in neurons:
            structs.append(N._population(index))
            index[N] = len(index)
        env, walls = self.Environment.device_tables(self._device)
        m = self._motion(self.dt, False, 1, {})
        Bp, T = self._Bp, int(n_steps)
        arr = (_L.RiabPopulation * len(neurons))()
        rates, spikes = [], []
        for i, (N, pop) in enumerate(zip(neurons, structs)):
            _L.C.memmove(_L.C.byref(arr, i * _L.POP_SIZE), _L.C.byref(pop), _L.POP_SIZE)
            fr = torch.empty((T, int(N.n), Bp), dtype=torch.float32, device=self._device)
            rates.append(fr)
            arr[i].rates_base, arr[i].capacity_rows, arr[i].spikes_base = fr.data_ptr(), T, None
            if N.save_spikes:
                sp = torch.empty((T, int(N.n), Bp), dtype=torch.uint8, device=self._device)
                spikes.append(sp)
                arr[i].spikes_base = sp.data_ptr()
        hist = torch.empty((T, _L.HIST_ROWS, Bp), dtype=torch.float32, device=self._device)
        run = _L.RiabSimulate()
        run.env, run.motion = _L.C.pointer(env), _L.C.pointer(m)
        run.state, run.B, run.agent_id0 = self._state.data_ptr(), Bp, int(self.agent_id0)
        run.seed, run.step0, run.T = int(self.rng_seed), int(self._step_index), T
        run.pops, run.n_pops = _L.C.cast(arr, _L.C.POINTER(_L.RiabPopulation)), len(neurons)
        run.hist, run.diag, run.ctrl, run.timed_pop = hist.data_ptr(), self._diag.data_ptr(), self._ctrl.data_ptr(), -1
        return types.SimpleNamespace(state=self._state, hist=hist, rates=rates, spikes=spikes, ctrl=self._ctrl, diag=self._diag,
                                     streamer=self._streamer.value, run=_L.C.addressof(run), hist_row=0,
                                     rate_rows=[0] * len(rates), spike_rows=[0] * len(spikes),
                                     _keep=(run, arr, structs, env, walls, m, neurons))

    def _native_failed(self, rc, t0, tc, dt):
        """A native call failed after earlier pieces (or, RIAB_EPARTIAL, its own trajectory kernel) had been launched:
        the device state has advanced — advance the host mirrors by what ran, then raise."""
        done = t0 + (tc if rc == _L.EPARTIAL else 0)
        for _ in range(done):
            self.prev_t = self.t
            self.t += dt
            if self.save_history:
                self._times.append(self.t)
        self._step_index += done
        self._last_row = None
        _L.check(rc, "riab_simulate")

    def _noise_tensor(self, noise, T):
        """explicit standard normals `(T, 2, B)` / `(T, B, 2)` / tensor -> device float64 `[T, 2, Bp]`"""
        zt = noise if torch.is_tensor(noise) else torch.as_tensor(np.asarray(noise, dtype=np.float64))
        zt = zt.to(self._device, torch.float64)
        if T == 1 and zt.dim() == 2:
            zt = zt.unsqueeze(0)
        if zt.shape[-1] == 2 and zt.shape[-2] != 2:
            zt = zt.transpose(-1, -2)
        if zt.shape[-1] != self._Bp:  # pad agents
            pad = zt[..., :1].expand(*zt.shape[:-1], self._Bp - zt.shape[-1])
            zt = torch.cat((zt, pad), dim=-1)
        z = zt.contiguous()
        assert z.shape == (T, 2, self._Bp), f"noise must be (T,2,B), got {tuple(z.shape)}"
        return z

    def _resample_tensor(self, resample, T):
        """where agents that end a step in a hole / outside a polygonal boundary are put (parity runs: the reference's
        np.random draws); (T, B, 2) / (B, 2) / (2,) -> device float64 `[T, 2, Bp]`"""
        r = np.asarray(resample, dtype=np.float64)
        r = np.broadcast_to(r.reshape((1,) * (3 - r.ndim) + r.shape) if r.ndim < 3 else r, (T, self._B, 2))
        full = np.empty((T, 2, self._Bp))
        full[:, :, :self._B] = np.transpose(r, (0, 2, 1))
        full[:, :, self._B:] = full[:, :, :1]
        return torch.from_numpy(full).to(self._device)

    def last_rate_kernel_ms(self):
        """Duration of the rate stage of the last native simulate() — one kernel for a single store-bound population,
        a sequence of kernels behind progress gates otherwise (`last_rate_stage_form()`) — from the device clock /
        HIP events on the stream it ran on, after a device synchronisation; None when not timed
        (`Agent._time_rate_kernel = True` enables it)."""
        if self._streamer is None:
            return None
        ms = float(_L.lib.riab_streamer_last_rate_ms(self._streamer))
        return ms if ms >= 0 else None

    def pipeline_info(self):
        """How the flag-coupled pipeline of this agent is set up (riab_streamer_info): launches of the last call; the
        trajectory kernel's stream — host microseconds of [a tiny launch on the caller's stream, one on that stream,
        synchronise both] against both launches on the caller's stream, and how many candidate streams were set aside
        before it (streams that share the caller's hardware queue are 30-50 us slower per pair: csrc/riab_simulate.hip
        side_stream_for); what the choice between the populations form and the chunk form compares."""
        if self._streamer is None:
            return None
        f = lambda k: int(_L.lib.riab_streamer_info(self._streamer, k))  # noqa: E731
        pair = f(4)
        return {"launches_last_call": f(3), "strict_last_call": bool(f(8)),
                "second_stream": {"screened": pair >= 0, "pair_us": round(pair / 1e3, 1) if pair >= 0 else None,
                                  "same_stream_pair_us": round(f(6) / 1e3, 1), "candidates_set_aside": f(5),
                                  "streams_held_by_the_pool": f(10)},
                "form_selection": {"trajectory_step_ns": f(0), "lead_store_MBps": f(1), "measured": bool(f(2))}}

    def last_rate_stage_form(self):
        """Which form the rate stage of the last native simulate() took: "one-kernel", "head+pieces" (a long run of one
        population: the row-following kernel for the first rows, ordinary launches behind progress gates after them),
        "populations" (a leading store-bound population like that, then the others over the whole run), "chunks",
        "serial" (forced positions) or None (riab_streamer_last_form)."""
        if self._streamer is None:
            return None
        return {1: "one-kernel", 2: "chunks", 3: "serial", 4: "populations", 5: "head+pieces"}.get(int(_L.lib.riab_streamer_last_form(self._streamer)))

    def __del__(self):
        try:
            if getattr(self, "_streamer", None):
                _L.lib.riab_streamer_destroy(self._streamer)
                self._streamer = None
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass

    @staticmethod
    def _chunk_schedule(n_steps, chunk):
        """Steps per trajectory launch: uniform chunks.  The pipeline is bound by the rate stage (rocprofv3
        trace: its launches run back to back, 17 us apart); what the trajectory stage adds is the fill, the first
        launch with nothing to overlap.  Measured at cfg 2 / K = 1024 and within the run-to-run spread, so not
        adopted: ramping the first launches (chunk/4, chunk/4, chunk/2, chunk/2) +2 %, but the rate kernel loses
        its large-launch efficiency on those chunks; alternating the rate launches over two streams to hide
        the 17 us: 0 %; tapering the tail: 0 %."""
        n_steps, chunk = int(n_steps), max(int(chunk), 1)
        sched = [chunk] * (n_steps // chunk)
        if n_steps % chunk:
            sched.append(n_steps % chunk)
        return sched

    def _make_streams(self):
        """Two HIP streams: trajectory kernel / firing-rate kernels.  (CU-masked streams,
        hipExtStreamCreateWithCUMask, were tried to keep the two kernels off each other's SIMDs:
        the mask is accepted but has no effect on this ROCm 7.2 stack — a 16-CU mask still fills
        at 4.2 TB/s — so plain streams are used.  Giving either stream a higher HIP priority changes
        nothing either: 1.30-1.32 G agent-steps/s in all three arrangements.)"""
        return (torch.cuda.Stream(device=self._device), torch.cuda.Stream(device=self._device))

    def preallocate_history(self, n_steps):
        """Allocate the HBM for `n_steps` more history rows of the agent and of every Neurons
        population now, so that the stepping loop does not allocate (new; optional)."""
        if self.save_history:
            self._hist.preallocate(n_steps)
        for N in self.Neurons:
            if N.save_history:
                N._hist_fr.preallocate(n_steps)
                if N.save_spikes:
                    N._hist_sp.preallocate(n_steps)

    # ---- history --------------------------------------------------------------------------------
    def _sync_plan(self):
        if self._plan is not None:
            self._plan.sync()

    def make_step_plan(self, neurons=None, capacity=1024):
        """A `StepPlan` (plan.py): `plan.step()` == `Agent.update(); N.update() for N in neurons` with the
        per-step host work done in one native call."""
        from .plan import StepPlan
        return StepPlan(self, neurons, capacity)

    def _materialise_history(self):
        self._sync_plan()
        self._settle_plan()
        self._check_pipeline()
        h = self._hist.stack()[:, :, :self._B].cpu().numpy()  # (T, 8, B)
        sq = (lambda a: a[:, 0]) if self._B == 1 else (lambda a: a)
        pair = lambda i: sq(np.stack((h[:, i], h[:, i + 1]), axis=-1))  # noqa: E731
        return {
            "t": np.array(self._times, dtype=float),
            "pos": pair(_L.H_POS_X),
            "distance_travelled": sq(h[:, _L.H_DIST]),
            "vel": pair(_L.H_VEL_X),
            "rot_vel": sq(h[:, _L.H_ROT_VEL]),
            "head_direction": pair(_L.H_HD_X),
        }

    def get_history_slice(self, t_start=None, t_end=None, framerate=None):
        """A python slice over the history between t_start and t_end at `framerate` frames per second
        (reference Agent.py:1068-1091)."""
        t = self.history["t"]
        t_start = t_start or t[0]
        startid = np.nanargmin(np.abs(t - t_start))
        t_end = t_end or t[-1]
        endid = np.nanargmin(np.abs(t - t_end))
        skiprate = 1 if framerate is None else max(1, int((1 / framerate) / self.dt))
        return slice(startid, endid, skiprate)

    def get_history_arrays(self):
        """history as a dict of NumPy arrays (reference Agent.py:1093-1102)."""
        return dict(self.history.items())

    def get_history_tensor(self):
        """Trajectory history on device: float32 `[T, 8, B_padded]` (rows RIAB_H_*).  After a native simulate() this
        accessor (like every host read) first looks at the pipeline's control words — which waits for the run to
        finish — and raises if a wait of the flag-coupled pipeline gave up and left rows unwritten; the tensor that
        simulate() itself returns is handed out while the kernels run and carries no such check."""
        self._sync_plan()
        self._check_pipeline()
        return self._hist.stack()

    def reset_history(self):
        if self._plan is not None:
            self._plan.close()
        self._check_pipeline()   # (rows about to be dropped may still be owed a recovery: settle that first)
        self._hist.reset()
        self._times = []

    def save_to_history(self, **kwargs):
        raise NotImplementedError("history rows are written by the motion kernel; there is no host-side append")

    # ---- imported trajectories (reference Agent.py:543-659, 255-266) -----------------------------
    def import_trajectory(self, times=None, positions=None, dataset=None, interpolate=True):
        """Replace the random-motion model by playback of `positions (N,2)` sampled at `times (N,)`
        (shared by all agents; `(N, n_agents, 2)` gives each agent its own), cubic-spline
        interpolated on the host (`scipy.interpolate.interp1d`, like the reference) and looped.
        `dataset`: path to an .npz with keys "t" and "pos" (the reference's data format).
        Each update()/simulate() step then moves dt along the trajectory on the device
        (`forced_pos` mode of riab_agent_step): measured velocity, head direction, distance and
        history are computed exactly as for simulated motion."""
        from scipy.interpolate import interp1d
        assert self.Environment.boundary_conditions == "solid", "Only solid boundary conditions are supported"
        if self._plan is not None:
            self._plan.close()  # (a recorded plan holds positions of the trajectory it was recorded with)
        self._trajectory_id = getattr(self, "_trajectory_id", 0) + 1
        if dataset is not None:
            data = np.load(dataset if str(dataset).endswith(".npz") else str(dataset) + ".npz")
            times, positions = data["t"], data["pos"]
        assert times is not None and positions is not None, "provide 'times' and 'positions' (or 'dataset')"
        times, positions = np.array(times, dtype=float), np.array(positions, dtype=float)
        assert len(positions) == len(times), "time and position arrays must have same length"
        times = times - min(times)
        positions = positions.reshape(len(times), -1, 2)
        assert positions.shape[1] in (1, self._B), "positions must be (N,2) or (N,n_agents,2)"
        self.interpolate = interpolate
        self.use_imported_trajectory = True
        self.t_interp = times
        if interpolate:
            self.pos_interp = interp1d(times, positions, axis=0, kind="cubic", fill_value="extrapolate")
            p0 = self.pos_interp(0)
        else:
            self.positions, self.times = positions, times
            self.t = -self.dt
            self.prev_t = -(times[1] - times[0])
            self.imported_trajectory_id = 0
            p0 = positions[0]
        self.pos = np.broadcast_to(p0, (self._B, 2))

    def _advance_imported(self, T, dt, kwargs, hist_view=None, stream=None):
        """T steps along the imported trajectory (Agent._update_position_along_imported_trajectory)."""
        dt = dt or self.dt
        if self.interpolate:
            ts = np.cumsum(np.concatenate(([self.t + dt], np.full(T - 1, dt))))  # the loop's clock: `t += dt` repeated
            pos = self.pos_interp(ts % max(self.t_interp))            # (T, 1|B, 2)
            pos = np.broadcast_to(pos, (T, self._B, 2))
            full = np.empty((T, 2, self._Bp))
            full[:, :, :self._B] = np.transpose(pos, (0, 2, 1))
            full[:, :, self._B:] = full[:, :, :1]
            self._advance(T, dt, None, 1, kwargs, hist_view=hist_view, stream=stream,
                          forced=torch.from_numpy(full).to(self._device))
        else:
            assert T == 1 and stream is None, "interpolate=False trajectories advance one sample per update()"
            i = self.imported_trajectory_id
            t_new = float(self.times[i])
            step_dt = t_new - self.t  # the reference resets dt to the sample spacing (Agent.py:262-263)
            pos = np.broadcast_to(self.positions[i], (self._B, 2))
            self.imported_trajectory_id = (i + 1) % len(self.times)
            t_before = self.t
            self._advance(1, step_dt, None, 1, kwargs, forced=self._as_device_f64(pos, 2).unsqueeze(0))
            self.prev_t, self.t = t_before, t_new
            if self.save_history:
                self._times[-1] = self.t

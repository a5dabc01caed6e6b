"""`StepPlan` — the closed-loop per-step path as one native call (include/riab_hip.h: riab_plan_*).

    plan = Ag.make_step_plan()          # records Ag and Ag.Neurons once
    for i in range(T):
        action = policy(Ag.state_tensor)            # device tensor (B,2) or None
        plan.step(drift_velocity=action)            # == Ag.update(drift_velocity=action); N.update() ...

`plan.step()` costs one ctypes call; row cursors, RNG counters and kernel arguments advance in C++
(csrc/riab_plan.hip).  Histories are written into rows opened in the same device chunks the eager
path uses; the Python mirrors (`history`, `firingrate`, `_times`) are brought up to date lazily by
`sync()` (called automatically by the accessors).  Results are identical to the eager loop: same
kernels, same arguments, same RNG counters.  Populations with `noise_std > 0` are covered (their OU
parameters are fixed at the plan's dt), FeedForwardLayers too as long as their input layers are
recorded before them (weights are read from the device copies made when the plan is built:
rebuild the plan after editing `inputs[...]["w"]`)."""
import numpy as np
import torch

from . import _lib

_L = _lib


def _attach_fused(plan, agent):
    """Arrival words for the one-launch step (riab_plan_set_fused, csrc/riab_step1.hip): whole 256-agent segments
    only; the library decides per step whether the step qualifies.  The plan is told how many compute units this
    process's workgroups really land on (`_lib.compute_units`: measured once per device — a CU mask or a partition is
    invisible to hipGetDeviceProperties): the one-launch step cuts its grid to that, and a task plan whose grid would
    not be resident at once keeps its two launches."""
    plan._sync_words = None
    plan._fused_seen = 0      # one-launch steps the plan had taken when its give-up counter was last looked at
    if agent._Bp % 256 == 0 and not _L.env("RIAB_NO_FUSED_STEP"):
        n = _L.step1_sync_words(agent._Bp)
        plan._sync_words = torch.zeros(n + 256, dtype=torch.int32, device=agent._device)   # (+ slack: tools/step1_profile.py)
        _L.check(_L.lib.riab_plan_set_fused(plan._h, _L.ptr(plan._sync_words), n), "riab_plan_set_fused")
        _L.check(_L.lib.riab_plan_set_compute_units(plan._h, _L.compute_units(agent._device_index)),
                 "riab_plan_set_compute_units")


_FUSABLE = ("PlaceCells", "GridCells", "HeadDirectionCells")   # (what csrc/riab_step1.hip step1_supported admits)


def _settle_fused(plan):
    """A workgroup of the one-launch step that gave up waiting (bounded spins, ~1 s: a writer for its segment's arrival
    words, or — task plans — another workgroup for the writer's verdict on this step's resets; a device shared with
    something that kept part of the grid off it for that long).  Looked at wherever the host reads results anyway (the
    first read after the steps synchronises) and when the plan closes.

    What such a step leaves wrong is RATES only: a writer works the new state out from its own loads and waits for the
    others only before it STORES; nobody but a writer stores state, history or task rows; a workgroup that comes late
    may read a state, an action or a position that is already the next step's, and writes its tile of the populations'
    rows from that.  The kernel remembers the first and the last step a wait gave up in: the fused populations' rows of
    those steps are recomputed here from the agent's history rows with the populations' own kernels — what the
    two-launch step runs, hence the bits it would have written — counted (`Agent.diagnostics["step1_recovered_steps"]`),
    one warning.  What cannot be redone raises as before: rows whose agent row is gone (no agent history kept)."""
    w = getattr(plan, "_sync_words", None)
    if w is None or not plan._h:
        return
    fused = int(_L.lib.riab_plan_info(plan._h, 0))
    if fused == plan._fused_seen:
        return
    plan._fused_seen = fused
    ag = plan.agent
    tail = _L.step1_sync_tail(ag._Bp)
    t = w[tail:tail + 4].cpu().numpy().astype(np.int64) & 0xFFFFFFFF     # (waits for the steps)
    n = int(t[_L.STEP1_SYNC_TIMEOUTS])
    if not n:
        return
    first, last = int(t[_L.STEP1_SYNC_FIRST_BAD]), int(t[_L.STEP1_SYNC_LAST_BAD])
    torch.cuda.synchronize(ag._device)
    w[tail:tail + 4] = 0
    if int(t[_L.STEP1_SYNC_FATAL]):   # (one-world task plans: a writer never learnt whether the world was reset)
        raise _L.RiabError(f"one-launch step of a one-world task: {int(t[_L.STEP1_SYNC_FATAL])} writer workgroups gave up waiting "
                           f"for the world's verdict in steps {first}..{last}; the agent state of those steps is not trustworthy")
    plan.sync()
    now = int(ag._step_index)
    if not (0 < first <= last <= now) or last - first >= 1 << 16:
        raise _L.RiabError(f"one-launch step: {n} waits gave up in steps {first}..{last} (now {now}): not a range this plan took")
    stream = _L.current_stream()
    redone = 0
    for s in range(first, last + 1):
        back = now - s                       # rows from the newest one
        if ag.save_history:
            k = len(ag._times) - 1 - back
            row = ag._hist.row(k) if k >= 0 else None
            t_s = ag._times[k] if k >= 0 else None
        else:
            row, t_s = (ag._last_row if back == 0 else None), None
        for N in plan.neurons:
            if N.__class__.__name__ not in _FUSABLE or N.noise_std != 0:
                continue
            if N.save_history:
                if row is None:
                    raise _L.RiabError(f"one-launch step: a wait gave up in step {s} and the agent's row of that step is "
                                       "not kept (save_history=False): the population's row cannot be recomputed")
                kp = None if t_s is None else next((j for j in range(len(N._times) - 1, max(-1, len(N._times) - 3 - back), -1)
                                                    if N._times[j] == t_s), None)
                if kp is None:
                    if back == 0 and hasattr(plan, "discard_ahead"):
                        plan.discard_ahead()   # (written ahead, its update() not called yet: that call recomputes it)
                    continue
                fr = N._hist_fr.row(kp)
                sp = N._hist_sp.row(kp) if N.save_spikes else None
            elif back == 0 and row is not None:
                fr, sp = N._rates, None
            else:
                continue                     # (a population without history: only its newest row exists)
            N._launch(row[_L.H_POS_X], row[_L.H_POS_Y], row[N._H_DIR[0]], row[N._H_DIR[1]], pos_ld=ag._Bp, T=1, B=ag._Bp,
                      rates=fr.unsqueeze(0), spikes=None if sp is None else sp.unsqueeze(0), u_in=None, dt=float(ag.dt),
                      step0=s, stream=stream)
            redone += 1
    torch.cuda.synchronize(ag._device)
    ag._step1_timeouts += n
    ag._step1_recovered_steps += last - first + 1
    if not getattr(ag, "_step1_warned", False):
        ag._step1_warned = True
        import warnings
        warnings.warn(f"one-launch step: {n} waits of its workgroups gave up in steps {first}..{last} (something kept part "
                      f"of the grid off the device for about a second); {redone} rows of the fused populations were "
                      "recomputed from the agent's history rows with the populations' own kernels "
                      "(diagnostics['step1_recovered_steps'])", RuntimeWarning)


def _plan_info(h):
    f = _L.lib.riab_plan_info
    pops = [int(f(h, 8 + k)) for k in range(_L.STEP1_MAX_POPS)]
    return {"fused_steps": int(f(h, 0)), "fused_population": int(f(h, 1)), "launches": int(f(h, 2)),
            "fused_enabled": bool(f(h, 3)), "fused_populations": [i for i in pops if i >= 0], "compute_units": int(f(h, 5))}


class _ForcedRows:
    """Positions of the coming steps of an agent that follows an imported trajectory (Agent.import_trajectory with
    interpolate=True; reference Agent.py:255-266), handed to a native plan in blocks (riab_plan_set_forced): the
    interpolation runs on the host once per block instead of once per update()."""

    def __init__(self, agent, handle, block=None):
        if not agent.interpolate:
            raise NotImplementedError("interpolate=False trajectories reset dt to the sample spacing every step: "
                                      "they advance through update()")
        self.agent, self._h = agent, handle
        self.block = int(block or max(64, min(1024, (1 << 22) // max(agent._Bp, 1))))
        self.left = 0
        self._buf = None

    def ensure(self, n, dt):
        """Rows for at least the next n steps (n <= block) are with the plan."""
        if self.left >= n:
            return
        ag = self.agent
        rows = max(self.block, n)
        t, ts = ag.t, np.empty(rows)
        for k in range(rows):   # the clock the eager path would have had at each of these steps: repeated `t += dt`
            t += dt
            ts[k] = t
        pos = np.broadcast_to(ag.pos_interp(ts % max(ag.t_interp)), (rows, ag._B, 2))
        full = np.empty((rows, 2, ag._Bp))
        full[:, :, :ag._B] = np.transpose(pos, (0, 2, 1))
        full[:, :, ag._B:] = full[:, :, :1]
        self._buf = torch.from_numpy(full).to(ag._device)   # (the previous block may still be read by queued kernels:
        self._keep_prev = getattr(self, "_buf_prev", None)  #  keep it alive for one more block)
        self._buf_prev = self._buf
        _L.check(_L.lib.riab_plan_set_forced(self._h, _L.ptr(self._buf), rows), "riab_plan_set_forced")
        self.left = rows

    def used(self, n):
        self.left -= n


class StepPlan:
    def __init__(self, agent, neurons=None, capacity=1024):
        self.agent = agent
        self.neurons = list(agent.Neurons if neurons is None else neurons)
        self.capacity = int(capacity)
        agent._sync_plan()
        if agent._plan is not None:
            agent._plan.close()
        Bp = agent._Bp
        self._row_scratch = torch.empty((_L.HIST_ROWS, Bp), dtype=torch.float32, device=agent._device)
        self._dt = agent.dt
        self._drift_key = None
        self._drift_t = None
        env, self._walls = agent.Environment.device_tables(agent._device)
        motion = agent._motion(agent.dt, False, 1, {})
        self._h = _L.lib.riab_plan_create(env, motion, _L.ptr(agent._state), Bp, int(agent.agent_id0), int(agent.rng_seed),
                                          int(agent._step_index), _L.ptr(self._row_scratch),
                                          _L.ptr(agent._diag))
        if not self._h:
            raise _L.RiabError("riab_plan_create failed")
        self._forced = _ForcedRows(agent, self._h, block=self.capacity) if agent.use_imported_trajectory else None
        self._step_fn = _L.lib.riab_plan_step
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._raw_stream = raw if raw is not None else (lambda _i: torch.cuda.current_stream().cuda_stream)
        self._dev_index = agent._device_index
        self._pops = []
        index = {}
        for N in self.neurons:
            pop = N._population(index)
            N._plan_scratch(pop)
            index[N] = len(index)
            idx = _L.lib.riab_plan_add(self._h, pop)
            if idx < 0:
                raise _L.RiabError(f"riab_plan_add failed: {_L.strerror(idx)}")
            self._pops.append(pop)
        _attach_fused(self, agent)
        self._pending = 0        # steps taken since the last sync()
        self._times_pending = []
        self._rows_open = 0      # rows still free in the attached chunks
        self._chunk_rows_done = 0
        self._agent_rows = None
        self._pop_rows = [None] * len(self.neurons)
        self._scratch_rates = [None] * len(self.neurons)
        self._task_env = None    # TaskEnvironment attached by attach_task()
        self._actions = None     # persistent drift buffer [2, Bp] the plan reads when a task is attached
        self._auto_reset = False
        self._scripted = False
        agent._plan = self
        self._attach()

    # ---- history chunks --------------------------------------------------------------------------
    def _attach(self, need=1):
        """Open up to `capacity` free rows in every history — the same number in each: what the shortest non-empty
        free tail holds, at least `need` (the rows of one step() call are contiguous: a tail shorter than that is left
        behind) — and hand their base pointers to the plan."""
        self.sync()
        ag = self.agent
        hists = ([ag._hist] if ag.save_history else []) + [h for N in self.neurons if N.save_history
                                                            for h in ((N._hist_fr, N._hist_sp) if N.save_spikes else (N._hist_fr,))]
        cap = self.capacity
        for h in hists:
            if 0 < h.free_rows() < need:
                h.preallocate(cap)
            if h.free_rows() > 0:
                cap = min(cap, h.free_rows())
        if ag.save_history:
            self._agent_rows = ag._hist.open_rows(cap, self.capacity)
            _L.check(_L.lib.riab_plan_set_agent_history(self._h, _L.ptr(self._agent_rows), cap), "riab_plan_set_agent_history")
        else:
            self._agent_rows = None
            _L.check(_L.lib.riab_plan_set_agent_history(self._h, None, 0), "riab_plan_set_agent_history")
        for i, N in enumerate(self.neurons):
            if N.save_history:
                fr = N._hist_fr.open_rows(cap, self.capacity)
                sp = N._hist_sp.open_rows(cap, self.capacity) if N.save_spikes else None
                self._pop_rows[i] = (fr, sp)
                rc = _L.lib.riab_plan_set_population_history(self._h, i, _L.ptr(fr), _L.ptr(sp), cap)
            else:
                if self._scratch_rates[i] is None:
                    self._scratch_rates[i] = torch.empty((1, int(N.n), ag._Bp), dtype=torch.float32, device=ag._device)
                self._pop_rows[i] = (self._scratch_rates[i], None)
                rc = _L.lib.riab_plan_set_population_history(self._h, i, _L.ptr(self._scratch_rates[i]), None, 0)
            _L.check(rc, "riab_plan_set_population_history")
        self._rows_open = cap
        self._chunk_rows_done = 0

    # ---- task ------------------------------------------------------------------------------------
    def attach_task(self, env, auto_reset=True, scripted_speed=None):
        """Make every plan step one `TaskEnvironment.step` (contribs/TaskEnvironment.py): after the motion
        kernel the task kernel updates rewards / goals of all lanes; with `auto_reset` the lanes that
        became terminal are reset right away (the caller's `if terminal: env.reset()`), before the
        populations are evaluated.  `scripted_speed`: the action of every step is
        `scripted_speed * unit goal vector`, computed on the device (no policy on the host)."""
        ag = self.agent
        assert env._agent is ag, "the plan's agent is not the one added to this TaskEnvironment"
        task = env._task_struct()
        n_sel = min(env.goal_cache.reset_n_goals, task.n_pool)
        self._task_env, self._task_struct_keep = env, task
        self._auto_reset, self._scripted = bool(auto_reset), bool(scripted_speed)
        self._actions = torch.zeros((2, ag._Bp), dtype=torch.float64, device=ag._device)
        rc = _L.lib.riab_plan_set_task(self._h, task, _L.ptr(env.task_state), env._B, float(env.t), float(env.dt),
                                       _L.ptr(env._reward), _L.ptr(env._terminal), _L.ptr(env._diag), int(self._auto_reset),
                                       int(n_sel), int(bool(env.goal_cache.reset_orders_goal)), env._task_seed,
                                       env._reset_counter, int(bool(env.teleport_on_reset)), _L.ptr(env._ep_log), env._ep_cap,
                                       _L.ptr(env._ep_count), float(scripted_speed or 0.0))
        _L.check(rc, "riab_plan_set_task")
        if env._shared:   # the lanes are the agents of one world: the shared state and the step kernel's scratch
            _L.check(_L.lib.riab_plan_set_task_world(self._h, _L.ptr(env._world), _L.ptr(env._met), _L.ptr(env._cand),
                                                     _L.ptr(env._ticket)), "riab_plan_set_task_world")
        self._drift_key = None
        return self

    # ---- stepping --------------------------------------------------------------------------------
    def step(self, n_steps=1, drift_velocity=None, drift_to_random_strength_ratio=1, dt=None):
        ag = self.agent
        if ag._plan is not self:
            raise RuntimeError("this plan was closed (the agent was stepped eagerly or its history reset)")
        dt = dt or ag.dt
        if self._task_env is not None:
            # actions land in the plan's persistent drift buffer: no per-step parameter resolution
            has_drift = drift_velocity is not None or self._scripted
            if drift_velocity is not None:
                a = drift_velocity if torch.is_tensor(drift_velocity) else torch.as_tensor(np.asarray(drift_velocity, dtype=np.float64))
                a = torch.nan_to_num(a.to(ag._device, torch.float64), nan=0.0)   # TaskEnvironment.py:403-404
                self._actions[:, :ag._B].copy_(a.reshape(-1, 2).t() if a.numel() > 2 else a.reshape(2, 1))
            key = (dt, has_drift, drift_to_random_strength_ratio, "task")
            if key != self._drift_key:
                motion = ag._motion(dt, has_drift, drift_to_random_strength_ratio, {})
                _L.check(_L.lib.riab_plan_set_motion(self._h, motion, _L.ptr(self._actions) if has_drift else None),
                         "riab_plan_set_motion")
                self._drift_key = key
                ag.dt = dt
        else:
            key = (dt, drift_velocity is not None, drift_to_random_strength_ratio)
            if key != self._drift_key or drift_velocity is not None:
                drift = ag._as_device_f64(drift_velocity, 2) if drift_velocity is not None else None
                motion = ag._motion(dt, drift is not None, drift_to_random_strength_ratio, {})
                _L.check(_L.lib.riab_plan_set_motion(self._h, motion, _L.ptr(drift)), "riab_plan_set_motion")
                self._drift_t, self._drift_key = drift, key
                ag.dt = dt
        if n_steps > self._rows_open:
            if n_steps > self.capacity:
                raise ValueError(f"n_steps {n_steps} exceeds the plan's chunk capacity {self.capacity}")
            self._attach(need=n_steps)
        if self._forced is not None:   # the agent follows its imported trajectory: positions of the coming steps
            if drift_velocity is not None or self._task_env is not None:
                raise NotImplementedError("an agent on an imported trajectory takes neither a drift velocity nor a task")
            self._forced.ensure(int(n_steps), dt)
        # (a step is ~9 us of GPU time: the stream handle comes straight from the C API by the agent's device index, the
        # return code is looked at here)
        rc = self._step_fn(self._h, n_steps, self._raw_stream(self._dev_index))
        if rc:
            _L.check(rc, "riab_plan_step")
        if self._forced is not None:
            self._forced.used(int(n_steps))
        self._rows_open -= n_steps
        self._pending += n_steps
        if n_steps == 1:
            ag.prev_t = ag.t
            ag.t += dt
            self._times_pending.append(ag.t)
        else:
            for _ in range(n_steps):
                ag.prev_t = ag.t
                ag.t += dt
                self._times_pending.append(ag.t)
        ag._step_index += n_steps
        env = self._task_env
        if env is not None:
            for _ in range(n_steps):
                env.update()               # the clock (the native plan advances its copy the same way)
            if self._auto_reset:
                env._reset_counter += n_steps

    def sync(self):
        """Publish the rows written since the last sync to the Python-side mirrors."""
        n = self._pending
        if not n:
            return
        self._pending = 0
        ag = self.agent
        times, self._times_pending = self._times_pending, []
        done = self._chunk_rows_done
        if ag.save_history:
            ag._hist.commit(n)
            ag._times.extend(times)
            ag._last_row = self._agent_rows[done + n - 1]
        else:
            ag._last_row = self._row_scratch
        for N, (fr, sp) in zip(self.neurons, self._pop_rows):
            if N.save_history:
                N._hist_fr.commit(n)
                if sp is not None:
                    N._hist_sp.commit(n)
                N._times.extend(times)
                N._rates = fr[done + n - 1]
                N._spikes_last = None if sp is None else sp[done + n - 1]
            else:
                N._rates = fr[0]
        self._chunk_rows_done = done + n

    def close(self):
        """Detach from the agent (called automatically when the agent is stepped eagerly)."""
        self.sync()
        if self.agent._plan is self:
            self.agent._plan = None
        if self._h:
            try:
                _settle_fused(self)
            finally:
                _L.lib.riab_plan_destroy(self._h)
                self._h = None

    def settle(self):
        """Look at the one-launch step's give-up counter; recover (see _settle_fused)."""
        _settle_fused(self)

    def info(self):
        """Launch accounting of the native plan (riab_plan_info)."""
        return _plan_info(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _L.lib.riab_plan_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass


class AutoStepper:
    """The reference's per-object loop served natively WITHOUT the caller changing a line:

        for i in range(T):
            Ag.update()          # -> one riab_plan_step_agent call
            PCs.update()         # -> one riab_plan_step_population call

    After `Agent.AUTO_AFTER` consecutive plain `Agent.update()` calls (no arguments) the Agent records itself and
    its populations in a native plan and serves the following plain `update()` calls — its own and its
    populations' — from it: the per-call host work drops from ~15 us (parameter resolution, struct filling, history
    bookkeeping) to one ctypes transition and a staleness check.  The calls keep their meaning: the same rows, arguments
    and RNG counters as the eager path, so results are bit-identical.  With the one-launch step (csrc/riab_step1.hip)
    `Agent.update()` launches ONE kernel that also writes this step's rows of the store-bound populations that keep a
    history (PlaceCells / GridCells / HeadDirectionCells without noise) — rows that are not yet part of the history — and
    each such population's `update()` only publishes its row; a population that keeps no history (its single row IS what
    `firingrate` shows until its own update()) always launches its own kernel at the time of its call, as do the
    populations the loop does not update after every agent step.  Any call that is not plain (kwargs, drift, another dt), any host edit of the
    agent state, `simulate()`, `reset_history()` or an explicit step plan closes the stepper and the eager path
    takes over (and re-engages later).  Attribute edits are caught by value: the motion parameters, the wall table
    and each population's tables are compared with what the plan was built from on every call.

    Registered in `agent._plan`, so the history accessors publish its pending rows like a StepPlan's."""

    CAPACITY = 4096          # rows per attached history chunk, at most
    CHUNK_BYTES = 4 << 30    # ... and at most this much HBM per population and chunk (cfg 2: 16.8 MB per row -> 255 rows)

    def __init__(self, agent):
        self.agent = agent
        self.neurons = list(agent.Neurons)
        agent._sync_plan()
        if agent._plan is not None:
            agent._plan.close()
        Bp = agent._Bp
        self._row_scratch = torch.empty((_L.HIST_ROWS, Bp), dtype=torch.float32, device=agent._device)
        self._dt = agent.dt
        self._env_struct, self._walls = agent.Environment.device_tables(agent._device)
        self._motion = agent._motion(agent.dt, False, 1, {})
        self._motion_key = agent._motion_cache[0]
        self._drift_buf = None
        self._h = _L.lib.riab_plan_create(self._env_struct, self._motion, _L.ptr(agent._state), Bp, int(agent.agent_id0),
                                          int(agent.rng_seed), int(agent._step_index),
                                          _L.ptr(self._row_scratch), _L.ptr(agent._diag))
        if not self._h:
            raise _L.RiabError("riab_plan_create failed")
        self._h = _L.C.c_void_p(self._h)
        self._forced = _ForcedRows(agent, self._h) if agent.use_imported_trajectory else None
        self._traj_id = getattr(agent, "_trajectory_id", 0)
        self._index, self._pops, self._keys = {}, [], []
        for N in self.neurons:
            pop = N._population(self._index)          # (raises NotImplementedError for populations a plan cannot hold)
            N._plan_scratch(pop)
            idx = _L.lib.riab_plan_add(self._h, pop)
            if idx < 0:
                raise _L.RiabError(f"riab_plan_add failed: {_L.strerror(idx)}")
            self._index[N] = idx
            self._pops.append(pop)
            self._keys.append(N._auto_key())
        _attach_fused(self, agent)
        # (two native calls per step of the user's loop: the stream handle straight from the C API by device index, the
        # entry points bound once)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._raw_stream = raw if raw is not None else (lambda _i: torch.cuda.current_stream().cuda_stream)
        self._dev_index = agent._device_index
        self._step_agent_fn, self._step_pop_fn = _L.lib.riab_plan_step_agent, _L.lib.riab_plan_step_population
        self._a_pending, self._a_times = 0, []
        self._p_pending = [0] * len(self.neurons)
        self._p_times = [[] for _ in self.neurons]
        self._agent_rows, self._a_done = None, 0
        self._pop_rows = [None] * len(self.neurons)
        self._p_done = [0] * len(self.neurons)
        self._scratch_rates = [None] * len(self.neurons)
        agent._plan = self
        self._attach()

    # ---- history chunks -------------------------------------------------------------------------
    def _attach(self):
        self._attach_agent()
        for i in range(len(self.neurons)):
            self._attach_pop(i)

    def _attach_agent(self):
        """Open history rows for the agent (what is left of the current chunk, else a new chunk) — the agent's only:
        a full agent chunk does not touch the populations' part-filled ones."""
        self.sync()
        ag = self.agent
        if ag.save_history:
            self._agent_rows = ag._hist.open_rows(self.CAPACITY)
            rc = _L.lib.riab_plan_set_agent_history(self._h, _L.ptr(self._agent_rows), int(self._agent_rows.shape[0]))
        else:
            self._agent_rows = None
            rc = _L.lib.riab_plan_set_agent_history(self._h, None, 0)
        _L.check(rc, "riab_plan_set_agent_history")
        self._a_done = 0

    def _attach_pop(self, i):
        N, ag = self.neurons[i], self.agent
        self._sync_pop(i)
        if N.save_history:
            full = cap = int(max(64, min(self.CAPACITY, self.CHUNK_BYTES // max(1, int(N.n) * ag._Bp * 4))))
            if N.save_spikes:  # (rates and spikes advance together; should their free tails differ, the shorter one counts)
                tails = [h.free_rows() for h in (N._hist_fr, N._hist_sp) if h.free_rows() > 0]
                cap = min([cap] + tails)
            fr = N._hist_fr.open_rows(cap, full)   # (a history without free rows opens a full-sized chunk, not a `cap`-row one)
            cap = int(fr.shape[0])
            sp = N._hist_sp.open_rows(cap, full) if N.save_spikes else None
            self._pop_rows[i] = (fr, sp)
            rc = _L.lib.riab_plan_set_population_history(self._h, i, _L.ptr(fr), _L.ptr(sp), cap)
        else:
            if self._scratch_rates[i] is None:
                self._scratch_rates[i] = torch.empty((1, int(N.n), ag._Bp), dtype=torch.float32, device=ag._device)
            self._pop_rows[i] = (self._scratch_rates[i], None)
            rc = _L.lib.riab_plan_set_population_history(self._h, i, _L.ptr(self._scratch_rates[i]), None, 0)
        _L.check(rc, "riab_plan_set_population_history")
        self._p_done[i] = 0

    # ---- the two fast paths ---------------------------------------------------------------------
    def step_agent(self, drift_velocity=None, ratio=1):
        """Agent.update() with no arguments, or with a drift velocity (the closed loop: `update(drift_velocity=
        policy(obs))`, TaskEnvironment.step).  False: something changed — the caller closes the stepper and goes eager."""
        ag = self.agent
        if ag.dt != self._dt:
            return False
        has = drift_velocity is not None
        key = ag._motion_key_now(ag.dt, has, ratio if has else 1)
        if key != self._motion_key:
            if key[3:] != self._motion_key[3:] or self._forced is not None:
                return False              # a motion parameter changed (or a drift on a replayed trajectory): eager
            # only (drift given?, its strength ratio) differ: the plan takes the other struct, same kernels
            self._motion = ag._motion(ag.dt, has, ratio, {})
            if has and self._drift_buf is None:
                self._drift_buf = torch.zeros((2, ag._Bp), dtype=torch.float64, device=ag._device)
            _L.check(_L.lib.riab_plan_set_motion(self._h, self._motion, _L.ptr(self._drift_buf) if has else None),
                     "riab_plan_set_motion")
            self._motion_key = key
        if has:
            self._write_drift(drift_velocity)
        env, _w = ag.Environment.device_tables(ag._device)
        if env is not self._env_struct:   # (device_tables returns the cached struct while the geometry is unchanged)
            return False
        if (self._forced is not None) != bool(ag.use_imported_trajectory) or self._traj_id != getattr(ag, "_trajectory_id", 0):
            return False                  # (a trajectory was imported, or another one, since the plan was recorded)
        if self._forced is not None:
            self._forced.ensure(1, self._dt)
        rc = self._step_agent_fn(self._h, self._raw_stream(self._dev_index))
        if rc:
            if rc == _L.EFULL:
                self._attach_agent()
                rc = self._step_agent_fn(self._h, self._raw_stream(self._dev_index))
            _L.check(rc, "riab_plan_step_agent")
        if self._forced is not None:
            self._forced.used(1)
        ag.prev_t = ag.t
        ag.t += self._dt
        ag._step_index += 1
        self._a_pending += 1
        self._a_times.append(ag.t)
        self._served = getattr(self, "_served", 0) + 1
        return True

    def _write_drift(self, x):
        """The step's drift velocity into the plan's persistent device buffer [2, Bp] (what Agent._as_device_f64 would
        have allocated: padded lanes carry agent 0's value)."""
        buf, ag = self._drift_buf, self.agent
        B, Bp = ag._B, ag._Bp
        if torch.is_tensor(x) and x.device == buf.device and x.dim() == 2:
            if x.shape == (B, 2):
                buf[:, :B].copy_(x.t())
            elif x.shape == (2, B):
                buf[:, :B].copy_(x)
            else:
                buf.copy_(ag._as_device_f64(x, 2))
                return
            if Bp != B:
                buf[:, B:] = buf[:, :1]
        else:
            buf.copy_(ag._as_device_f64(x, 2))

    def discard_ahead(self):
        """The positions were edited on the device after this step's Agent.update() (TaskEnvironment.reset teleported agents):
        the row the one-launch step wrote ahead for its population is stale — its update() recomputes it."""
        _L.check(_L.lib.riab_plan_discard_ahead(self._h), "riab_plan_discard_ahead")

    def step_population(self, N):
        """N.update() with no arguments, N one of the recorded populations with unchanged tables."""
        i = self._index.get(N)
        if i is None or N._auto_key() != self._keys[i]:
            return False
        rc = self._step_pop_fn(self._h, i, self._raw_stream(self._dev_index))
        if rc:
            if rc == _L.EFULL:
                self._attach_pop(i)
                rc = self._step_pop_fn(self._h, i, self._raw_stream(self._dev_index))
            _L.check(rc, "riab_plan_step_population")
        self._p_pending[i] += 1
        self._p_times[i].append(self.agent.t)
        return True

    # ---- publishing -----------------------------------------------------------------------------
    def _sync_pop(self, i):
        n = self._p_pending[i]
        if not n or self._pop_rows[i] is None:
            return
        self._p_pending[i] = 0
        N = self.neurons[i]
        fr, sp = self._pop_rows[i]
        times, self._p_times[i] = self._p_times[i], []
        if N.save_history:
            done = self._p_done[i]
            N._hist_fr.commit(n)
            if sp is not None:
                N._hist_sp.commit(n)
            N._times.extend(times)
            N._rates = fr[done + n - 1]
            N._spikes_last = None if sp is None else sp[done + n - 1]
            self._p_done[i] = done + n
        else:
            N._rates = fr[0]

    def sync(self):
        """Publish the rows written since the last sync to the Python-side mirrors."""
        ag = self.agent
        n = self._a_pending
        if n:
            self._a_pending = 0
            times, self._a_times = self._a_times, []
            if ag.save_history:
                ag._hist.commit(n)
                ag._times.extend(times)
                ag._last_row = self._agent_rows[self._a_done + n - 1]
                self._a_done += n
            else:
                ag._last_row = self._row_scratch
        for i in range(len(self.neurons)):
            self._sync_pop(i)

    def close(self):
        self.sync()
        ag = self.agent
        if ag._plan is self:
            ag._plan = None
        ag._auto_streak = 0
        # back-off: a stepper that was closed after a handful of steps was not worth recording
        served = getattr(self, "_served", 0)
        if served < ag.AUTO_KEEP:
            ag._auto_after = min(2 * max(ag._auto_after, 1), ag.AUTO_AFTER_MAX)
        else:
            ag._auto_after = ag.AUTO_AFTER
        if self._h:
            try:
                _settle_fused(self)
            finally:
                _L.lib.riab_plan_destroy(self._h)
                self._h = None

    def settle(self):
        _settle_fused(self)

    def info(self):
        return _plan_info(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _L.lib.riab_plan_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass

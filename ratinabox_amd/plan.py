"""`StepPlan` — the closed-loop per-step path as one native call (include/riab_hip.h: riab_plan_*).

    plan = Ag.make_step_plan()          # records Ag and Ag.Neurons once
    for i in range(T):
        action = policy(Ag.state_tensor)            # device tensor (B,2) or None
        plan.step(drift_velocity=action)            # == Ag.update(drift_velocity=action); N.update() ...

`plan.step()` costs one ctypes call; row cursors, RNG counters and kernel arguments advance in C++
(csrc/riab_plan.hip).  Histories are written into rows opened in the same device chunks the eager
path uses; the Python mirrors (`history`, `firingrate`, `_times`) are brought up to date lazily by
`sync()` (called automatically by the accessors).  Results are identical to the eager loop: same
kernels, same arguments, same RNG counters.  Populations with `noise_std > 0` and
FeedForwardLayers are not covered (use `update()`)."""
import numpy as np
import torch

from . import _lib

_L = _lib


class StepPlan:
    def __init__(self, agent, neurons=None, capacity=1024):
        self.agent = agent
        self.neurons = list(agent.Neurons if neurons is None else neurons)
        self.capacity = int(capacity)
        agent._sync_plan()
        if agent._plan is not None:
            agent._plan.close()
        if agent.use_imported_trajectory:
            raise NotImplementedError("imported trajectories advance through update()/simulate()")
        Bp = agent._Bp
        self._row_scratch = torch.empty((_L.HIST_ROWS, Bp), dtype=torch.float32, device=agent._device)
        self._dt = agent.dt
        self._drift_key = None
        self._drift_t = None
        env, self._walls = agent.Environment.device_tables(agent._device)
        motion = agent._motion(agent.dt, False, 1, {})
        self._h = _L.lib.riab_plan_create(env, motion, _L.ptr(agent._state), Bp, int(agent.agent_id0), int(agent.seed),
                                          int(agent._step_index), int(agent.precision), _L.ptr(self._row_scratch),
                                          _L.ptr(agent._diag))
        if not self._h:
            raise _L.RiabError("riab_plan_create failed")
        self._pops = []
        for N in self.neurons:
            pop = N._population()
            idx = _L.lib.riab_plan_add(self._h, pop)
            if idx < 0:
                raise _L.RiabError(f"riab_plan_add failed: {_L.strerror(idx)}")
            self._pops.append(pop)
        self._pending = 0        # steps taken since the last sync()
        self._times_pending = []
        self._rows_open = 0      # rows still free in the attached chunks
        self._chunk_rows_done = 0
        self._agent_rows = None
        self._pop_rows = [None] * len(self.neurons)
        self._scratch_rates = [None] * len(self.neurons)
        agent._plan = self
        self._attach()

    # ---- history chunks --------------------------------------------------------------------------
    def _attach(self):
        """Open `capacity` fresh rows in every history and hand their base pointers to the plan."""
        self.sync()
        ag = self.agent
        cap = self.capacity
        if ag.save_history:
            self._agent_rows = ag._hist.open_rows(cap)
            _L.check(_L.lib.riab_plan_set_agent_history(self._h, _L.ptr(self._agent_rows), cap), "riab_plan_set_agent_history")
        else:
            self._agent_rows = None
            _L.check(_L.lib.riab_plan_set_agent_history(self._h, None, 0), "riab_plan_set_agent_history")
        for i, N in enumerate(self.neurons):
            if N.save_history:
                fr = N._hist_fr.open_rows(cap)
                sp = N._hist_sp.open_rows(cap) if N.save_spikes else None
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

    # ---- stepping --------------------------------------------------------------------------------
    def step(self, n_steps=1, drift_velocity=None, drift_to_random_strength_ratio=1, dt=None):
        ag = self.agent
        if ag._plan is not self:
            raise RuntimeError("this plan was closed (the agent was stepped eagerly or its history reset)")
        dt = dt or ag.dt
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
            self._attach()
        rc = _L.lib.riab_plan_step(self._h, int(n_steps), _L.current_stream())
        _L.check(rc, "riab_plan_step")
        self._rows_open -= n_steps
        self._pending += n_steps
        for _ in range(n_steps):
            ag.prev_t = ag.t
            ag.t += dt
            self._times_pending.append(ag.t)
        ag._step_index += n_steps

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
            _L.lib.riab_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _L.lib.riab_plan_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass

"""`Neurons`, `PlaceCells`, `GridCells`, `BoundaryVectorCells`,
`HeadDirectionCells` — batched, device-resident drop-ins for the firing-rate
path of the reference (reference ratinabox/Neurons.py).

Surface kept from the reference: `Cells(Agent, params)`, `update(**kwargs)`,
`get_state(evaluate_at="agent"|"all"|None, pos=..., head_direction=...)` returning
`(n, P)`, attributes `firingrate, n, noise, history{"t","firingrate","spikes"}`,
`get_history_arrays()`, `reset_history()`, `default_params` /
`get_all_default_params()`, and the per-class tuning attributes users edit
(`place_cell_centres`, `place_cell_widths`, `gridscales`, `phase_offsets`, `w`,
`tuning_distances`, ..., `preferred_angles`).  Device tables are rebuilt whenever
those attributes changed (content hash), so `PCs.place_cell_centres[-1] = ...`
works as in the reference (tests/test_advanced.py:59).

Every `get_state` evaluates on the GPU through the C ABI (include/riab_hip.h);
init-time parameter sampling stays in NumPy with the reference's draw order.
With `Agent.n_agents == B > 1`, `firingrate` is `(n, B)` and
`history["firingrate"]` `(T, n, B)`."""
import copy
import warnings

import numpy as np
import torch

from . import _lib, utils
from ._history import DeviceHistory, HistoryView

_Tensor = torch.Tensor

_L = _lib
LOG2E = 1.4426950408889634


def _pad4(n):
    return (int(n) + 3) // 4 * 4


class Neurons:
    default_params = {
        "n": 10,
        "name": "Neurons",
        "color": None,
        "noise_std": 0,
        "noise_coherence_time": 0.5,
        "min_fr": 0.0,
        "max_fr": 1.0,
        "save_history": True,
        # --- batched extension (not in the reference) ---
        "save_spikes": True,  # False: skip the Poisson-spike draw and its history (rates only)
    }

    # which per-agent direction rows feed io.hd_x / hd_y: (history-record rows, float64 state rows)
    _H_DIR = (_L.H_HD_X, _L.H_HD_Y)
    _S_DIR = (_L.S_HD_X, _L.S_HD_Y)
    # populations the rate stage of the flag-coupled pipeline covers (the one-kernel form of riab_simulate) set this
    _stream_kind = None
    # get_state() through the registered PyTorch operator (ops.py: torch.ops.riab.*): set by the classes that have one.
    # Called with float32 rows `d [4, P]` = (pos x, pos y, direction x, direction y); returns rates `[n, P]`.
    _state_op = None

    # Agent._fast_record: the float64 array attributes `_call` builds this population's device tables from, and the scalar
    # parameters it reads besides them — ONLY on the classes listed in FAST_REPEAT_TYPES (exact types: a subclass may
    # read more), whose `_call` reads nothing else
    _watch_arrays = None
    _watch_scalars = ()
    _COMMON_SCALARS = ("n", "min_fr", "max_fr", "noise_std", "noise_coherence_time", "save_history", "save_spikes")

    @classmethod
    def _fast_getter(cls):
        g = cls.__dict__.get("_fast_getter_cached")
        if g is None:
            import operator
            g = operator.attrgetter(*(cls._COMMON_SCALARS + tuple(cls._watch_scalars)))
            cls._fast_getter_cached = g
        return g

    def _auto_key(self):
        """What `update()` depends on besides the agent's state, by VALUE (users edit tuning arrays in place,
        reference tests/test_advanced.py:59): compared on every update() served by plan.AutoStepper."""
        f = self._call(None, None)  # (content-keyed device tables: identical objects while nothing changed)
        return (tuple([id(v) if type(v) is _Tensor else v for v in f.values()]), float(self.min_fr), float(self.max_fr),
                self.noise_std, self.noise_coherence_time, bool(self.save_history), bool(self.save_spikes))

    def _env_op_args(self):
        """(walls tensor or None, env list, periodic) of the Environment for the operators."""
        return self.Agent.Environment.op_env_args(self._device)

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.Agent.Neurons.append(self)
        self.pop_id = len(self.Agent.Neurons) - 1  # keys this population's RNG stream

        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        utils.update_class_params(self, self.params, get_all_defaults=True)
        utils.check_params(self, params.keys())

        self._device = Agent._device
        self._B, self._Bp = Agent._B, Agent._Bp
        self._table_cache = {}
        self._alloc_state()

    def _alloc_state(self):
        n = int(self.n)
        self._rates = torch.zeros((n, self._Bp), dtype=torch.float32, device=self._device)
        self._noise = torch.zeros((n, self._Bp), dtype=torch.float32, device=self._device)
        self._spikes_last = None
        self._hist_fr = DeviceHistory((n, self._Bp), torch.float32, self._device)
        self._hist_sp = DeviceHistory((n, self._Bp), torch.uint8, self._device)
        self._times = []
        self.history = HistoryView(("t", "firingrate", "spikes"), self._materialise_history,
                                   lambda: (self.Agent._sync_plan(), self._hist_fr.version)[1])

    @classmethod
    def get_all_default_params(cls, verbose=False):
        all_params = utils.collect_all_params(cls, dict_name="default_params")
        if verbose:
            import pprint
            pprint.pprint(all_params)
        return all_params

    def _population(self, plan_index=None):
        """This population as the ABI's RiabPopulation (for a step plan); keeps its tables alive.
        `plan_index`: {Neurons object: index in the plan} of the populations recorded before this one
        (what a FeedForwardLayer's inputs are looked up in)."""
        if not isinstance(self, FeedForwardLayer):
            # (the descriptor only changes when a table was rebuilt: simulate() of a short run asks for it every time,
            # and building the ctypes struct costs ~50 us)
            f = self._call(None, None)
            vals = tuple(f.values())
            hit = self.__dict__.get("_pop_cache")
            if hit is not None and self.noise_std == 0 and hit[1] == (float(self.min_fr), float(self.max_fr)) and \
                    len(hit[0]) == len(vals) and all(a is b if torch.is_tensor(a) else a == b for a, b in zip(hit[0], vals)):
                return hit[2]
        pop = _L.RiabPopulation()
        pop.n = int(self.n)
        pop.io.min_fr, pop.io.max_fr, pop.io.pop_id = float(self.min_fr), float(self.max_fr), int(self.pop_id)
        if isinstance(self, FeedForwardLayer):
            entries = list(self.inputs.values())
            if len(entries) > 8:
                raise ValueError("a FeedForwardLayer in a step plan takes at most 8 input layers")
            pop.kind, pop.n_inputs = _L.POP_KINDS["ff"], len(entries)
            keep = []
            for l, e in enumerate(entries):
                if plan_index is None or e["layer"] not in plan_index:
                    raise NotImplementedError(f"input layer {e['layer'].name} must be recorded in the plan before "
                                              f"{self.name} (recurrent inputs advance through update())")
                wt = self._device_weights(e)
                pop.input_index[l], pop.input_wt[l] = plan_index[e["layer"]], wt.data_ptr()
                keep.append(wt)
            bias = np.asarray(self.biases, dtype=np.float32).reshape(-1)
            bias_t = self._tables((bias,), lambda: torch.from_numpy(bias.copy()).to(self._device))
            act, pars = self._activation()
            pop.bias, pop.activation, pop.rates_prime = bias_t.data_ptr(), act, self._rates_prime.data_ptr()
            for i in range(4):
                pop.act_params[i] = pars[i]
            self._plan_tables = keep + [bias_t]
        else:
            pop.kind = f["kind"]
            self._plan_tables = [v for v in f.values() if torch.is_tensor(v)]
            for k, v in f.items():
                if k != "kind":
                    setattr(pop, k, v.data_ptr() if torch.is_tensor(v) else v)
            if self.noise_std == 0:
                self.__dict__["_pop_cache"] = (vals, (float(self.min_fr), float(self.max_fr)), pop)
        if self.noise_std != 0:  # the OU parameters of update() (Neurons.py:153-168), fixed for the plan's dt
            tau, dt = float(self.noise_coherence_time), float(self.Agent.dt)
            pop.noise_state = self._noise.data_ptr()
            pop.noise_theta_dt = dt / tau
            pop.noise_sigma_dt = float(np.sqrt((2 * float(self.noise_std) ** 2) / (tau * dt))) * dt
        return pop

    def _plan_scratch(self, pop):
        """What a step plan needs of this population besides its tables (called once per plan, with the plan's
        RiabPopulation): nothing by default."""

    def return_list_of_neurons(self, chosen_neurons="all"):
        """Indices of a selection of cells: "all", an int or digit string (that many, evenly spread),
        "<k>rand" (k at random), or an explicit list / array (reference Neurons.py:779-810)."""
        if isinstance(chosen_neurons, str):
            if chosen_neurons == "all":
                chosen_neurons = np.arange(self.n)
            elif chosen_neurons.isdigit():
                chosen_neurons = np.linspace(0, self.n - 1, min(self.n, int(chosen_neurons))).astype(int)
            elif chosen_neurons[-4:] == "rand":
                chosen_neurons = np.random.choice(np.arange(self.n), size=int(chosen_neurons[:-4]), replace=False)
        if type(chosen_neurons) is int:
            chosen_neurons = np.linspace(0, self.n - 1, min(self.n, chosen_neurons))
        if isinstance(chosen_neurons, list):
            chosen_neurons = list(np.array(chosen_neurons).astype(int))
        if isinstance(chosen_neurons, np.ndarray):
            chosen_neurons = list(chosen_neurons.astype(int))
        return chosen_neurons

    # ---- attributes -----------------------------------------------------------------------------
    @property
    def firingrate(self):
        self.Agent._sync_plan()
        self.Agent._settle_plan()
        self.Agent._check_pipeline()
        a = self._rates[:, :self._B].cpu().numpy().astype(np.float64)
        return a[:, 0] if self._B == 1 else a

    @property
    def noise(self):
        a = self._noise[:, :self._B].cpu().numpy().astype(np.float64)
        return a[:, 0] if self._B == 1 else a

    @property
    def firingrate_tensor(self):
        """Device firing rates of the last update: float32 `[n, B_padded]` (checked like Agent.get_history_tensor)."""
        self.Agent._sync_plan()
        self.Agent._check_pipeline()
        return self._rates

    # ---- the reference's per-step entry point (Neurons.py:145-171) ---------------------------
    def update(self, **kwargs):
        """firingrate = get_state() (+ OU noise when noise_std > 0); append t, rates and
        Poisson spikes `U(0,1) < dt*rate` to the history.  kwargs: `spike_uniforms=`
        `(n, B)` and `noise_normals=` `(n, B)` replace the in-kernel Philox draws."""
        Ag = self.Agent
        st = Ag._plan
        if st is not None:
            if not kwargs and st.__class__.__name__ == "AutoStepper" and st.step_population(self):
                return  # served from the native plan of the unchanged per-step loop (plan.AutoStepper)
            st.close()  # eager stepping resumes: the plan's open rows and cursors would go stale
        u = kwargs.pop("spike_uniforms", None)
        zn = kwargs.pop("noise_normals", None)
        save = bool(self.save_history)
        need_noise = self.noise_std != 0
        if save:
            rates = self._hist_fr.reserve(1)
            spikes = self._hist_sp.reserve(1) if self.save_spikes else None
        else:
            rates, spikes = self._rates.unsqueeze(0), None
        u_t = None if u is None else self._as_rows(u, torch.float32).unsqueeze(0)
        # spikes are drawn on the final rate: fuse them only when no noise is added afterwards
        io_spikes = spikes if not need_noise else None
        row = Ag._last_row
        if row is not None:
            # the motion kernel already left this step's fp32 positions / head directions
            self._launch(row[_L.H_POS_X], row[_L.H_POS_Y], row[self._H_DIR[0]], row[self._H_DIR[1]], pos_ld=self._Bp, T=1,
                         B=self._Bp, rates=rates, spikes=io_spikes, u_in=u_t if not need_noise else None,
                         dt=float(Ag.dt), step0=Ag._step_index)
        else:
            st = Ag._state
            self._launch(st[_L.S_POS_X], st[_L.S_POS_Y], st[self._S_DIR[0]], st[self._S_DIR[1]], pos_ld=self._Bp, T=1,
                         B=self._Bp, rates=rates, spikes=io_spikes, u_in=u_t if not need_noise else None,
                         dt=float(Ag.dt), step0=Ag._step_index, from_f64=True)
        if need_noise:
            tau = float(self.noise_coherence_time)
            sigma = float(np.sqrt((2 * float(self.noise_std) ** 2) / (tau * Ag.dt)))
            z_t = None if zn is None else self._as_rows(zn, torch.float32)
            rc = _L.lib.riab_neuron_noise(_L.ptr(self._noise), _L.ptr(rates), _L.ptr(z_t), int(self.n), self._Bp, 1,
                                          float(Ag.dt / tau), float(sigma * Ag.dt), int(Ag.rng_seed),
                                          int(Ag._step_index), int(self.pop_id), int(Ag.agent_id0),
                                          _L.current_stream())
            _L.check(rc, "riab_neuron_noise")
            if spikes is not None:
                self._spike_pass(rates, spikes, u_t, float(Ag.dt), Ag._step_index)
        self._rates = rates[0]
        if save:
            self._spikes_last = None if spikes is None else spikes[0]
            self._times.append(Ag.t)

    def _spike_pass(self, rates, spikes, u_t, dt, step0):
        io = self._io(None, None, None, None, self._Bp, rates.shape[0], self._Bp, rates, spikes, u_t, dt, step0)
        _L.check(_L.lib.riab_spikes(io, int(self.n), _L.current_stream()), "riab_spikes")

    def _as_rows(self, x, dtype):
        """array `(n, B)` -> device `[n, Bp]`."""
        t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
        t = t.to(self._device, dtype).reshape(int(self.n), -1)
        if t.shape[1] != self._Bp:
            pad = t[:, :1].expand(-1, self._Bp - t.shape[1])
            t = torch.cat((t, pad), dim=1)
        return t.contiguous()

    # ---- evaluation at arbitrary positions (Neurons.py:943-949 etc.) ---------------------------
    def get_state(self, evaluate_at="agent", **kwargs):
        """Firing rates `(n, P)` float64 on the host (like the reference).  evaluate_at:
        "agent" (current agent positions, P = n_agents), "all" (the environment's
        discretised coordinates) or None with `pos=(P,2)` (+ `head_direction=(P,2)`)."""
        return self.get_state_tensor(evaluate_at, **kwargs)[:, :self._last_P].cpu().numpy().astype(np.float64)

    def get_head_direction_averaged_state(self, evaluate_at="agent", angular_resolution_degrees=10, **kwargs):
        """get_state() averaged over head directions 0..2pi (reference Neurons.py:176-192); only differs
        from get_state() for cells tuned to the head direction (HeadDirectionCells, egocentric cells)."""
        if evaluate_at == "agent":  # evaluate at the agents' positions, but with the head directions imposed
            kwargs = dict(kwargs, pos=np.asarray(self.Agent.pos, dtype=np.float64).reshape(-1, 2))
            evaluate_at = None
        n_angles = int(360 / angular_resolution_degrees)
        total = None
        for ang in np.linspace(0, 2 * np.pi, n_angles):
            fr = self.get_state(evaluate_at=evaluate_at, **dict(kwargs, head_direction=np.array([np.cos(ang), np.sin(ang)])))
            total = fr if total is None else total + fr
        return total / n_angles

    def get_state_tensor(self, evaluate_at="agent", **kwargs):
        """As get_state but returns the device tensor float32 `[n, P_padded]`."""
        Ag = self.Agent
        if evaluate_at == "agent":
            st = Ag._state
            P = self._Bp
            self._last_P = self._B
            px, py, hx, hy = (st[_L.S_POS_X], st[_L.S_POS_Y], st[self._S_DIR[0]], st[self._S_DIR[1]])
            if self._state_op is not None and self._device.type == "cuda":
                return self._state_op(torch.stack((px, py, hx, hy)).to(torch.float32))
            out = torch.empty((1, int(self.n), P), dtype=torch.float32, device=self._device)
            self._launch(px, py, hx, hy, pos_ld=P, T=1, B=P, rates=out, spikes=None, u_in=None, dt=float(Ag.dt),
                         step0=0, from_f64=True)
            return out[0]
        if evaluate_at == "all":
            pos = Ag.Environment.flattened_discrete_coords
        else:
            pos = kwargs["pos"]
        pos = np.asarray(pos, dtype=np.float64).reshape(-1, 2)
        P = pos.shape[0]
        Pp = _pad4(P)
        self._last_P = P
        if P == 0:  # no positions: an (n, 0) result, nothing to launch (the reference's broadcasting gives the same shape)
            return torch.empty((int(self.n), 0), dtype=torch.float32, device=self._device)
        buf = np.zeros((4, Pp), dtype=np.float32)
        buf[0, :P], buf[1, :P] = pos[:, 0], pos[:, 1]
        buf[0, P:], buf[1, P:] = pos[0, 0], pos[0, 1]
        hd = kwargs.get("head_direction", kwargs.get("vel", None))
        if hd is None:
            buf[2], buf[3] = 1.0, 0.0  # the reference's default direction [1, 0]
        else:
            hd = np.asarray(hd, dtype=np.float64).reshape(-1, 2)
            hd = np.broadcast_to(hd, (P, 2)) if hd.shape[0] == 1 else hd
            buf[2, :P], buf[3, :P] = hd[:, 0], hd[:, 1]
            buf[2, P:], buf[3, P:] = hd[0, 0], hd[0, 1]
        d = torch.from_numpy(buf).to(self._device)
        if self._state_op is not None and self._device.type == "cuda":
            return self._state_op(d)  # torch.ops.riab.* (ops.py)
        out = torch.empty((1, int(self.n), Pp), dtype=torch.float32, device=self._device)
        self._launch(d[0], d[1], d[2], d[3], pos_ld=Pp, T=1, B=Pp, rates=out, spikes=None, u_in=None,
                     dt=float(Ag.dt), step0=0, from_f64=False)
        return out[0]

    # ---- plumbing ------------------------------------------------------------------------------
    def _io(self, px, py, hx, hy, pos_ld, T, B, rates, spikes, u_in, dt, step0):
        """Fill the ABI's RiabRateIO (one persistent struct per population; the library reads it
        before the call returns)."""
        io = self.__dict__.get("_io_struct")
        if io is None:
            io = self.__dict__["_io_struct"] = _L.RiabRateIO()
        io.pos_x = px.data_ptr() if px is not None else None
        io.pos_y = py.data_ptr() if py is not None else None
        io.hd_x = hx.data_ptr() if hx is not None else None
        io.hd_y = hy.data_ptr() if hy is not None else None
        io.pos_ld, io.T, io.B = pos_ld, T, B
        io.rates = rates.data_ptr()
        io.spikes = spikes.data_ptr() if spikes is not None else None
        io.u_in = u_in.data_ptr() if u_in is not None else None
        io.dt = dt
        io.min_fr, io.max_fr = self.min_fr, self.max_fr
        io.seed = self.Agent.rng_seed
        io.step0 = step0
        io.agent_id0 = self.Agent.agent_id0
        io.pop_id = self.pop_id
        return io

    def _launch(self, px, py, hx, hy, pos_ld, T, B, rates, spikes, u_in, dt, step0, from_f64=False, stream=None):
        """Run this population's rate kernel.  Positions are fp32 rows on device; the
        agent state is float64, so `from_f64` first rounds the four rows to fp32."""
        if from_f64:
            f = torch.stack((px, py, hx, hy)).to(torch.float32)
            px, py, hx, hy = f[0], f[1], f[2], f[3]
            pos_ld = f.shape[1]
        io = self._io(px, py, hx, hy, pos_ld, T, B, rates, spikes, u_in, dt, step0)
        self._keep = (px, py, hx, hy, rates, spikes, u_in)
        self._call(io, _L.current_stream() if stream is None else stream)

    def _call(self, io, stream):
        raise NotImplementedError("Neurons object needs a get_state() method")

    def _tables(self, key_arrays, build):
        """Device tables cached on the content of the host attributes they derive from."""
        key = tuple(np.ascontiguousarray(a).tobytes() if isinstance(a, np.ndarray) else a for a in key_arrays)
        hit = self._table_cache.get("t")
        if hit is not None and hit[0] == key:
            return hit[1]
        tabs = build()
        self._table_cache["t"] = (key, tabs)
        return tabs

    # ---- fused path hooks (called by Agent.simulate) ---------------------------------------------
    def _reserve_rows(self, n_steps, ring):
        n = int(self.n)
        if self.save_history:
            return dict(fr=self._hist_fr.reserve(n_steps),
                        sp=self._hist_sp.reserve(n_steps) if self.save_spikes else None, ring=None)
        rows = min(n_steps, 2 * ring)
        return dict(fr=torch.empty((rows, n, self._Bp), dtype=torch.float32, device=self._device), sp=None,
                    ring=rows)

    def _reserve_rows_at(self, n_steps, ring):
        """_reserve_rows without making the views (each a torch slicing call of a few microseconds, and simulate() of
        a short run is in a hurry to launch): (rates chunk, first row, spikes chunk or None, first row, ring rows or
        None); `_rows_views` turns it into the dict the other hooks take."""
        if self.save_history:
            fr_c, fr_s = self._hist_fr.reserve_at(n_steps)
            sp_c, sp_s = self._hist_sp.reserve_at(n_steps) if self.save_spikes else (None, 0)
            return fr_c, fr_s, sp_c, sp_s, None
        rows = min(n_steps, 2 * ring)
        return torch.empty((rows, int(self.n), self._Bp), dtype=torch.float32, device=self._device), 0, None, 0, rows

    @staticmethod
    def _rows_views(at, n_steps):
        fr_c, fr_s, sp_c, sp_s, ring = at
        if ring is not None:
            return dict(fr=fr_c, sp=None, ring=ring)
        return dict(fr=fr_c[fr_s:fr_s + n_steps], sp=None if sp_c is None else sp_c[sp_s:sp_s + n_steps], ring=None)

    def _unreserve_rows(self, out, n_steps):
        if out["ring"] is None:
            self._hist_fr.unreserve(n_steps)
            if out["sp"] is not None:
                self._hist_sp.unreserve(n_steps)

    def _rates_from_trajectory(self, traj, out, t0, tc, step0, dt, stream):
        """Rates (+ spikes) for trajectory rows `traj [tc, 8, Bp]`, written to rows
        t0..t0+tc of the reserved output."""
        Bp = self._Bp
        fr, sp = self._chunk_view(out, t0, tc)
        ld = _L.HIST_ROWS * Bp
        hook = getattr(self.Agent, "_profile_hook", None)
        if hook is not None:
            hook(self, "begin", tc)
        noisy = self.noise_std != 0
        self._launch(traj[0, _L.H_POS_X], traj[0, _L.H_POS_Y], traj[0, self._H_DIR[0]], traj[0, self._H_DIR[1]], pos_ld=ld,
                     T=tc, B=Bp, rates=fr, spikes=None if noisy else sp, u_in=None, dt=dt, step0=step0 + 1,
                     stream=stream)
        if hook is not None:
            hook(self, "end", tc)
        if noisy:
            # rates -> + OU noise (sequential over the chunk's rows) -> spikes on the noisy rates
            tau = float(self.noise_coherence_time)
            sigma = float(np.sqrt((2 * float(self.noise_std) ** 2) / (tau * dt)))
            Ag = self.Agent
            rc = _L.lib.riab_neuron_noise(_L.ptr(self._noise), _L.ptr(fr), None, int(self.n), Bp, int(tc),
                                          float(dt / tau), float(sigma * dt), int(Ag.rng_seed), int(step0 + 1),
                                          int(self.pop_id), int(Ag.agent_id0), stream)
            _L.check(rc, "riab_neuron_noise")
            if sp is not None:
                io = self._io(None, None, None, None, Bp, tc, Bp, fr, sp, None, dt, step0 + 1)
                _L.check(_L.lib.riab_spikes(io, int(self.n), stream), "riab_spikes")

    def _chunk_view(self, out, t0, tc):
        """(rates, spikes) rows of the chunk [t0, t0+tc) inside a reserved output."""
        if out["ring"] is None:
            return out["fr"][t0:t0 + tc], (None if out["sp"] is None else out["sp"][t0:t0 + tc])
        r0 = t0 % out["ring"]
        if r0 + tc > out["ring"]:
            r0 = 0
        fr = out["fr"][r0:r0 + tc]
        out["last"] = fr[tc - 1]
        return fr, None

    def _finish_rows(self, out, n_steps, times):
        if out["ring"] is None:
            self._rates = out["fr"][n_steps - 1]
            self._spikes_last = None if out["sp"] is None else out["sp"][n_steps - 1]
            self._times.extend(times)
        else:
            self._rates = out["last"]

    # ---- history ---------------------------------------------------------------------------------
    def _materialise_history(self):
        self.Agent._sync_plan()
        self.Agent._settle_plan()
        self.Agent._check_pipeline()
        fr = self._hist_fr.stack()[:, :, :self._B].cpu().numpy()
        sp = self._hist_sp.stack()[:, :, :self._B].cpu().numpy().astype(bool)
        if len(sp) == 0:
            sp = np.zeros((0,) + fr.shape[1:], dtype=bool)
        if self._B == 1:
            fr, sp = fr[:, :, 0], sp[:, :, 0]
        return {"t": np.array(self._times, dtype=float), "firingrate": fr, "spikes": sp}

    def get_history_arrays(self):
        return dict(self.history.items())

    def get_history_tensors(self):
        """(firingrate float32 [T, n, Bp], spikes uint8 [T, n, Bp]) on device (checked like Agent.get_history_tensor)."""
        self.Agent._sync_plan()
        self.Agent._check_pipeline()
        return self._hist_fr.stack(), self._hist_sp.stack()

    def reset_history(self):
        if self.Agent._plan is not None:
            self.Agent._plan.close()  # (its open rows live in the chunks dropped here)
        self.Agent._check_pipeline()   # (rows about to be dropped may still be owed a recovery)
        self._hist_fr.reset()
        self._hist_sp.reset()
        self._times = []

    def save_to_history(self):
        raise NotImplementedError("history rows are written by the kernels; there is no host-side append")


# ================================================================================================
class PlaceCells(Neurons):
    """Place cells: a function of the distance from the agent to each cell's centre
    (reference Neurons.py:827-981).  descriptions: gaussian (default),
    gaussian_threshold, diff_of_gaussians, top_hat, one_hot; wall geometries:
    euclidean, line_of_sight, geodesic."""

    _stream_kind = "place"
    _watch_arrays = ("place_cell_centres", "place_cell_widths")
    _watch_scalars = ("description", "wall_geometry", "widths")
    default_params = {
        "n": 10,
        "name": "PlaceCells",
        "description": "gaussian",
        "widths": 0.20,
        "place_cell_centres": None,
        "wall_geometry": "geodesic",
        "min_fr": 0,
        "max_fr": 1,
    }

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        pcc = self.params["place_cell_centres"]
        if pcc is None:
            self.params["place_cell_centres"] = Agent.Environment.sample_positions(
                n=self.params["n"], method="uniform_jitter")
        elif isinstance(pcc, str):
            if pcc not in ("random", "uniform", "uniform_jitter"):
                raise ValueError("self.params['place_cell_centres'] must be None, an array of locations or one of "
                                 "the instructions ['random', 'uniform', 'uniform_jitter']")
            self.params["place_cell_centres"] = Agent.Environment.sample_positions(n=self.params["n"], method=pcc)
        else:
            self.params["place_cell_centres"] = np.asarray(pcc, dtype=float)
            self.params["n"] = self.params["place_cell_centres"].shape[0]
        self.place_cell_widths = self.params["widths"] * np.ones(self.params["n"])
        super().__init__(Agent, self.params)

        Env = Agent.Environment
        if self.wall_geometry in ("line_of_sight", "geodesic") and Env.boundary_conditions == "periodic":
            print(f"{self.wall_geometry} wall geometry only possible in 2D when the boundary conditions are solid. "
                  "Using 'euclidean' instead.")
            self.wall_geometry = "euclidean"
        if self.wall_geometry == "geodesic" and len(Env.walls) > 5:
            print("'geodesic' wall geometry only supported for enivironments with 1 additional wall (4 bounding "
                  "walls + 1 additional). Sorry. Using 'line_of_sight' instead.")
            self.wall_geometry = "line_of_sight"

    def _call(self, io, stream):
        n = int(self.n)
        centres = np.asarray(self.place_cell_centres, dtype=np.float64).reshape(-1, 2)
        widths = np.asarray(self.place_cell_widths, dtype=np.float64)
        if widths.shape != (n,):
            widths = widths * np.ones(n)

        def build():
            tab = np.empty((n, 3), dtype=np.float64)
            tab[:, 0], tab[:, 1] = centres[:, 0], centres[:, 1]
            tab[:, 2] = -LOG2E / (2 * widths ** 2)
            return torch.from_numpy(tab.astype(np.float32)).to(self._device)

        tab = self._tables((centres, widths), build)
        geom = self.wall_geometry
        if geom == "geodesic" and len(self.Agent.Environment.walls) <= 4:
            geom = "euclidean"  # Environment.py:741-742
        thw = float(np.asarray(self.widths, dtype=float).reshape(-1)[0])
        if io is None:  # descriptor for a step plan
            return dict(kind=_L.POP_KINDS["place"], table=tab, description=_L.PC_DESCRIPTIONS[self.description],
                        geometry=_L.GEOMETRIES[geom], top_hat_width=thw)
        env, _w = self.Agent.Environment.device_tables(self._device)
        rc = _L.lib.riab_place_cells(env, io, _L.ptr(tab), n, _L.PC_DESCRIPTIONS[self.description],
                                     _L.GEOMETRIES[geom], thw, stream)
        _L.check(rc, "riab_place_cells")

    def _state_op(self, d):
        from . import ops  # noqa: F401  (registers torch.ops.riab.*)
        f = self._call(None, None)
        walls, env, periodic = self._env_op_args()
        return torch.ops.riab.place_cells(d[0:2], f["table"], walls, env, periodic, f["description"], f["geometry"],
                                          f["top_hat_width"], float(self.min_fr), float(self.max_fr))

    def remap(self):
        self.place_cell_centres = self.Agent.Environment.sample_positions(n=self.n, method="uniform_jitter")
        np.random.shuffle(self.place_cell_centres)


# ================================================================================================
class GridCells(Neurons):
    """Grid cells: rectified or shifted sum of three cosines 60 degrees apart
    (reference Neurons.py:1033-1256)."""

    _stream_kind = "grid"
    _watch_arrays = ("gridscales", "phase_offsets", "w")
    _watch_scalars = ("description", "width_ratio")
    default_params = {
        "n": 30,
        "gridscale_distribution": "modules",
        "gridscale": (0.3, 0.5, 0.8),
        "orientation_distribution": "modules",
        "orientation": (0, 0.1, 0.2),
        "phase_offset_distribution": "uniform",
        "phase_offset": (0, 2 * np.pi),
        "description": "rectified_cosines",
        "width_ratio": 4 / (3 * np.sqrt(3)),
        "min_fr": 0,
        "max_fr": 1,
        "name": "GridCells",
    }

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        p = self.params
        if p["description"] in ("three_rectified_cosines", "three_shifted_cosines"):
            p["description"] = p["description"][6:]
            warnings.warn(f"the 'three_' prefix on the 'description' parameter is deprecated, in the future please "
                          f"use '{p['description']}' instead")
        if isinstance(p["gridscale"], (list, np.ndarray)):
            self.gridscales = np.array(p["gridscale"])
            p["n"] = len(self.gridscales)
        else:
            self.gridscales = utils.distribution_sampler(p["gridscale_distribution"], p["gridscale"], shape=(p["n"],))
        super().__init__(Agent, p)
        if isinstance(p["phase_offset"], (list, np.ndarray)) and np.array(p["phase_offset"]).ndim == 2:
            self.phase_offsets = np.array(p["phase_offset"])
            assert len(self.phase_offsets) == p["n"], "number of phase offsets supplied incompatible with number of neurons"
        elif p["phase_offset_distribution"] == "grid":
            self.phase_offsets = self.set_phase_offsets_on_grid()
        else:
            self.phase_offsets = utils.distribution_sampler(p["phase_offset_distribution"], p["phase_offset"],
                                                            shape=(p["n"], 2))
        if isinstance(p["orientation"], (list, np.ndarray)):
            self.orientations = np.array(p["orientation"])
            assert len(self.orientations) == p["n"], "number of orientations supplied incompatible with number of neurons"
        else:
            self.orientations = utils.distribution_sampler(p["orientation_distribution"], p["orientation"],
                                                           shape=(p["n"],))
        w = []
        for i in range(self.n):
            w1 = utils.rotate(np.array([1, 0]), self.orientations[i])
            w.append(np.array([w1, utils.rotate(w1, np.pi / 3), utils.rotate(w1, 2 * np.pi / 3)]))
        self.w = np.array(w)
        if self.description == "rectified_cosines":
            assert 0 < self.width_ratio <= 1, "width_ratio must be between 0 and 1"

    def set_phase_offsets_on_grid(self):
        """Phase offsets tiling [0, 2pi)^2 on an n_x x n_y grid, the remaining cells uniform at random
        (reference Neurons.py:1238-1256)."""
        n_x = int(np.sqrt(self.n))
        n_y = self.n // n_x
        n_remaining = self.n - n_x * n_y
        dx, dy = 2 * np.pi / n_x, 2 * np.pi / n_y
        grid = np.mgrid[(0 + dx / 2):(2 * np.pi - dx / 2):(n_x * 1j), (0 + dy / 2):(2 * np.pi - dy / 2):(n_y * 1j)]
        grid = grid.reshape(2, -1).T
        remaining = np.random.uniform(0, 2 * np.pi, size=(n_remaining, 2))
        return np.vstack([grid, remaining])

    def _call(self, io, stream):
        n = int(self.n)
        gs = np.asarray(self.gridscales, dtype=np.float64)
        ph = np.asarray(self.phase_offsets, dtype=np.float64)
        w = np.asarray(self.w, dtype=np.float64)

        def build():
            # phase_i / 2pi = (origin - p) . w_i / lambda = a_i - (x bx_i + y by_i)   (Neurons.py:1192-1203)
            origin = gs.reshape(-1, 1) * ph / (2 * np.pi)
            tab = np.empty((n, 9), dtype=np.float64)
            for i in range(3):
                a = (origin[:, 0] * w[:, i, 0] + origin[:, 1] * w[:, i, 1]) / gs
                tab[:, 3 * i] = a - np.floor(a)
                tab[:, 3 * i + 1] = w[:, i, 0] / gs
                tab[:, 3 * i + 2] = w[:, i, 1] / gs
            return torch.from_numpy(tab.astype(np.float32)).to(self._device)

        tab = self._tables((gs, ph, w), build)
        f0 = (1 / 3) * (2 * np.cos(np.sqrt(3) * np.pi * self.width_ratio / 2) + 1)
        if io is None:
            return dict(kind=_L.POP_KINDS["grid"], table=tab, description=_L.GC_DESCRIPTIONS[self.description],
                        f0=float(f0))
        rc = _L.lib.riab_grid_cells(io, _L.ptr(tab), n, _L.GC_DESCRIPTIONS[self.description], float(f0), stream)
        _L.check(rc, "riab_grid_cells")

    def _state_op(self, d):
        from . import ops  # noqa: F401  (registers torch.ops.riab.*)
        f = self._call(None, None)
        return torch.ops.riab.grid_cells(d[0:2], f["table"], f["description"], f["f0"], float(self.min_fr), float(self.max_fr))


# ================================================================================================
class VectorCells(Neurons):
    """Base of the vector-cell family: preferred distance / angle tuning per cell
    (reference Neurons.py:1259-1437).  Only BoundaryVectorCells is accelerated."""

    default_params = {
        "n": 10,
        "reference_frame": "allocentric",
        "cell_arrangement": "random",
        "tuning_distance_distribution": "uniform",
        "tuning_distance": (0.05, 0.3),
        "sigma_distance_distribution": "diverging",
        "sigma_distance": (0.08, 12),
        "tuning_angle_distribution": "uniform",
        "tuning_angle": (0.0, 360),
        "angular_spread_distribution": "uniform",
        "angular_spread": (10, 30),
    }

    def __init__(self, Agent, params={}):
        if type(self) is VectorCells:
            raise RuntimeError("Cannot instantiate VectorCells on their own. Must be instantiated through one of the "
                               "subclasses, e.g. BoundaryVectorCells")
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        super().__init__(Agent, self.params)
        arr = self.cell_arrangement
        (self.tuning_distances, self.tuning_angles, self.sigma_distances,
         self.sigma_angles) = self.set_tuning_parameters(**self.params)
        manifold = isinstance(arr, str) and arr.endswith("manifold")
        if getattr(self, "_warn_if_n_changes", False) and (manifold or self.n != len(self.tuning_distances)):
            warnings.warn(f"Ignoring 'n' parameter value ({params['n']}) that was passed, and setting number of "
                          f"{self.name} neurons to {len(self.tuning_distances)}, inferred from the cell arrangement.")
        self.n = len(self.tuning_distances)
        self._alloc_state()


    def set_tuning_parameters(self, **kwargs):
        """The cells' tuning (reference VectorCells.set_tuning_parameters, Neurons.py:1388-1437): four arrays of equal
        length — tuning distances, tuning angles (rad), distance sigmas, angle sigmas — from `cell_arrangement`:
        "random…" / None (`utils.create_random_assembly`), "uniform_manifold", "diverging_manifold", or a function of
        the keyword arguments returning the four lists.  Returns them; the constructor stores them."""
        arr = self.cell_arrangement
        if callable(arr):
            tuning = arr(**kwargs)
        elif arr is None or arr[:6] == "random":
            tuning = utils.create_random_assembly(**kwargs)
        elif arr == "uniform_manifold":
            tuning = utils.create_uniform_radial_assembly(**kwargs)
        elif arr == "diverging_manifold":
            tuning = utils.create_diverging_radial_assembly(**kwargs)
        else:
            raise ValueError("cell_arrangement must be 'random', 'uniform_manifold', 'diverging_manifold' or a function")
        mu_d, mu_t, sg_d, sg_t = (np.array(v) for v in tuning)
        assert len(mu_d) == len(mu_t) == len(sg_d) == len(sg_t), "All manifold tuning parameters must be of the same length"
        return mu_d, mu_t, sg_d, sg_t


class BoundaryVectorCells(VectorCells):
    """Boundary vector cells (reference Neurons.py:1535-1778): for each of K test
    directions the distance to the first wall, weighted by a gaussian in distance and
    a von Mises in angle, summed over directions and normalised analytically."""

    # share of a cell's angular weight that its direction window may leave out (see _call): bounds the change of a
    # normalised rate, a tenth of the parity tolerance's floor
    BVC_WINDOW_SHARE = 1e-6

    default_params = {
        "n": 10,
        "name": "BoundaryVectorCells",
        "dtheta": 2,
        "max_fr": 1.0,
        "min_fr": 0.0,
    }

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        if not hasattr(self, "_warn_if_n_changes"):  # warn only when the USER passed an n that gets overridden
            self._warn_if_n_changes = "n" in params and params["n"] is not None
        super().__init__(Agent, self.params)
        assert self.Agent.Environment.boundary_conditions == "solid", \
            "boundary cells only possible with solid boundary conditions"
        # K = int(360/dtheta) angles [0] + [2 pi i dtheta / 360, i = 0..K-2]: 0 is duplicated and the last
        # angle missing, as in the reference (Neurons.py:1584-1596)
        self.n_test_angles = int(360 / self.dtheta)
        angles = [0.0] + [2 * np.pi * i * self.dtheta / 360 for i in range(self.n_test_angles - 1)]
        self.test_angles = np.array(angles)
        self.test_directions = np.array([utils.rotate(np.array([1, 0]), a) for a in angles])
        self.test_directions[0] = np.array([1.0, 0.0])
        kappa = 1 / np.asarray(self.sigma_angles, dtype=float).reshape(-1, 1) ** 2
        self.cell_fr_norm = np.exp(kappa * (np.cos(self.test_angles.reshape(1, -1)) - 1)).sum(axis=1)

    def boundary_vector_preference_function(self, x):
        """Preference of a ray for each wall from the pair of line parameters `x[..., :] = (l_a, l_b)` that
        `utils.vector_intercepts(rays, walls)` gives: `1 / l_a` (nearer is better) when the crossing lies ahead of the
        ray's origin and on the wall, -1 when it lies behind (`l_a < 0`) or off the wall (`l_b < 0` or `l_b > 1`)
        (reference BoundaryVectorCells.boundary_vector_preference_function, Neurons.py:1746-1778; the device form of
        this selection is stage A of csrc/riab_bvc.hip).  Host side, for user code; not on the accelerated path."""
        x = np.asarray(x, dtype=float)
        assert x.shape[-1] == 2
        la, lb = x[..., 0], x[..., 1]
        with np.errstate(divide="ignore"):
            pref = np.where(la > 0, 1 / la, np.where(la < 0, -1.0, 0.0))
        return np.where((lb < 0) | (lb > 1), -1.0, pref)

    def _call(self, io, stream):
        n, K = int(self.n), int(self.n_test_angles)
        mu_d = np.asarray(self.tuning_distances, dtype=np.float64)
        sg_d = np.asarray(self.sigma_distances, dtype=np.float64)
        mu_t = np.asarray(self.tuning_angles, dtype=np.float64)
        sg_t = np.asarray(self.sigma_angles, dtype=np.float64)
        ang = np.asarray(self.test_angles, dtype=np.float64)
        dirs = np.asarray(self.test_directions, dtype=np.float64)
        norm = np.asarray(self.cell_fr_norm, dtype=np.float64)
        ego = self.reference_frame == "egocentric"

        def build():
            a = np.sqrt(LOG2E / 2) / sg_d
            kappa = 1 / sg_t ** 2
            cells = np.zeros((4, n))
            cells[0], cells[1], cells[2] = a * mu_d, a, kappa * LOG2E
            diff = ang[None, :] - mu_t[:, None]
            # table rows padded to a multiple of 4.  The kernel gives the pad directions an infinite distance,
            # so their terms are exp2(-inf) = 0 whatever the table holds (an -inf COSINE entry times a negative
            # cos(head bearing) would be +inf: the egocentric pads are zeros)
            Kp = (K + 3) // 4 * 4
            if ego:
                vm = np.zeros((2, n, Kp))
                vm[0, :, :K], vm[1, :, :K] = np.cos(diff), np.sin(diff)
            else:
                vm = np.full((n, Kp), -np.inf)
                vm[:, :K] = LOG2E * kappa[:, None] * (np.cos(diff) - 1)
            # Direction windows (allocentric, K a multiple of 4).  A rate is sum_k g_k w_k / sum_k w_k with radial
            # factors g_k <= 1 and von Mises weights w_k (peak 1; the denominator is the reference's cell_fr_norm,
            # Neurons.py:1598-1604): leaving directions out changes it by at most the share of their weights in the
            # sum.  Each cell keeps its heaviest directions and drops the rest as long as the dropped share stays
            # below BVC_WINDOW_SHARE = 1e-6 — a tenth of the floor of the parity tolerance (1e-5 relative + 1e-5 of
            # the range), and for narrow tunings an arc of 4.9 sigma either side instead of the 5.8 sigma of the
            # first criterion (weight below 2^-24 of the peak).  The kernel accumulates four table rows at a time,
            # so rows are regrouped by (arc length, tuning angle) and every group of four gets the smallest
            # arc, in whole quads of directions, that holds all its cells' arcs; the rest is skipped.
            rows_t = win_t = None
            inv = 1 / norm
            if use_windows and not ego and K % 4 == 0:
                w_lin = np.exp2(vm[:, :K])
                idx = np.argsort(w_lin, axis=1)
                dropped = np.cumsum(np.take_along_axis(w_lin, idx, axis=1), axis=1) <= self.BVC_WINDOW_SHARE * w_lin.sum(axis=1)[:, None]
                keep = np.ones((n, K), dtype=bool)
                np.put_along_axis(keep, idx, ~dropped, axis=1)
                self._window_stats = dict(share=self.BVC_WINDOW_SHARE, cells_need=float(keep.mean()))
                if not keep.all():
                    band = np.minimum(keep.sum(axis=1) // 24, 7)
                    order = np.lexsort((np.mod(mu_t, 2 * np.pi), band))
                    cells, vm, inv = cells[:, order], vm[order], inv[order]
                    keep = keep[order]
                    win = np.zeros(((n + 3) // 4, 2), dtype=np.int32)
                    for g in range(len(win)):
                        u = keep[np.minimum(np.arange(4 * g, 4 * g + 4), n - 1)].any(axis=0)
                        gap_len, gap_start, run = 0, 0, 0
                        for k in range(2 * K):  # longest circular run of skippable directions
                            run = run + 1 if not u[k % K] else 0
                            if min(run, K) > gap_len:
                                gap_len, gap_start = min(run, K), (k - min(run, K) + 1) % K
                        first = (gap_start + gap_len) % K
                        k0 = first // 4 * 4
                        length = min(K, (first - k0 + (K - gap_len) + 3) // 4 * 4)
                        win[g] = (0, K) if length >= K else (k0, length)
                    rows_t = torch.from_numpy(order.astype(np.int32)).to(self._device)
                    win_t = torch.from_numpy(win).to(self._device)
                    self._window_stats["issued"] = float(win[:, 1].sum() * 4 / (len(win) * 4 * K))
            f32 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self._device)  # noqa: E731
            # position-independent denominators of the ray/wall intercepts (utils.py:96): sa . sb_p
            s_w = walls[:, 1, :] - walls[:, 0, :]
            with np.errstate(divide="ignore"):
                rden = 1.0 / (dirs[:, None, 0] * (-s_w[None, :, 1]) + dirs[:, None, 1] * s_w[None, :, 0])
            f64 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self._device)  # noqa: E731
            dirs_dev = dirs
            if os.environ.get("RIAB_EXP_NO_RAY_PAIRS"):   # (A/B: a table without exact opposites -> one ray at a time)
                dirs_dev = dirs.copy()
                dirs_dev[2, 0] += 1e-9
            return (f64(dirs_dev), f64(rden), f32(cells), f32(vm), f32(inv), rows_t, win_t)

        import os
        use_windows = os.environ.get("RIAB_NO_BVC_WINDOWS") is None  # (A/B switch: every direction for every cell)
        walls = np.asarray(self.Agent.Environment.walls, dtype=np.float64).reshape(-1, 2, 2)
        dirs_t, rden_t, cells_t, vm_t, inv_t, rows_t, win_t = self._tables(
            (mu_d, sg_d, mu_t, sg_t, ang, dirs, norm, ego, walls, use_windows), build)
        if io is None:
            d = dict(kind=_L.POP_KINDS["bvc"], table=cells_t, test_dirs=dirs_t, ray_rden=rden_t, K=K, vm_table=vm_t,
                     inv_norm=inv_t, egocentric=1 if ego else 0)
            if rows_t is not None:
                d.update(cell_rows=rows_t, windows=win_t)
            return d
        env, _w = self.Agent.Environment.device_tables(self._device)
        rc = _L.lib.riab_boundary_vector_cells_windowed(env, io, _L.ptr(dirs_t), _L.ptr(rden_t), K, _L.ptr(cells_t),
                                                        _L.ptr(vm_t), _L.ptr(inv_t), n, 1 if ego else 0, None,
                                                        _L.ptr(rows_t), _L.ptr(win_t), stream)
        _L.check(rc, "riab_boundary_vector_cells")

    def _plan_scratch(self, pop):
        """A step plan's one-row launches: scratch for the ray exchange between the workgroups that share a tile (riab_hip.h
        RiabPopulation.bvc_xch); the arrival counters start at zero with every plan."""
        tiles, Kp = (self._Bp + 63) // 64, (int(pop.K) + 3) // 4 * 4
        if getattr(self, "_bvc_xch", None) is None or tuple(self._bvc_xch.shape) != (tiles, Kp, 64):
            self._bvc_xch = torch.empty((tiles, Kp, 64), dtype=torch.float32, device=self._device)
            self._bvc_xch_count = torch.zeros(tiles, dtype=torch.int32, device=self._device)
        else:
            self._bvc_xch_count.zero_()
        pop.bvc_xch, pop.bvc_xch_count = self._bvc_xch.data_ptr(), self._bvc_xch_count.data_ptr()

    def _state_op(self, d):
        from . import ops  # noqa: F401  (registers torch.ops.riab.*)
        f = self._call(None, None)
        walls, env, periodic = self._env_op_args()
        ego = bool(f["egocentric"])
        return torch.ops.riab.boundary_vector_cells(d[0:2], d[2:4] if ego else None, walls, env, periodic, f["test_dirs"],
                                                    f["ray_rden"], f["table"], f["vm_table"], f["inv_norm"], ego,
                                                    f.get("cell_rows"), f.get("windows"), float(self.min_fr),
                                                    float(self.max_fr))


class FieldOfViewBVCs(BoundaryVectorCells):
    """Egocentric boundary vector cells tiling the agent's field of view in concentric rows
    (reference Neurons.py:1847-1888): a parameterisation of the egocentric BVC kernel.
    `cell_arrangement`: "diverging_manifold" (field size grows with distance, default) or
    "uniform_manifold"."""

    default_params = {
        "distance_range": [0.02, 0.4],
        "angle_range": [0, 75],
        "spatial_resolution": 0.02,
        "cell_arrangement": "diverging_manifold",
        "beta": 5,
        "color": [0.3, 0.3, 0.3, 1],   # (reference Neurons.py:1869)
    }

    def __init__(self, Agent, params={}):
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        self.params["reference_frame"] = "egocentric"
        assert self.params["cell_arrangement"] is not None, "cell_arrangement must be set for FoV Neurons"
        self._warn_if_n_changes = "n" in params and params["n"] is not None
        super().__init__(Agent, self.params)


class ObjectVectorCells(VectorCells):
    """Object vector cells (reference Neurons.py:1892-2116): gaussian in the distance and von
    Mises in the bearing to the objects of the cell's preferred type, summed over those objects;
    with `walls_occlude` an object behind a wall is not seen (line_of_sight geometry)."""

    default_params = {
        "n": 10,
        "name": "ObjectVectorCell",
        "walls_occlude": True,
        "object_tuning_type": "random",  # "random", an int, or a list / array of n ints
    }

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        if not hasattr(self, "_warn_if_n_changes"):
            self._warn_if_n_changes = "n" in params and params["n"] is not None
        super().__init__(Agent, self.params)
        self.object_locations = self.Agent.Environment.objects["objects"]
        if len(self.object_locations) == 0:
            raise RuntimeError(f"Cannot initialize {self.params['name']}, as there are no objects in the environment.")
        self.tuning_types = None
        self.set_tuning_types(self.object_tuning_type)
        self.wall_geometry = "line_of_sight" if self.walls_occlude else "euclidean"

    def set_tuning_types(self, tuning_types=None):
        """Preferred object type of every cell ("random": drawn from the types present)."""
        if isinstance(tuning_types, str) and tuning_types == "random":
            self.object_types = self.Agent.Environment.objects["object_types"]
            self.tuning_types = np.random.choice(np.unique(self.object_types), replace=True, size=(self.n,))
            return
        if isinstance(tuning_types, (int, np.integer)):
            tuning_types = np.repeat(tuning_types, self.n)
        elif isinstance(tuning_types, list):
            tuning_types = np.array(tuning_types)
        assert isinstance(tuning_types, np.ndarray), "tuning_types must be an integer, list or numpy array"
        assert tuning_types.shape[0] == self.n, f"Tuning types must be a vector of length ({self.n},)"
        self.tuning_types = tuning_types

    def _call(self, io, stream):
        n = int(self.n)
        Env = self.Agent.Environment
        objs = np.asarray(Env.objects["objects"], dtype=np.float64).reshape(-1, 2)
        otypes = np.asarray(Env.objects["object_types"], dtype=np.int64)
        mu_d = np.asarray(self.tuning_distances, dtype=np.float64)
        sg_d = np.asarray(self.sigma_distances, dtype=np.float64)
        mu_t = np.asarray(self.tuning_angles, dtype=np.float64)
        sg_t = np.asarray(self.sigma_angles, dtype=np.float64)
        ttypes = np.asarray(self.tuning_types, dtype=np.int64)
        occlude = self.wall_geometry == "line_of_sight"
        if occlude:
            assert Env.boundary_conditions == "solid", \
                "line of sight geometry not available for periodic boundary conditions"

        def build():
            a = np.sqrt(LOG2E / 2) / sg_d
            cells = np.stack((a * mu_d, a, np.cos(mu_t), np.sin(mu_t), LOG2E / sg_t ** 2, ttypes.astype(float)), axis=-1)
            return (torch.from_numpy(np.ascontiguousarray(objs, dtype=np.float32)).to(self._device),
                    torch.from_numpy(np.ascontiguousarray(otypes, dtype=np.int32)).to(self._device),
                    torch.from_numpy(np.ascontiguousarray(cells, dtype=np.float32)).to(self._device))

        objs_t, types_t, cells_t = self._tables((objs, otypes, mu_d, sg_d, mu_t, sg_t, ttypes), build)
        if io is None:
            return dict(kind=_L.POP_KINDS["ovc"], table=cells_t, objects=objs_t, object_types=types_t,
                        n_objects=int(len(objs)), walls_occlude=1 if occlude else 0,
                        egocentric=1 if self.reference_frame == "egocentric" else 0)
        env, _w = Env.device_tables(self._device)
        rc = _L.lib.riab_object_vector_cells(env, io, _L.ptr(objs_t), _L.ptr(types_t), int(len(objs)), _L.ptr(cells_t),
                                             n, 1 if occlude else 0, 1 if self.reference_frame == "egocentric" else 0,
                                             stream)
        _L.check(rc, "riab_object_vector_cells")


class FieldOfViewOVCs(ObjectVectorCells):
    """Egocentric object vector cells tiling the agent's field of view (reference
    Neurons.py:2119-2150)."""

    default_params = {
        "distance_range": [0.02, 0.4],
        "angle_range": [0, 75],
        "spatial_resolution": 0.02,
        "beta": 5,
        "cell_arrangement": "diverging_manifold",
        "object_tuning_type": None,
    }

    def __init__(self, Agent, params={}):
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        if self.params["object_tuning_type"] is None:
            warnings.warn("For FieldOfViewOVCs you must specify the object type they are selective for with the "
                          "'object_tuning_type' parameter ('random' or an integer). For now defaulting to "
                          "params['object_tuning_type'] = 0.")
            self.params["object_tuning_type"] = 0
        self.params["reference_frame"] = "egocentric"
        assert self.params["cell_arrangement"] is not None, "cell_arrangement must be set for FOV Neurons"
        self._warn_if_n_changes = "n" in params and params["n"] is not None
        super().__init__(Agent, self.params)


class AgentVectorCells(VectorCells):
    """Vector cells tuned to ANOTHER AGENT: ObjectVectorCells whose single object is `Other_Agent`'s
    position (reference Neurons.py:2151-2320).  Batched: lane b of `Agent` sees lane b of `Other_Agent`
    (equal `n_agents`), or every lane sees the one agent of a single-agent `Other_Agent`.  The other
    agent's position is read where it stands when `update()` / `get_state()` is called, as in the
    reference's interleaved `for Ag in Env.Agents: Ag.update(); ...` loops; `Agent.simulate()` and step
    plans advance one Agent object on its own and refuse these cells."""

    default_params = {
        "name": "AgentVectorCell",
        "walls_occlude": True,
    }

    def __init__(self, Agent, Other_Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        if not hasattr(self, "_warn_if_n_changes"):
            self._warn_if_n_changes = "n" in params and params["n"] is not None
        super().__init__(Agent, self.params)
        self.tuning_type_agent = Other_Agent
        self.wall_geometry = "line_of_sight" if self.walls_occlude else "euclidean"
        if Other_Agent is not None and Other_Agent._B not in (1, self._B):
            raise ValueError(f"Other_Agent has {Other_Agent._B} agents; expected 1 or {self._B} (one per lane)")

    def _other_rows(self, width):
        """float32 [2, width] device rows of the other agent's position, one per lane of this launch."""
        Other = self.tuning_type_agent
        Other._sync_plan()
        xy = Other._state[_L.S_POS_X:_L.S_POS_Y + 1].to(torch.float32)
        if Other._B == 1:
            return xy[:, :1].expand(2, width).contiguous()
        if xy.shape[1] != width:
            raise ValueError("AgentVectorCells away from the agents need a single-agent Other_Agent, or one position "
                             "per lane")
        return xy.contiguous()

    def get_state_tensor(self, evaluate_at="agent", **kwargs):
        if self.tuning_type_agent is None:
            self._last_P = self._B
            return torch.zeros((int(self.n), self._Bp), dtype=torch.float32, device=self._device)
        return super().get_state_tensor(evaluate_at, **kwargs)

    def _rates_from_trajectory(self, traj, out, t0, tc, step0, dt, stream):
        raise NotImplementedError("AgentVectorCells follow another Agent object step by step: advance both agents "
                                  "with update() (simulate() runs one Agent on its own)")

    def _call(self, io, stream):
        if io is None:
            raise NotImplementedError("AgentVectorCells cannot be recorded in a step plan (a plan steps one Agent "
                                      "object); advance them with update()")
        n = int(self.n)
        Env = self.Agent.Environment
        mu_d = np.asarray(self.tuning_distances, dtype=np.float64)
        sg_d = np.asarray(self.sigma_distances, dtype=np.float64)
        mu_t = np.asarray(self.tuning_angles, dtype=np.float64)
        sg_t = np.asarray(self.sigma_angles, dtype=np.float64)
        occlude = self.wall_geometry == "line_of_sight"
        if occlude:
            assert Env.boundary_conditions == "solid", \
                "line of sight geometry not available for periodic boundary conditions"

        def build():
            a = np.sqrt(LOG2E / 2) / sg_d
            cells = np.stack((a * mu_d, a, np.cos(mu_t), np.sin(mu_t), LOG2E / sg_t ** 2, np.zeros(n)), axis=-1)
            return torch.from_numpy(np.ascontiguousarray(cells, dtype=np.float32)).to(self._device)

        cells_t = self._tables((mu_d, sg_d, mu_t, sg_t), build)
        other = self._other_rows(int(io.B))
        self._keep_other = other
        env, _w = Env.device_tables(self._device)
        rc = _L.lib.riab_agent_vector_cells(env, io, _L.ptr(other[0]), _L.ptr(other[1]), 0, _L.ptr(cells_t), n,
                                            1 if occlude else 0, 1 if self.reference_frame == "egocentric" else 0,
                                            stream)
        _L.check(rc, "riab_agent_vector_cells")


class FieldOfViewAVCs(AgentVectorCells):
    """Egocentric agent vector cells tiling the agent's field of view (reference Neurons.py:2323-2355)."""

    default_params = {
        "distance_range": [0.02, 0.4],
        "angle_range": [0, 75],
        "spatial_resolution": 0.02,
        "beta": 5,
        "cell_arrangement": "diverging_manifold",
    }

    def __init__(self, Agent, Other_Agent, params={}):
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        self.params["reference_frame"] = "egocentric"
        assert self.params["cell_arrangement"] is not None, "cell_arrangement must be set for FOV Neurons"
        self._warn_if_n_changes = "n" in params and params["n"] is not None
        super().__init__(Agent, Other_Agent, self.params)


# ================================================================================================
class HeadDirectionCells(Neurons):
    """Head direction cells: von Mises tuning to the agent's head direction
    (reference Neurons.py:2357-2485)."""

    _stream_kind = "hdc"
    _watch_arrays = ("preferred_angles", "angular_tunings")
    _watch_scalars = ()
    default_params = {
        "min_fr": 0,
        "max_fr": 1,
        "n": 10,
        "angular_spread_degrees": 45,
        "name": "HeadDirectionCells",
    }

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        self.n = self.params["n"]
        self.preferred_angles = np.linspace(0, 2 * np.pi, self.n + 1)[:-1]
        self.angular_tunings = np.array([self.params["angular_spread_degrees"] * np.pi / 180] * self.n)
        super().__init__(Agent, self.params)

    def get_state_tensor(self, evaluate_at="agent", use_velocity=False, **kwargs):
        """`use_velocity=True` tunes to the direction of the agent's velocity (or the `velocity=`
        kwarg) instead of its head direction (reference Neurons.py:2421-2461)."""
        if not use_velocity:
            return super().get_state_tensor(evaluate_at, **kwargs)
        if evaluate_at == "agent":
            vel = np.asarray(self.Agent.velocity, dtype=np.float64).reshape(-1, 2)
            pos = np.asarray(self.Agent.pos, dtype=np.float64).reshape(-1, 2)
        else:
            vel = np.asarray(kwargs.get("velocity", [1.0, 0.0]), dtype=np.float64).reshape(-1, 2)
            pos = (self.Agent.Environment.flattened_discrete_coords if evaluate_at == "all"
                   else np.asarray(kwargs.get("pos", np.zeros((len(vel), 2))), dtype=np.float64).reshape(-1, 2))
        direction = vel / np.linalg.norm(vel, axis=-1, keepdims=True)
        return super().get_state_tensor(None, pos=pos, head_direction=direction)

    def get_state(self, evaluate_at="agent", use_velocity=False, **kwargs):
        t = self.get_state_tensor(evaluate_at, use_velocity=use_velocity, **kwargs)
        return t[:, :self._last_P].cpu().numpy().astype(np.float64)

    def _call(self, io, stream):
        n = int(self.n)
        pref = np.asarray(self.preferred_angles, dtype=np.float64)
        sig = np.asarray(self.angular_tunings, dtype=np.float64)

        def build():
            f32 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self._device)  # noqa: E731
            return f32(np.stack((np.cos(pref), np.sin(pref), LOG2E / sig ** 2), axis=-1))

        tab = self._tables((pref, sig), build)
        if io is None:
            return dict(kind=_L.POP_KINDS["hdc"], table=tab)
        rc = _L.lib.riab_head_direction_cells(io, _L.ptr(tab), n, stream)
        _L.check(rc, "riab_head_direction_cells")

    def _state_op(self, d):
        from . import ops  # noqa: F401  (registers torch.ops.riab.*)
        f = self._call(None, None)
        return torch.ops.riab.head_direction_cells(d[2:4], f["table"], float(self.min_fr), float(self.max_fr))


# ================================================================================================
class VelocityCells(HeadDirectionCells):
    """HeadDirectionCells tuned to the direction of the agent's velocity, scaled by
    `|velocity| / one_sigma_speed`, one_sigma_speed = speed_mean + speed_std when the cells are made
    (reference Neurons.py:2534-2583).  At the agent the reference reads `Agent.velocity` (the state of
    the motion model, not the measured velocity the history keeps), which lives in the float64 agent
    state only: `update()`, `get_state()` and step plans read it there; `Agent.simulate()` (whose rate
    stage otherwise runs on the float32 history records) advances such populations through a native step plan."""

    _stream_kind = None  # reads the float64 velocity state, not the history rows
    _state_op = None     # (its own kernel entry: riab_velocity_cells)
    default_params = {
        "min_fr": 0,
        "max_fr": 1,
        "name": "VelocityCells",
    }

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        self.one_sigma_speed = float(self.Agent.speed_mean + self.Agent.speed_std)
        self._vel_from_state = False
        super().__init__(Agent, self.params)

    def update(self, **kwargs):
        self._vel_from_state = True
        try:
            super().update(**kwargs)
        finally:
            self._vel_from_state = False

    def get_state_tensor(self, evaluate_at="agent", **kwargs):
        if evaluate_at == "agent":
            self._vel_from_state = True
            try:
                return Neurons.get_state_tensor(self, "agent")
            finally:
                self._vel_from_state = False
        # away from the agent the reference tunes to the given `velocity=` but still scales by the
        # AGENT's speed (Neurons.py:2581): the direction tuning runs as HeadDirectionCells, the scale
        # is one factor per agent
        vel = np.asarray(kwargs.get("velocity", [1.0, 0.0]), dtype=np.float64).reshape(-1, 2)
        pos = (self.Agent.Environment.flattened_discrete_coords if evaluate_at == "all"
               else np.asarray(kwargs.get("pos", np.zeros((len(vel), 2))), dtype=np.float64).reshape(-1, 2))
        direction = vel / np.linalg.norm(vel, axis=-1, keepdims=True)
        self._as_hdc = True
        try:
            out = Neurons.get_state_tensor(self, None, pos=pos, head_direction=direction)
        finally:
            self._as_hdc = False
        speed = torch.linalg.vector_norm(self.Agent._state[_L.S_VEL_X:_L.S_VEL_Y + 1, :self._B], dim=0)
        if self._B == 1:
            return out * (speed[0] / self.one_sigma_speed).to(torch.float32)
        if self._last_P != self._B:
            raise ValueError("VelocityCells.get_state away from the agents scales by each agent's own speed: "
                             "pass one velocity per agent")
        scale = torch.ones(out.shape[1], dtype=torch.float32, device=out.device)
        scale[:self._B] = (speed / self.one_sigma_speed).to(torch.float32)
        return out * scale

    def get_state(self, evaluate_at="agent", **kwargs):
        t = self.get_state_tensor(evaluate_at, **kwargs)
        return t[:, :self._last_P].cpu().numpy().astype(np.float64)

    _reads_agent_state = True  # Agent.simulate() runs such populations through a native step plan

    def _rates_from_trajectory(self, traj, out, t0, tc, step0, dt, stream):
        raise NotImplementedError("VelocityCells read Agent.velocity, which the history rows do not keep "
                                  "(Agent.simulate() advances them through a step plan instead)")

    def _call(self, io, stream):
        if getattr(self, "_as_hdc", False):
            return super()._call(io, stream)
        f = super()._call(None, None)
        if io is None:
            return dict(kind=_L.POP_KINDS["velocity"], table=f["table"], one_sigma_speed=float(self.one_sigma_speed))
        st = self.Agent._state
        vx, vy = (_L.ptr(st[_L.S_VEL_X]), _L.ptr(st[_L.S_VEL_Y])) if self._vel_from_state else (None, None)
        rc = _L.lib.riab_velocity_cells(io, _L.ptr(f["table"]), int(self.n), float(self.one_sigma_speed), vx, vy, stream)
        _L.check(rc, "riab_velocity_cells")


class SpeedCell(Neurons):
    """One cell whose rate is `|measured velocity| / one_sigma_speed` scaled to [min_fr, max_fr]
    (reference Neurons.py:2586-2651; `evaluate_at="agent"` reads `Agent.history["vel"][-1]`, the
    measured velocity of the newest step; otherwise pass `vel=`)."""

    default_params = {
        "min_fr": 0,
        "max_fr": 1,
        "name": "SpeedCell",
    }
    _H_DIR = (_L.H_VEL_X, _L.H_VEL_Y)
    _S_DIR = (_L.S_MVEL_X, _L.S_MVEL_Y)

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        if "n" in params and params["n"] != 1:
            warnings.warn(f"Ignoring 'n' parameter value ({params['n']}) that was passed for "
                          f"{self.params['name']}. Only 1 speed cell is needed.")
        self.params["n"] = 1
        self.one_sigma_speed = float(self.Agent.speed_mean + self.Agent.speed_std)
        super().__init__(Agent, self.params)
        self.n = 1

    def get_state_tensor(self, evaluate_at="agent", **kwargs):
        if evaluate_at == "agent":
            return super().get_state_tensor("agent")
        vel = np.asarray(kwargs["vel"], dtype=np.float64).reshape(-1, 2)
        return super().get_state_tensor(None, pos=np.zeros((len(vel), 2)), head_direction=vel)

    def _call(self, io, stream):
        if io is None:
            return dict(kind=_L.POP_KINDS["speed"], one_sigma_speed=float(self.one_sigma_speed))
        _L.check(_L.lib.riab_speed_cell(io, float(self.one_sigma_speed), stream), "riab_speed_cell")


# ================================================================================================
class RandomSpatialNeurons(Neurons):
    """Neurons with smooth random spatial tuning: per neuron a function sampled from a Gaussian-process
    prior (squared-exponential kernel of the environment distance, `lengthscale`) on a grid of anchor
    points, squashed by a sigmoid into [min_fr, max_fr]; anywhere else the rate is the kernel-weighted
    local average of the anchors' targets (reference Neurons.py:2865-2960).  The sampling is host NumPy
    at construction (same `np.random` call as the reference); the local average runs on device."""

    default_params = {
        "lengthscale": 0.1,
        "max_fr": 1,
        "min_fr": 0,
        "n": 10,
        "wall_geometry": "geodesic",
        "name": "RandomSpatialNeurons",
    }

    def __init__(self, Agent, params={}):
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        super().__init__(Agent, self.params)
        Env = self.Agent.Environment
        if self.wall_geometry == "geodesic" and len(Env.walls) > 5:
            print("Geodesic wall geometry only possible in environments with one or no additional walls. Using "
                  "'line_of_sight' instead. If this is slow, consider trying 'euclidean'")
            self.wall_geometry = "line_of_sight"
        assert self.lengthscale >= 0.02, "lengthscale must be greater than 0.02 m"
        # Neurons.py:2899-2912: anchors at least as dense as the lengthscale, one draw of n functions
        self.X = Env.discretise_environment(dx=min(0.05, self.lengthscale))
        self.X = self.X.reshape(-1, self.X.shape[-1])
        self.Q = self.kernel(self.X, self.X)
        warnings.filterwarnings("ignore", category=RuntimeWarning)
        self.targets = np.random.multivariate_normal(mean=np.zeros(self.Q.shape[0]), cov=self.Q, size=self.n).T
        warnings.filterwarnings("default", category=RuntimeWarning)
        # utils.activate(..., "sigmoid", mid_x=0, width_x=2) (utils.py:962-975)
        beta = np.log((1 - 0.05) / 0.05) / (0.5 * 2)
        self.targets = (self.max_fr - self.min_fr) / (1 + np.exp(-beta * self.targets)) + self.min_fr

    def kernel(self, x1, x2):
        """Squared-exponential covariance `(len(x1), len(x2))` of the environment distance (Neurons.py:2944-2956)."""
        d = self.Agent.Environment.get_distances_between___accounting_for_environment(
            x1, x2, wall_geometry=self.wall_geometry)
        return np.exp(-(d ** 2) / (2 * self.lengthscale ** 2))

    def _call(self, io, stream):
        X = np.asarray(self.X, dtype=np.float64).reshape(-1, 2)
        targets = np.asarray(self.targets, dtype=np.float64)
        ell = float(self.lengthscale)

        def build():
            tab = np.empty((len(X), 3), dtype=np.float64)
            tab[:, 0], tab[:, 1] = X[:, 0], X[:, 1]
            tab[:, 2] = -LOG2E / (2 * ell ** 2)
            f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self._device)  # noqa: E731
            return f32(tab), f32(targets)

        anchors, tg = self._tables((X, targets, ell), build)
        geom = self.wall_geometry
        if geom == "geodesic" and len(self.Agent.Environment.walls) <= 4:
            geom = "euclidean"  # Environment.py:741-742
        if io is None:
            return dict(kind=_L.POP_KINDS["random_spatial"], table=anchors, targets=tg, n_anchors=len(X),
                        geometry=_L.GEOMETRIES[geom])
        env, _w = self.Agent.Environment.device_tables(self._device)
        rc = _L.lib.riab_random_spatial_neurons(env, io, _L.ptr(anchors), len(X), _L.ptr(tg), int(self.n),
                                                _L.GEOMETRIES[geom], stream)
        _L.check(rc, "riab_random_spatial_neurons")


# ================================================================================================
class FeedForwardLayer(Neurons):
    """A layer whose firing rates are an activated linear combination of the rates of its input
    layers (reference Neurons.py:2654-2860): `firingrate = act(sum_l W_l @ I_l + biases)`.

    `inputs[name]` keeps the reference's dictionary (`"layer"`, `"w"` (n, n_in) NumPy — edit it
    freely, e.g. in a learning rule; the device copy is refreshed when it changes —, `"w_init"`,
    `"n"`, `"recurrent"`; `"I"` is NOT refreshed per step as the reference does (Neurons.py:2825): the inputs
    stay on the device, read `inputs[name]["layer"].firingrate` instead).  The contraction runs on the fp32 matrix cores
    (`riab_feedforward`, csrc/riab_ff.hip) directly on the input layers' device-resident rates;
    `firingrate_prime` holds the activation derivative like the reference.  Activation functions:
    the reference's named ones (linear, sigmoid, relu, tanh, retanh, softmax); a Python callable
    cannot run inside the kernel and raises NotImplementedError."""

    default_params = {
        "n": 10,
        "input_layers": [],
        "activation_function": {"activation": "linear"},
        "name": "FeedForwardLayer",
        "biases": None,
    }

    def __init__(self, Agent, params={}):
        self.Agent = Agent
        self.params = copy.deepcopy(__class__.default_params)
        self.params.update(params)
        if "activation_params" in self.params:
            warnings.warn("The parameter 'activation_params' is deprecated. Use 'activation_function' instead.")
            self.params["activation_function"] = self.params["activation_params"]
        super().__init__(Agent, self.params)
        assert isinstance(self.input_layers, list), "param['input_layers'] must be a list."
        if len(self.input_layers) == 0:
            warnings.warn("No input layers have been provided. Either hand them in in the params dictionary "
                          "params['input_layers']=[list,of,inputs] or use self.add_input_layer() to add them manually.")
        if not isinstance(self.activation_function, dict):
            raise NotImplementedError("a Python callable activation cannot run inside the HIP kernel; use one of "
                                      "the named activations {'activation': 'linear'|'sigmoid'|'relu'|'tanh'|"
                                      "'retanh'|'softmax', ...}")
        self.inputs = {}
        for layer in self.input_layers:
            self.add_input(layer)
        if self.biases is None:
            self.biases = np.zeros(self.n)
        self._rates_prime = torch.zeros((int(self.n), self._Bp), dtype=torch.float32, device=self._device)

    def add_input(self, input_layer, w=None, w_init_scale=1, recurrent=False, **kwargs):
        """Add an input layer with weights `w` (n, n_in); None draws N(0, w_init_scale/sqrt(n_in))."""
        n_in, name = input_layer.n, input_layer.name
        if w is None:
            w = np.random.normal(loc=0, scale=w_init_scale / np.sqrt(n_in), size=(self.n, n_in))
        self.inputs[name] = {"layer": input_layer, "w": w, "w_init": w.copy(), "I": np.zeros(n_in), "n": n_in,
                             "recurrent": recurrent}
        self.inputs[name].update(kwargs)

    add_input_layer = add_input

    @property
    def firingrate_prime(self):
        a = self._rates_prime[:, :self._B].cpu().numpy().astype(np.float64)
        return a[:, 0] if self._B == 1 else a

    # ---- activation parameters -> the kernel's 4 floats (utils.activate, utils.py:919-1026) -----
    def _activation(self):
        spec = dict(self.activation_function)
        name = spec.get("activation", "sigmoid")
        if "function" in spec:
            raise NotImplementedError("bespoke Python activation functions cannot run inside the HIP kernel")
        assert name in _L.ACTIVATIONS, f"unknown activation {name}"
        if name == "sigmoid":
            d = {"max_fr": 1, "min_fr": 0, "mid_x": 1, "width_x": 2}
            d.update(spec)
            beta = np.log((1 - 0.05) / 0.05) / (0.5 * d["width_x"])
            p = (d["max_fr"], d["min_fr"], d["mid_x"], beta)
        elif name == "linear":
            p = (0, 0, 0, 0)
        else:
            d = {"gain": 1, "threshold": 0}
            d.update(spec)
            p = (d["gain"], d["threshold"], 0, 0)
        return _L.ACTIVATIONS[name], (_L.C.c_float * 4)(*[float(x) for x in p])

    def _auto_key(self):
        acts = self._activation()
        return (tuple(id(self._device_weights(e)) for e in self.inputs.values()),
                np.asarray(self.biases, dtype=np.float32).tobytes(), acts[0], tuple(float(x) for x in acts[1]),
                float(self.min_fr), float(self.max_fr), self.noise_std, self.noise_coherence_time,
                bool(self.save_history), bool(self.save_spikes))

    def _device_weights(self, entry):
        """W^T padded to a multiple of 32 outputs, float32 on device; refreshed when `w` changes."""
        w = np.ascontiguousarray(np.asarray(entry["w"], dtype=np.float64))
        key = w.tobytes()
        hit = entry.get("_dev")
        if hit is not None and hit[0] == key:
            return hit[1]
        n, n_in = w.shape
        Mp = (n + 31) // 32 * 32
        wt = np.zeros((n_in, Mp), dtype=np.float32)
        wt[:, :n] = w.T
        t = torch.from_numpy(wt).to(self._device)
        entry["_dev"] = (key, t)
        return t

    def _gemm(self, input_tensors, T, B, out, out_prime, stream):
        """out[T][n][B] = act(sum_l W_l @ input_tensors[l] + biases); input tensors [T][n_in][B]."""
        entries = list(self.inputs.values())
        arr = (_L.RiabFFInput * len(entries))()
        keep = []
        for i, (e, x) in enumerate(zip(entries, input_tensors)):
            wt = self._device_weights(e)
            arr[i].rates, arr[i].wt, arr[i].n_in = x.data_ptr(), wt.data_ptr(), int(e["n"])
            keep.append((wt, x))
        bias = np.asarray(self.biases, dtype=np.float32).reshape(-1)
        bias_t = self._tables((bias,), lambda: torch.from_numpy(bias.copy()).to(self._device))
        act, pars = self._activation()
        rc = _L.lib.riab_feedforward(arr, len(entries), _L.ptr(bias_t), int(self.n), int(T), int(B), act, pars,
                                     _L.ptr(out), _L.ptr(out_prime), stream)
        _L.check(rc, "riab_feedforward")
        self._keep = (keep, bias_t, out, out_prime)

    # ---- Neurons plumbing ------------------------------------------------------------------------
    def _launch(self, px, py, hx, hy, pos_ld, T, B, rates, spikes, u_in, dt, step0, from_f64=False, stream=None):
        """update() path: inputs are the input layers' LAST firing rates (evaluate_at='last')."""
        stream = _L.current_stream() if stream is None else stream
        xs = []
        for e in self.inputs.values():
            x = e["layer"]._rates
            xs.append(x.reshape(1, *x.shape))
        self._gemm(xs, 1, self._Bp, rates, self._rates_prime, stream)
        if spikes is not None:
            io = self._io(None, None, None, None, self._Bp, 1, self._Bp, rates, spikes, u_in, dt, step0)
            _L.check(_L.lib.riab_spikes(io, int(self.n), stream), "riab_spikes")

    def get_state_tensor(self, evaluate_at="last", max_recurrence=None, **kwargs):
        if evaluate_at in ("last", "agent"):
            out = torch.empty((1, int(self.n), self._Bp), dtype=torch.float32, device=self._device)
            xs = [e["layer"]._rates.reshape(1, *e["layer"]._rates.shape) for e in self.inputs.values()]
            self._last_P = self._B
            self._gemm(xs, 1, self._Bp, out, self._rates_prime, _L.current_stream())
            return out[0]
        xs, P = [], None
        for e in self.inputs.values():
            kw = dict(kwargs)
            if isinstance(e["layer"], FeedForwardLayer):
                nxt = max_recurrence
                if max_recurrence is not None and e["recurrent"]:
                    if max_recurrence <= 0:
                        continue
                    nxt = max_recurrence - 1
                kw["max_recurrence"] = nxt
            x = e["layer"].get_state_tensor(evaluate_at, **kw)
            P = e["layer"]._last_P
            xs.append(x.reshape(1, *x.shape))
        Pp = xs[0].shape[-1]
        self._last_P = P
        out = torch.empty((1, int(self.n), Pp), dtype=torch.float32, device=self._device)
        self._gemm(xs, 1, Pp, out, None, _L.current_stream())
        return out[0]

    def get_state(self, evaluate_at="last", max_recurrence=None, **kwargs):
        t = self.get_state_tensor(evaluate_at, max_recurrence=max_recurrence, **kwargs)
        return t[:, :self._last_P].cpu().numpy().astype(np.float64)

    def _rates_from_trajectory(self, traj, out, t0, tc, step0, dt, stream):
        """Fused path: the input layers' rows of the same chunk feed the GEMM in place; then, like every
        population (Neurons._rates_from_trajectory), the OU noise pass and the spikes on the final rates."""
        outs = self.Agent._sim_outs
        order = list(outs)
        xs = []
        for e in self.inputs.values():
            if e["recurrent"] or e["layer"] is self:
                raise NotImplementedError("recurrent inputs need one step per launch: use update(), not simulate()")
            if e["layer"] not in outs or order.index(e["layer"]) >= order.index(self):
                raise ValueError(f"simulate(): input layer {e['layer'].name} must be in `neurons` BEFORE {self.name} "
                                 "(its rows of the chunk are read in place)")
            xs.append(e["layer"]._chunk_view(outs[e["layer"]], t0, tc)[0])
        fr, sp = self._chunk_view(out, t0, tc)
        self._gemm(xs, tc, self._Bp, fr, None, stream)
        if self.noise_std != 0:
            tau = float(self.noise_coherence_time)
            sigma = float(np.sqrt((2 * float(self.noise_std) ** 2) / (tau * dt)))
            Ag = self.Agent
            rc = _L.lib.riab_neuron_noise(_L.ptr(self._noise), _L.ptr(fr), None, int(self.n), self._Bp, int(tc),
                                          float(dt / tau), float(sigma * dt), int(Ag.rng_seed), int(step0 + 1),
                                          int(self.pop_id), int(Ag.agent_id0), stream)
            _L.check(rc, "riab_neuron_noise")
        if sp is not None:
            io = self._io(None, None, None, None, self._Bp, tc, self._Bp, fr, sp, None, dt, step0 + 1)
            _L.check(_L.lib.riab_spikes(io, int(self.n), stream), "riab_spikes")


# Populations whose device tables derive from float64 array attributes alone (Agent._fast_record): exact types
FAST_REPEAT_TYPES = (PlaceCells, GridCells, HeadDirectionCells)

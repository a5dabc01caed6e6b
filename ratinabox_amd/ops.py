"""`torch.ops.riab.*` — the C ABI of libriab_hip.so (include/riab_hip.h) registered as PyTorch custom operators.

The reference's only PyTorch touch-point is a user network consuming firing rates
(ratinabox/contribs/NeuralNetworkNeurons.py:6-7); registering the kernels as operators is what lets such code —
and `torch.compile` / CUDA-graph capture around it — call the accelerated path directly on device tensors:

    rates = torch.ops.riab.place_cells(pos, table, None, [0, 1, 0, 1, 1.0], False, 0, 0, 0.2, 0.0, 1.0)   # (n, P)

Every operator is a thin wrapper: it checks shapes, fills the ABI's structs with `data_ptr()`s and enqueues the
kernel on the CURRENT stream; nothing is synchronised.  Each has a fake (meta) implementation, so the operators
trace under `torch.compile(fullgraph=True)` and FakeTensorMode.  Layouts are the ABI's: the position / agent axis
is the last (fastest) one and must be a multiple of 4 (pad; padding columns are computed like any other).

Functional operators (return a new tensor)
    place_cells, grid_cells, head_direction_cells, boundary_vector_cells   rates float32 (n, P)
    spikes                                                              uint8 (T, n, B) from rates (T, n, B)
    feedforward                                                         float32 (T, n_out, B)
In-place operators
    agent_step_      T fused Agent.update() steps on the float64 state (12, B); writes the history rows
    simulate_        T x (Agent.update(); N.update() for N in populations) as ONE native call (riab_simulate): the state,
                     the trajectory rows and every population's rate / spike rows are written

`Neurons.get_state()`, `Agent.update()` and `Agent.simulate()` go through these operators (step plans call the ABI
directly: their launches are issued from C++)."""
from typing import List, Optional

import torch
from torch import Tensor

from . import _lib as _L

C = _L.C

# Schemas are DEFINED once (Library "DEF") and the implementations registered for the CUDA (= HIP) key directly:
# `torch.library.custom_op` wraps every call in ~15 us of Python (measured on the per-step path), this route costs
# the dispatcher's boxed call into the Python function and nothing else.
_ns = torch.library.Library("riab", "DEF")
_TAGS = (torch.Tag.pt2_compliant_tag,)


def _register(schema, fake):
    """decorator: define `riab::<schema>`, register the function for CUDA and `fake` as its meta implementation"""
    name = schema.split("(", 1)[0]

    def deco(fn):
        _ns.define(schema, tags=_TAGS)
        _ns.impl(name, fn, "CUDA")
        torch.library.register_fake("riab::" + name, fake, lib=_ns)
        return fn
    return deco


def _env_struct(walls: Optional[Tensor], env: List[float], periodic: bool):
    """RiabEnv from operator arguments: env = [left, right, bottom, top, scale] for a box, or
    [left, right, bottom, top, scale, polygon, n_boundary, hole_mask_low32, hole_mask_high32] (include/riab_hip.h:
    RiabEnv) when the boundary is a general polygon and / or the environment has holes.  (The 64-bit edge mask travels
    as two 32-bit halves: a float list holds each of them exactly, all 64 walls of the C ABI included.)"""
    if len(env) not in (5, 9):
        raise ValueError("env must be [left, right, bottom, top, scale] (+ [polygon, n_boundary, hole_mask_lo, hole_mask_hi])")
    e = _L.RiabEnv()
    for i in range(4):
        e.extent[i] = float(env[i])
    e.scale = float(env[4])
    e.periodic = 1 if periodic else 0
    if len(env) == 9:
        e.polygon, e.n_boundary, e.hole_mask = int(env[5]), int(env[6]), int(env[7]) | (int(env[8]) << 32)
    if walls is None:
        e.n_walls, e.walls = 0, None
    else:
        if walls.dtype != torch.float64 or walls.dim() != 2 or walls.shape[1] != 4 or not walls.is_contiguous():
            raise ValueError("walls must be a contiguous float64 tensor (n_walls, 4) = (ax, ay, bx, by)")
        e.n_walls, e.walls = int(walls.shape[0]), walls.data_ptr()
    return e


def _rows(t: Tensor, rows: int, what: str):
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[0] != rows or not t.is_contiguous() or t.shape[1] % 4:
        raise ValueError(f"{what} must be a contiguous float32 tensor ({rows}, P) with P a multiple of 4, got "
                         f"{tuple(t.shape)} {t.dtype}")
    return int(t.shape[1])


def _io(pos: Optional[Tensor], hd: Optional[Tensor], P: int, out: Tensor, min_fr: float, max_fr: float):
    io = _L.RiabRateIO()
    if pos is not None:
        io.pos_x, io.pos_y = pos[0].data_ptr(), pos[1].data_ptr()
    if hd is not None:
        io.hd_x, io.hd_y = hd[0].data_ptr(), hd[1].data_ptr()
    io.pos_ld, io.T, io.B = P, 1, P
    io.rates = out.data_ptr()
    io.dt = 0.0
    io.min_fr, io.max_fr = float(min_fr), float(max_fr)
    return io


# ---- PlaceCells ------------------------------------------------------------------------------------------------
@_register("place_cells(Tensor pos, Tensor table, Tensor? walls, float[] env, bool periodic, int description, "
           "int geometry, float top_hat_width, float min_fr, float max_fr) -> Tensor",
           lambda pos, table, walls, env, periodic, description, geometry, top_hat_width, min_fr, max_fr:
           pos.new_empty((table.shape[0], pos.shape[1])))
def place_cells(pos: Tensor, table: Tensor, walls: Optional[Tensor], env: List[float], periodic: bool, description: int,
                geometry: int, top_hat_width: float, min_fr: float, max_fr: float) -> Tensor:
    """PlaceCells.get_state (riab_place_cells).  pos float32 (2, P); table float32 (n, 3) = (centre x, centre y,
    -log2(e)/(2 w^2)); walls float64 (n_walls, 4) in Environment.walls order or None; env = [l, r, b, t, scale];
    description RIAB_PC_*; geometry RIAB_GEOM_*.  Returns rates float32 (n, P)."""
    P = _rows(pos, 2, "pos")
    n = int(table.shape[0])
    out = torch.empty((n, P), dtype=torch.float32, device=pos.device)
    e = _env_struct(walls, env, periodic)
    io = _io(pos, None, P, out, min_fr, max_fr)
    _L.check(_L.lib.riab_place_cells(e, io, _L.ptr(table), n, int(description), int(geometry), float(top_hat_width),
                                     _L.current_stream()), "riab_place_cells")
    return out


# ---- GridCells -------------------------------------------------------------------------------------------------
@_register("grid_cells(Tensor pos, Tensor table, int description, float f0, float min_fr, float max_fr) -> Tensor",
           lambda pos, table, description, f0, min_fr, max_fr: pos.new_empty((table.shape[0], pos.shape[1])))
def grid_cells(pos: Tensor, table: Tensor, description: int, f0: float, min_fr: float, max_fr: float) -> Tensor:
    """GridCells.get_state (riab_grid_cells).  table float32 (n, 9), see include/riab_hip.h."""
    P = _rows(pos, 2, "pos")
    n = int(table.shape[0])
    out = torch.empty((n, P), dtype=torch.float32, device=pos.device)
    io = _io(pos, None, P, out, min_fr, max_fr)
    _L.check(_L.lib.riab_grid_cells(io, _L.ptr(table), n, int(description), float(f0), _L.current_stream()),
             "riab_grid_cells")
    return out


# ---- HeadDirectionCells ----------------------------------------------------------------------------------------
@_register("head_direction_cells(Tensor head_direction, Tensor table, float min_fr, float max_fr) -> Tensor",
           lambda head_direction, table, min_fr, max_fr: head_direction.new_empty((table.shape[0], head_direction.shape[1])))
def head_direction_cells(head_direction: Tensor, table: Tensor, min_fr: float, max_fr: float) -> Tensor:
    """HeadDirectionCells.get_state (riab_head_direction_cells).  head_direction float32 (2, P); table float32 (n, 3)
    = (cos, sin of the preferred angle, log2(e)/sigma^2)."""
    P = _rows(head_direction, 2, "head_direction")
    n = int(table.shape[0])
    out = torch.empty((n, P), dtype=torch.float32, device=head_direction.device)
    io = _io(None, head_direction, P, out, min_fr, max_fr)
    _L.check(_L.lib.riab_head_direction_cells(io, _L.ptr(table), n, _L.current_stream()), "riab_head_direction_cells")
    return out


# ---- BoundaryVectorCells ---------------------------------------------------------------------------------------
@_register("boundary_vector_cells(Tensor pos, Tensor? head_direction, Tensor walls, float[] env, bool periodic, "
           "Tensor test_dirs, Tensor ray_rden, Tensor cells, Tensor vm_table, Tensor inv_norm, bool egocentric, "
           "Tensor? cell_rows, Tensor? windows, float min_fr, float max_fr) -> Tensor",
           lambda pos, head_direction, walls, env, periodic, test_dirs, ray_rden, cells, vm_table, inv_norm, egocentric,
           cell_rows, windows, min_fr, max_fr: pos.new_empty((cells.shape[1], pos.shape[1])))
def boundary_vector_cells(pos: Tensor, head_direction: Optional[Tensor], walls: Tensor, env: List[float], periodic: bool,
                          test_dirs: Tensor, ray_rden: Tensor, cells: Tensor, vm_table: Tensor, inv_norm: Tensor,
                          egocentric: bool, cell_rows: Optional[Tensor], windows: Optional[Tensor], min_fr: float,
                          max_fr: float) -> Tensor:
    """BoundaryVectorCells.get_state (riab_boundary_vector_cells_windowed).  Tables as documented in
    include/riab_hip.h (test_dirs float64 (K, 2), ray_rden float64 (K, n_walls), cells float32 (4, n), vm_table,
    inv_norm float32 (n,), optional direction windows)."""
    P = _rows(pos, 2, "pos")
    if egocentric:
        if head_direction is None or _rows(head_direction, 2, "head_direction") != P:
            raise ValueError("egocentric cells need head_direction (2, P)")
    n, K = int(cells.shape[1]), int(test_dirs.shape[0])
    out = torch.empty((n, P), dtype=torch.float32, device=pos.device)
    e = _env_struct(walls, env, periodic)
    io = _io(pos, head_direction if egocentric else None, P, out, min_fr, max_fr)
    _L.check(_L.lib.riab_boundary_vector_cells_windowed(e, io, _L.ptr(test_dirs), _L.ptr(ray_rden), K, _L.ptr(cells),
                                                        _L.ptr(vm_table), _L.ptr(inv_norm), n, 1 if egocentric else 0, None,
                                                        _L.ptr(cell_rows), _L.ptr(windows), _L.current_stream()),
             "riab_boundary_vector_cells")
    return out


# ---- Poisson spikes --------------------------------------------------------------------------------------------
@_register("spikes(Tensor rates, Tensor? uniforms, float dt, int seed, int step0, int pop_id, int agent_id0) -> Tensor",
           lambda rates, uniforms, dt, seed, step0, pop_id, agent_id0: rates.new_empty(rates.shape, dtype=torch.uint8))
def spikes(rates: Tensor, uniforms: Optional[Tensor], dt: float, seed: int, step0: int, pop_id: int,
           agent_id0: int) -> Tensor:
    """Neurons.save_to_history's spike rule `u < dt * rate` (riab_spikes) on rates float32 (T, n, B): with explicit
    `uniforms` (same shape) or, when None, the Philox uniforms keyed by (seed; step0 + t, cell, agent id, pop_id).
    Returns uint8 (T, n, B)."""
    if rates.dtype != torch.float32 or rates.dim() != 3 or not rates.is_contiguous() or rates.shape[2] % 4:
        raise ValueError("rates must be a contiguous float32 tensor (T, n, B) with B a multiple of 4")
    T, n, B = (int(x) for x in rates.shape)
    if uniforms is not None and (uniforms.shape != rates.shape or uniforms.dtype != torch.float32 or
                                 not uniforms.is_contiguous()):
        raise ValueError("uniforms must match rates (contiguous float32)")
    out = torch.empty((T, n, B), dtype=torch.uint8, device=rates.device)
    io = _L.RiabRateIO()
    io.pos_ld, io.T, io.B = B, T, B
    io.rates, io.spikes = rates.data_ptr(), out.data_ptr()
    io.u_in = uniforms.data_ptr() if uniforms is not None else None
    io.dt = float(dt)
    io.seed, io.step0, io.agent_id0, io.pop_id = int(seed) & 0xFFFFFFFFFFFFFFFF, int(step0), int(agent_id0), int(pop_id)
    _L.check(_L.lib.riab_spikes(io, n, _L.current_stream()), "riab_spikes")
    return out


# ---- FeedForwardLayer ------------------------------------------------------------------------------------------
@_register("feedforward(Tensor[] inputs, Tensor[] weights_t, Tensor bias, int activation, float[] act_params) -> Tensor",
           lambda inputs, weights_t, bias, activation, act_params:
           inputs[0].new_empty((inputs[0].shape[0], bias.shape[0], inputs[0].shape[2])))
def feedforward(inputs: List[Tensor], weights_t: List[Tensor], bias: Tensor, activation: int,
                act_params: List[float]) -> Tensor:
    """FeedForwardLayer.get_state (riab_feedforward, fp32 matrix cores): out[t][m][b] = act(sum_l sum_k
    w_l[m][k] * inputs_l[t][k][b] + bias[m]).  inputs_l float32 (T, n_in_l, B); weights_t_l float32 (n_in_l, Mp) =
    the weight matrix TRANSPOSED, rows zero-padded to Mp = n_out rounded up to 32; bias float32 (n_out,);
    activation RIAB_ACT_*; act_params as in include/riab_hip.h.  Returns float32 (T, n_out, B)."""
    if not inputs or len(inputs) != len(weights_t) or len(inputs) > 8:
        raise ValueError("1..8 input layers, one transposed weight matrix each")
    T, _, B = (int(x) for x in inputs[0].shape)
    n_out = int(bias.shape[0])
    arr = (_L.RiabFFInput * len(inputs))()
    for l, (x, w) in enumerate(zip(inputs, weights_t)):
        if x.dtype != torch.float32 or x.dim() != 3 or not x.is_contiguous() or x.shape[0] != T or x.shape[2] != B:
            raise ValueError("inputs must be contiguous float32 (T, n_in, B) with equal T and B")
        if w.dtype != torch.float32 or not w.is_contiguous() or w.shape[0] != x.shape[1] or w.shape[1] % 32 or \
                w.shape[1] < n_out:
            raise ValueError("weights_t[l] must be contiguous float32 (n_in_l, Mp), Mp = n_out rounded up to 32")
        arr[l].rates, arr[l].wt, arr[l].n_in = x.data_ptr(), w.data_ptr(), int(x.shape[1])
    pars = (C.c_float * 4)(*[float(p) for p in (list(act_params) + [0.0] * 4)[:4]])
    out = torch.empty((T, n_out, B), dtype=torch.float32, device=inputs[0].device)
    _L.check(_L.lib.riab_feedforward(arr, len(inputs), _L.ptr(bias), n_out, T, B, int(activation), pars, _L.ptr(out), None,
                                     _L.current_stream()), "riab_feedforward")
    return out


# ---- Agent.update ----------------------------------------------------------------------------------------------
MOTION_FIELDS = ("dt", "rot_theta_kw", "rot_sigma_kw", "rot_drift_kw", "speed_theta_kw", "speed_sigma_kw", "speed_mean_kw",
                 "speed_mean", "speed_std_is_zero", "has_drift", "drift_theta", "wall_repel_strength_kw",
                 "wall_repel_distance_kw", "thigmotaxis_kw", "hd_tau")


def seed_arg(seed: int) -> int:
    """A uint64 Philox key as the int64 the operator schemas carry (two's complement; the operators undo it)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed - (1 << 64) if seed >= (1 << 63) else seed


def motion_list(m) -> List[float]:
    """RiabMotion struct -> the flat list `agent_step_` takes (MOTION_FIELDS order)."""
    return [float(getattr(m, k)) for k in MOTION_FIELDS]


@_register("agent_step_(Tensor(a!) state, Tensor(b!)? hist, Tensor(c!)? diag, Tensor? walls, float[] env, bool periodic, "
           "float[] motion, Tensor? drift, Tensor? noise, Tensor(d!)? noise_out, Tensor? forced_pos, Tensor? resample_pos, "
           "int seed, int step0, int agent_id0, int T) -> ()",
           lambda state, hist, diag, walls, env, periodic, motion, drift, noise, noise_out, forced_pos, resample_pos, seed,
           step0, agent_id0, T: None)
def agent_step_(state: Tensor, hist: Optional[Tensor], diag: Optional[Tensor], walls: Optional[Tensor], env: List[float],
                periodic: bool, motion: List[float], drift: Optional[Tensor], noise: Optional[Tensor],
                noise_out: Optional[Tensor], forced_pos: Optional[Tensor], resample_pos: Optional[Tensor], seed: int,
                step0: int, agent_id0: int, T: int) -> None:
    """T fused Agent.update() steps (riab_agent_step), in place: state float64 (12, B) (rows RIAB_S_*), hist float32
    (T, 8, B) or None, diag int32 (4,) or None.  motion = the RiabMotion fields in MOTION_FIELDS order; drift float64
    (2, B); noise float64 (T, 2, B) explicit standard normals (None: Philox keyed by (seed; step0 + t, agent_id0 + b));
    noise_out records the normals used; forced_pos float64 (T, 2, B) moves the agents instead of the motion model;
    resample_pos float64 (T, 2, B): explicit replacement positions for agents that end a step in a hole / outside a
    polygonal boundary (None: Philox draws)."""
    if state.dtype != torch.float64 or state.dim() != 2 or state.shape[0] != _L.STATE_ROWS or not state.is_contiguous():
        raise ValueError("state must be a contiguous float64 tensor (12, B)")
    B = int(state.shape[1])
    if len(motion) != len(MOTION_FIELDS):
        raise ValueError(f"motion must hold the {len(MOTION_FIELDS)} RiabMotion fields")
    m = _L.RiabMotion()
    for k, v in zip(MOTION_FIELDS, motion):
        setattr(m, k, int(v) if k in ("speed_std_is_zero", "has_drift") else float(v))
    if hist is not None and (hist.dtype != torch.float32 or tuple(hist.shape) != (T, _L.HIST_ROWS, B) or
                             not hist.is_contiguous()):
        raise ValueError("hist must be a contiguous float32 tensor (T, 8, B)")
    for name, t in (("noise", noise), ("noise_out", noise_out), ("forced_pos", forced_pos), ("resample_pos", resample_pos)):
        if t is not None and (t.dtype != torch.float64 or tuple(t.shape) != (T, 2, B) or not t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous float64 tensor (T, 2, B)")
    if drift is not None and (drift.dtype != torch.float64 or tuple(drift.shape) != (2, B) or not drift.is_contiguous()):
        raise ValueError("drift must be a contiguous float64 tensor (2, B)")
    e = _env_struct(walls, env, periodic)
    _L.check(_L.lib.riab_agent_step(e, m, _L.ptr(state), B, int(agent_id0), _L.ptr(drift), _L.ptr(noise), _L.ptr(noise_out),
                                    _L.ptr(forced_pos), _L.ptr(resample_pos), int(seed) & 0xFFFFFFFFFFFFFFFF, int(step0),
                                    int(T), _L.ptr(hist), _L.ptr(diag), _L.current_stream()),
             "riab_agent_step")


# ---- the open-loop run: riab_simulate ----------------------------------------------------------------------------
@_register("simulate_(Tensor(a!) state, Tensor(b!) hist, Tensor(c!)[] rates, Tensor(d!)[] spikes, Tensor(e!) ctrl, "
           "Tensor(f!)? diag, int streamer, int run, int hist_row, int[] rate_rows, int[] spike_rows) -> ()",
           lambda state, hist, rates, spikes, ctrl, diag, streamer, run, hist_row, rate_rows, spike_rows: None)
def simulate_(state: Tensor, hist: Tensor, rates: List[Tensor], spikes: List[Tensor], ctrl: Tensor, diag: Optional[Tensor],
              streamer: int, run: int, hist_row: int, rate_rows: List[int], spike_rows: List[int]) -> None:
    """One riab_simulate call (include/riab_hip.h).  `run` is the address of a RiabSimulate argument block — built by
    `Agent.simulate_args()` or by hand through `ratinabox_amd._lib.RiabSimulate`; the caller keeps it, its population
    array and the tables they point to alive — and `streamer` the RiabStreamer handle.  What the call WRITES is taken
    from the tensors given here, not from the block (a tracer may hand the operator other tensors than the ones the
    block was built from: functionalisation runs it on copies): state float64 (12, B); hist float32 (rows, 8, B), the
    T rows from `hist_row` on; rates[i] float32 (rows, n_i, B) from `rate_rows[i]` on, one tensor per population of the
    block; spikes uint8 likewise for the populations whose block entry has a spike buffer, in order; ctrl int32
    control words; diag int32 (4,) or None.  Enqueues on the current stream, synchronises nothing.  Raises `RiabError`
    on a non-zero return code of the ABI (its `.code` holds it: RIAB_EUNSUPPORTED = nothing was launched)."""
    r = C.cast(C.c_void_p(run), C.POINTER(_L.RiabSimulate)).contents
    B, T, npop = int(r.B), int(r.T), int(r.n_pops)
    if state.dtype != torch.float64 or tuple(state.shape) != (_L.STATE_ROWS, B) or not state.is_contiguous():
        raise ValueError("state must be a contiguous float64 tensor (12, B) with the block's B")
    if hist.dtype != torch.float32 or hist.dim() != 3 or tuple(hist.shape[1:]) != (_L.HIST_ROWS, B) or \
            hist_row < 0 or hist_row + T > hist.shape[0] or not hist.is_contiguous():
        raise ValueError("hist must be a contiguous float32 tensor (rows, 8, B) holding rows hist_row .. hist_row + T")
    if len(rates) != npop or len(rate_rows) != npop:
        raise ValueError("one rates tensor and one first row per population of the block")
    r.state, r.hist, r.ctrl = state.data_ptr(), hist.data_ptr() + hist_row * _L.HIST_ROWS * B * 4, ctrl.data_ptr()
    r.diag = diag.data_ptr() if diag is not None else None
    j = 0
    for i in range(npop):
        q = r.pops[i]
        n, fr = int(q.n), rates[i]
        if fr.dtype != torch.float32 or fr.dim() != 3 or tuple(fr.shape[1:]) != (n, B) or not fr.is_contiguous() or \
                rate_rows[i] < 0 or rate_rows[i] + int(q.capacity_rows) > fr.shape[0]:
            raise ValueError(f"rates[{i}] must be a contiguous float32 tensor (rows, {n}, {B}) holding the block's rows")
        q.rates_base = fr.data_ptr() + rate_rows[i] * n * B * 4
        if q.spikes_base:
            sp = spikes[j]
            if sp.dtype != torch.uint8 or tuple(sp.shape[1:]) != (n, B) or not sp.is_contiguous() or \
                    spike_rows[j] < 0 or spike_rows[j] + int(q.capacity_rows) > sp.shape[0]:
                raise ValueError(f"spikes[{j}] must be a contiguous uint8 tensor (rows, {n}, {B}) holding the block's rows")
            q.spikes_base = sp.data_ptr() + spike_rows[j] * n * B
            j += 1
    rc = int(_L.lib.riab_simulate(C.c_void_p(streamer), C.byref(r), _L.current_stream()))
    if rc:
        e = _L.RiabError(f"riab_simulate failed with code {rc}: {_L.strerror(rc)}")
        e.code = rc
        raise e

"""GPU tests of the registered PyTorch operators (ratinabox_amd/ops.py: torch.ops.riab.*): called DIRECTLY on device
tensors against the reference goldens, checked with torch.library.opcheck (schema, fake implementation, mutation
annotations) and traced under torch.compile(fullgraph=True)."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634


@pytest.fixture(scope="module")
def riab():
    assert torch.cuda.is_available(), "these tests need the GPU"
    import ratinabox_amd
    import ratinabox_amd.ops  # noqa: F401
    return ratinabox_amd


def _pos_rows(pos):
    P = len(pos)
    Pp = (P + 3) // 4 * 4
    buf = np.zeros((2, Pp), dtype=np.float32)
    buf[:, :P] = np.asarray(pos, dtype=np.float64).T
    buf[:, P:] = buf[:, :1]
    return torch.from_numpy(buf).cuda(), P


def _pc_table(centres, widths):
    tab = np.empty((len(centres), 3))
    tab[:, :2] = centres
    tab[:, 2] = -LOG2E / (2 * np.asarray(widths, dtype=np.float64) ** 2)
    return torch.from_numpy(tab.astype(np.float32)).cuda()


def test_place_cells_op_vs_reference(riab):
    """torch.ops.riab.place_cells on raw tensors == the reference's PlaceCells.get_state (rates.npz), 1e-5."""
    g = gu.load("rates.npz")
    pos, P = _pos_rows(g["pos"])
    tab = _pc_table(g["pc_gaussian_centres"], g["pc_gaussian_widths"])
    out = torch.ops.riab.place_cells(pos, tab, None, [0.0, 1.0, 0.0, 1.0, 1.0], False, 0, 0, 0.2, 0.1, 2.0)
    np.testing.assert_allclose(out[:, :P].cpu().numpy(), g["pc_gaussian_rates"], rtol=1e-5, atol=1e-37)
    # line of sight through the maze's walls (float64 wall table, Environment.walls order)
    walls = torch.from_numpy(np.ascontiguousarray(g["maze_walls"].reshape(-1, 4))).cuda()
    tab = _pc_table(g["pc_los_centres"], 0.25 * np.ones(len(g["pc_los_centres"])))
    out = torch.ops.riab.place_cells(pos, tab, walls, [0.0, 1.0, 0.0, 1.0, 1.0], False, 0, 1, 0.25, 0.0, 1.0)
    np.testing.assert_allclose(out[:, :P].cpu().numpy(), g["pc_los_rates"], rtol=1e-5, atol=1e-37)


def test_ops_pass_opcheck(riab):
    g = gu.load("rates.npz")
    pos, P = _pos_rows(g["pos"][:64])
    tab = _pc_table(g["pc_gaussian_centres"], g["pc_gaussian_widths"])
    torch.library.opcheck(torch.ops.riab.place_cells.default,
                          (pos, tab, None, [0.0, 1.0, 0.0, 1.0, 1.0], False, 0, 0, 0.2, 0.0, 1.0))
    rates = torch.rand((3, 8, 64), device="cuda") * 50
    torch.library.opcheck(torch.ops.riab.spikes.default, (rates, None, 0.01, 7, 0, 1, 0))
    torch.library.opcheck(torch.ops.riab.spikes.default, (rates, torch.rand_like(rates), 0.01, 7, 0, 1, 0))
    x = torch.rand((2, 48, 64), device="cuda")
    wt = torch.zeros((48, 32), device="cuda")
    wt[:, :5] = torch.randn((48, 5), device="cuda")
    torch.library.opcheck(torch.ops.riab.feedforward.default, ([x], [wt], torch.zeros(5, device="cuda"), 3, [1.0, 0.0]))


def test_spikes_and_feedforward_ops_match_their_definitions(riab):
    rates = torch.rand((4, 16, 128), device="cuda") * 60
    u = torch.rand_like(rates)
    sp = torch.ops.riab.spikes(rates, u, 0.01, 0, 0, 0, 0)
    assert torch.equal(sp.bool(), u < torch.tensor(0.01, dtype=torch.float32, device="cuda") * rates)
    x = torch.rand((2, 40, 64), device="cuda")
    w = torch.randn((7, 40), device="cuda")
    wt = torch.zeros((40, 32), device="cuda")
    wt[:, :7] = w.t()
    b = torch.randn(7, device="cuda")
    out = torch.ops.riab.feedforward([x], [wt], b, 2, [1.5, 0.1])  # relu(gain, threshold): gain * max(x - threshold, 0)
    ref = 1.5 * torch.clamp(torch.einsum("mk,tkb->tmb", w.double(), x.double()) + b.double()[None, :, None] - 0.1, min=0)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)


def test_ops_trace_under_torch_compile(riab):
    """fullgraph=True: the operators' fake implementations and mutation annotations are enough for Dynamo /
    AOTAutograd to trace a function that mixes them with ordinary torch code (backend aot_eager: the traced graph
    runs as is; this image has no Triton for inductor's own kernels, and none is wanted)."""
    g = gu.load("rates.npz")
    pos, P = _pos_rows(g["pos"])
    tab = _pc_table(g["pc_gaussian_centres"], g["pc_gaussian_widths"])

    def population_vector(pos, tab):
        r = torch.ops.riab.place_cells(pos, tab, None, [0.0, 1.0, 0.0, 1.0, 1.0], False, 0, 0, 0.2, 0.0, 1.0)
        return (r / r.sum(0, keepdim=True).clamp_min(1e-12)).t() @ tab[:, :2]   # decoded positions (P, 2)

    eager = population_vector(pos, tab)
    compiled = torch.compile(population_vector, fullgraph=True, backend="aot_eager")(pos, tab)
    torch.testing.assert_close(compiled, eager, rtol=0, atol=0)

    # the in-place operator: state is mutated, the history rows are written
    np.random.seed(0)
    ag = riab.Agent(riab.Environment(), {"n_agents": 64, "dt": 0.01, "seed": 3})
    from ratinabox_amd import ops
    m = ops.motion_list(ag._motion(0.01, False, 1, {}))
    walls = ag.Environment.device_tables(ag._device)[1]
    env = [0.0, 1.0, 0.0, 1.0, 1.0]

    def two_steps(state, hist):
        torch.ops.riab.agent_step_(state, hist, None, walls, env, False, m, None, None, None, None, None, 3, 0, 0, 2)
        return hist[:, 0] + 0.0

    s_a, s_b = ag.state_tensor.clone(), ag.state_tensor.clone()
    h_a = torch.zeros((2, 8, 64), dtype=torch.float32, device="cuda")
    h_b = torch.zeros_like(h_a)
    x_a = two_steps(s_a, h_a)
    x_b = torch.compile(two_steps, fullgraph=True, backend="aot_eager")(s_b, h_b)
    assert torch.equal(s_a, s_b) and torch.equal(h_a, h_b) and torch.equal(x_a, x_b)
    assert not torch.equal(s_a, ag.state_tensor), "the operator must have advanced the state"


def test_simulate_operator_traces_and_matches_the_method(riab):
    """torch.ops.riab.simulate_ (riab_simulate as a mutating operator): opcheck on its schema / fake implementation /
    mutation annotations; a function that simulates K steps and reduces the rates compiles with fullgraph=True and
    gives what eager gives; and Agent.simulate() itself dispatches the operator (same rows as the direct ABI call)."""
    def world(seed=4, B=256):
        np.random.seed(seed)
        ag = riab.Agent(riab.Environment(), {"n_agents": B, "dt": 0.01, "seed": 21})
        np.random.seed(seed + 1)
        return ag, riab.PlaceCells(ag, {"n": 64, "save_spikes": False}), riab.GridCells(ag, {"n": 16, "save_spikes": True})

    K = 24
    ag, pcs, gcs = world()
    a = ag.simulate_args(K)
    torch.library.opcheck(torch.ops.riab.simulate_, (a.state.clone(), a.hist, a.rates, a.spikes, a.ctrl, a.diag, a.streamer, a.run,
                                                     0, a.rate_rows, a.spike_rows), test_utils=("test_schema", "test_faketensor"))

    def run_and_reduce(state, hist, rates, spikes, ctrl, diag, streamer, run):
        torch.ops.riab.simulate_(state, hist, rates, spikes, ctrl, diag, streamer, run, 0, [0, 0], [0])
        return rates[0].mean(dim=(0, 2)) + rates[1].sum() * 0.0, hist[-1, 0:2].clone()

    res = []
    for compiled in (False, True):
        ag, pcs, gcs = world()
        a = ag.simulate_args(K)
        fn = torch.compile(run_and_reduce, fullgraph=True, backend="aot_eager") if compiled else run_and_reduce
        m, last = fn(a.state, a.hist, a.rates, a.spikes, a.ctrl, a.diag, a.streamer, a.run)
        torch.cuda.synchronize()
        res.append((m.cpu(), last.cpu(), a.state.clone().cpu(), a.rates[0].cpu(), a.spikes[0].cpu()))
        assert ag.diagnostics["pipeline_timeouts"] == 0
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    # the method: same rows, through the operator (recorded) and through the direct call
    from torch.utils._python_dispatch import TorchDispatchMode
    names = []

    class Rec(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            names.append(str(func))
            return func(*args, **(kwargs or {}))

    got = []
    for direct in (False, True):
        ag, pcs, gcs = world()
        ag.DIRECT_NATIVE_CALL = direct
        with Rec():
            ag.simulate(K)
        torch.cuda.synchronize()
        got.append((ag.get_history_tensor().cpu(), pcs.get_history_tensors()[0].cpu(), gcs.get_history_tensors()[1].cpu()))
        assert any("riab.simulate_" in n for n in names), sorted(set(names))   # (a dispatch mode is active: the operator)
        names.clear()
    # without a dispatch mode the two routes are the attribute's choice; the rows are the same
    for direct in (False, True):
        ag, pcs, gcs = world()
        ag.DIRECT_NATIVE_CALL = direct
        ag.simulate(K)
        torch.cuda.synchronize()
        assert torch.equal(ag.get_history_tensor().cpu(), got[0][0]) and torch.equal(pcs.get_history_tensors()[0].cpu(), got[0][1])
    for x, y in zip(got[0], got[1]):
        assert torch.equal(x, y)
    assert torch.equal(got[0][1], res[0][3][:, :, :]) and torch.equal(got[0][0][-1, 0:2], res[0][1])


def test_get_state_and_update_go_through_the_operators(riab):
    """Neurons.get_state / Agent.update dispatch torch.ops.riab.* (recorded with a TorchDispatchMode), with
    unchanged results."""
    from torch.utils._python_dispatch import TorchDispatchMode
    names = []

    class Rec(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            names.append(str(func))
            return func(*args, **(kwargs or {}))

    np.random.seed(1)
    ag = riab.Agent(riab.Environment({"walls": [[[0.5, 0.0], [0.5, 0.4]]]}), {"n_agents": 8, "dt": 0.01})
    pcs = riab.PlaceCells(ag, {"n": 12, "wall_geometry": "euclidean"})
    gcs = riab.GridCells(ag, {"n": 6})
    hds = riab.HeadDirectionCells(ag, {"n": 5})
    bvs = riab.BoundaryVectorCells(ag, {"n": 4})
    q = np.array([[0.3, 0.4], [0.9, 0.1]])
    with Rec():
        fr = pcs.get_state(evaluate_at=None, pos=q)
        gcs.get_state(evaluate_at=None, pos=q)
        hds.get_state(evaluate_at=None, pos=q, head_direction=np.array([[1.0, 0.0], [0.0, 1.0]]))
        bvs.get_state(evaluate_at=None, pos=q)
        p0 = np.array(ag.pos)
        ag.update()
    for op in ("riab.place_cells", "riab.grid_cells", "riab.head_direction_cells", "riab.boundary_vector_cells",
               "riab.agent_step_"):
        assert any(op in n for n in names), f"{op} was not dispatched: {sorted(set(names))}"
    d = np.linalg.norm(pcs.place_cell_centres[:, None, :] - q[None], axis=-1)
    np.testing.assert_allclose(fr, np.exp(-d ** 2 / (2 * 0.2 ** 2)), rtol=1e-5)
    assert not np.allclose(np.array(ag.pos), p0)

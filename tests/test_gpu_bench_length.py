"""Bench-length parity of BASELINE configs[1] - [4] (cfg 2 - cfg 5) in the forms `bench.py` times them in.

`tests/test_gpu_fused.py::test_cfg2_full_length_run_in_every_form` runs cfg 2 through every form behind the started
gate.  Here: the worlds `bench.py` builds (`bench.build_world`: 4096 agents x 1024 PlaceCells; 4096 agents x (1024
GridCells + 256 BoundaryVectorCells) in the nine-wall maze; 4096 agents x 4096 PlaceCells; 8192 agents x (1024 PlaceCells
+ 512 GridCells + 256 BoundaryVectorCells + 256 HeadDirectionCells) with Poisson spikes), run the way `bench.py` runs
them (32 warm-up steps, a synchronisation, then ONE 1024-step `simulate()`), once through the native engine — the
row-following kernel in its reserving twelve-wave shape for cfg 2 / cfg 4, the chunk form for cfg 3, the populations
form for cfg 5: at full occupancy, where a row consumed before it was published would show — and once through the
Python-driven comparator (`RIAB_NO_NATIVE=1`).  Compared:

* the trajectories, bit for bit;
* per time row and population: float64 sum and sum of squares of the rates and the spike count, reduced on the
  device (90 GB of rows per run are not downloaded) — equal across the two engines;
* 32 agents spread over the batch on several rows — the first row, both sides of a mid-run boundary between two
  launches of the rate stage, the last row — against the oracle at 1e-5 (north_star), and on two of those rows every
  spike of every population against the exactly-specified rule on host-regenerated Philox uniforms.

Reference: ratinabox/Neurons.py:1172-1236 (GridCells), 1617-1744 (BoundaryVectorCells), 936-981 (PlaceCells),
2421-2485 (HeadDirectionCells), 681-687 (spikes)."""
import gc
import os
import sys

import numpy as np
import pytest
import torch

from oracle import riab_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RTOL = 1e-5
WARM, STEPS = 32, 1024


@pytest.fixture(scope="module")
def riab():
    assert torch.cuda.is_available(), "these tests need the GPU"
    import ratinabox_amd
    return ratinabox_amd


def _checksums(pops):
    """per population: (float64 row sums, row sums of squares, row spike counts or None), on the host"""
    out = []
    for p in pops:
        fr, sp = p.get_history_tensors()
        T = fr.shape[0]
        slab = max(1, (1 << 27) // (fr.shape[1] * fr.shape[2]))      # <= 1 GiB of float64 at a time
        s1, s2, sc = [], [], []
        for i in range(0, T, slab):
            x = fr[i:i + slab].to(torch.float64)
            s1.append(x.sum((1, 2)))
            s2.append((x * x).sum((1, 2)))
            if sp.numel():
                sc.append(sp[i:i + slab].sum((1, 2), dtype=torch.int64))
            del x
        out.append((torch.cat(s1).cpu(), torch.cat(s2).cpu(), torch.cat(sc).cpu() if sc else None))
    return out


def _assert_rates(got, ref, what, floor=1.0):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    assert got.shape == ref.shape, what
    bad = np.abs(got - ref) > RTOL * np.abs(ref) + floor * RTOL + 1e-37
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} outside 1e-5; worst abs err {np.abs(got - ref).max():.3e}"


def _oracle_rows(name, cfg, env, ag, pops, rows, sel, spike_rows, seed=1234):
    """sampled agents of `rows` against the oracle; every spike of `spike_rows` against the exact rule"""
    traj = ag.get_history_tensor()
    oenv = orc.EnvSpec()   # (PlaceCells only occur in the open box of cfg 5, with euclidean distances)
    assert not (cfg["place"] and cfg["walls"])
    for r in rows:
        row = traj[r].cpu().numpy()
        pos = row[0:2, sel].T.astype(np.float64)
        hd = row[4:6, sel].T.astype(np.float64)
        assert (pos > 0).all() and (pos < 1).all(), "an agent left the box"
        for p in pops:
            kind = type(p).__name__
            if kind == "PlaceCells":
                ref = orc.place_cells(oenv, pos, p.place_cell_centres, p.place_cell_widths)
            elif kind == "GridCells":
                ref = orc.grid_cells(pos, p.gridscales, p.phase_offsets, p.w)
            elif kind == "BoundaryVectorCells":
                ref = orc.bvc(pos, env.walls, p.tuning_distances, p.tuning_angles, p.sigma_distances, p.sigma_angles)
            else:
                ref = orc.head_direction_cells(hd, int(p.n))
            fr = p.get_history_tensors()[0]
            _assert_rates(fr[r][:, sel].cpu().numpy(), ref, f"{name} row {r} {kind}")
    for r in spike_rows:
        for p in pops:
            fr, sp = p.get_history_tensors()
            if not sp.numel():
                continue
            u = orc.spike_uniforms(seed, r + 1, p.pop_id, int(p.n), ag._Bp)     # Neurons.update after the (r + 1)-th Agent.update
            want = orc.spikes_f32(fr[r].cpu().numpy(), u, 0.01)
            assert np.array_equal(sp[r].cpu().numpy().astype(bool), want), f"{name} row {r} {type(p).__name__}: spikes"


@pytest.mark.parametrize("name,form", [("cfg2", "one-kernel"), ("cfg3", "chunks"), ("cfg4", "one-kernel"), ("cfg5", "populations")])
def test_bench_length_run_native_against_comparator_and_oracle(riab, name, form):
    import bench
    cfg = bench.CONFIGS[name]
    gc.collect()
    torch.cuda.empty_cache()
    res = {}
    for engine, envs in (("native", {}), ("python", {"RIAB_NO_NATIVE": "1"})):
        os.environ.update(envs)
        try:
            env, ag, pops = bench.build_world(riab, cfg, 0)
            ag.simulate(WARM)
            torch.cuda.synchronize()   # (as in bench.py: the timed call finds the stream idle — and, with several populations,
            ag.simulate(STEPS)         # reads the warm-up call's device-clock stamps before it chooses its form)
            torch.cuda.synchronize()
            d = ag.diagnostics
            assert d["bounce_saturations"] == 0
            if engine == "native":
                assert ag.engine_runs["native"] == 2 and ag.last_rate_stage_form() == form, (ag.engine_runs, ag.last_rate_stage_form())
                assert d["pipeline_timeouts"] == 0 and d["pipeline_serialised"] <= 1, d   # (queue sharing shows on EVERY call; a lone count is a device hiccup)
                info = ag.pipeline_info()
                if len(pops) > 1:
                    assert info["form_selection"]["measured"], "the form was chosen without the measured step time"
                else:   # one store-bound population, more than 256 steps: the strict mode by default (opening kernel, started
                    assert info["strict_last_call"] and info["launches_last_call"] == 4, info   # gate, trajectory, rate kernel)
                B = cfg["agents"]
                sel = np.arange(0, B, B // 32) + 5
                # rows of the whole history (warm-up call first): the first row, the long call's first row, both sides of a
                # boundary between two launches of the rate stage inside the long call (chunk form in a solid rectangular
                # room: chunks of 16, 28, 44, 64, 96, 128, 128 ...: rows 375 | 376 of the call; populations form: the
                # row-following kernel has no boundary, the others' launches are 1024 rows: mid-run rows), the last
                rows = [0, WARM, WARM + 375, WARM + 376, WARM + STEPS - 1]
                _oracle_rows(name, cfg, env, ag, pops, rows, sel, spike_rows=[WARM + 376, WARM + STEPS - 1] if cfg["spikes"] else [])
            else:
                assert ag.engine_runs["chunks"] == 2 and ag.engine_runs["native"] == 0
            traj = ag.get_history_tensor()
            assert traj.shape[0] == WARM + STEPS
            res[engine] = (traj.cpu(), _checksums(pops))
            del env, ag, pops, traj
            gc.collect()               # (Agent <-> Neurons <-> history views are reference cycles: 70-90 GB of rows each)
            torch.cuda.empty_cache()
        finally:
            for k in envs:
                os.environ.pop(k, None)
    assert torch.equal(res["native"][0], res["python"][0]), "trajectories differ"
    for i, (a, b) in enumerate(zip(res["native"][1], res["python"][1])):
        for what, x, y in zip(("sum", "sum of squares", "spike count"), a, b):
            if x is None:
                assert y is None
                continue
            bad = torch.nonzero(x != y).flatten()
            assert bad.numel() == 0, f"{name} population {i}: per-row {what} differs on rows {bad[:8].tolist()} ..."
    if cfg["spikes"]:   # property: spike count of the run within 5 sigma of sum(dt * rate)
        for (s1, _s2, sc) in res["native"][1]:
            expected = 0.01 * float(s1.sum())
            assert abs(float(sc.sum()) - expected) < 5 * np.sqrt(expected) + 1

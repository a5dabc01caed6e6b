"""CPU baseline worker for bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE, like the rest of oracle/: the
float64 NumPy restatement of the path, never imported by the product).

    python -m oracle.cpu_bench --agents 16 --cells 1024 --seconds 8 [--walls-json ...] [--spikes]

steps `agents` agents (Agent.update + PlaceCells.update, the cfg-2 shape of SURVEY.md §8(d)) for about `seconds`
seconds on ONE core and prints `agent_steps wall_seconds cpu_seconds`.  bench.py starts one such process per host core and sums the
rates; each process pins BLAS/OpenMP to a single thread."""
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_v] = "1"

import argparse  # noqa: E402
import json  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402


def run(agents, cells, seconds, walls, spikes, seed=0):
    from oracle import riab_oracle as orc
    rs = np.random.RandomState(seed)
    env = orc.EnvSpec(walls=np.asarray(walls, dtype=float).reshape(-1, 2, 2))
    st = orc.init_state(env, agents, 0.08, rs)
    side = int(np.sqrt(max(cells, 1)))
    gx = (np.arange(side) + 0.5) / side
    centres = np.stack(np.meshgrid(gx, gx), -1).reshape(-1, 2)
    steps = 0
    t0 = time.perf_counter()
    while True:
        z = rs.standard_normal((2, agents))
        st = orc.agent_step(env, st, 0.01, z[0], z[1])
        if cells:
            fr = orc.place_cells(env, st["pos"], centres, 0.2)
            if spikes:
                orc.spikes_ref(fr, rs.random_sample(fr.shape), 0.01)
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds:
            return agents * steps, el


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=16)
    ap.add_argument("--cells", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--walls-json", default="[]")
    ap.add_argument("--spikes", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    c0 = time.process_time()
    n, el = run(a.agents, a.cells, a.seconds, json.loads(a.walls_json), a.spikes, a.seed)
    # (agent-steps, wall seconds, CPU seconds of this process: CPU / wall well below 1 = the worker did not have a core)
    print(n, el, time.process_time() - c0, flush=True)

"""One-kernel form against chunk form of the rate stage for one PlaceCells population of n cells, K steps per call
(4096 agents): RIAB_POLL_MAX=8192 python tools/form_probe.py K n [n ...]   vs   python tools/form_probe.py K n [n ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
K = int(sys.argv[1])
for n in map(int, sys.argv[2:]):
    np.random.seed(0)
    ag = riab.Agent(riab.Environment({}), {"n_agents": 4096, "dt": 0.01, "seed": 1})
    pcs = riab.PlaceCells(ag, {"n": n, "widths": 0.2, "save_spikes": False})
    ag.simulate(K); torch.cuda.synchronize()
    ts = []
    for r in range(6):
        ag.reset_history(); pcs.reset_history()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ag.simulate(K); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    m = float(np.median(ts[1:]))
    print("n=%5d K=%d poll_max=%s: %.3f ms  %.1f M agent-steps/s  %.2f TB/s  timeouts %s" % (
        n, K, os.environ.get("RIAB_POLL_MAX", "default"), m * 1e3, 4096 * K / m / 1e6, 4096 * K * n * 4 / m / 1e12,
        ag.diagnostics.get("pipeline_timeouts", 0)))
    del ag, pcs
    torch.cuda.empty_cache()

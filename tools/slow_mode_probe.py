"""Does a freshly created Agent sometimes run its two kernels one after the other?  Creates worlds in a row (cfg 2),
times 60 regions of simulate(20) each, and prints the device-clock stamps of the last call: trajectory start / end,
rate kernel first-wave start / last-wave end (us after the trajectory's start), and the serialised counter.
    python tools/slow_mode_probe.py [n_worlds] [keep|drop]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ratinabox_amd as riab
L = riab._lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
keep = (sys.argv[2] if len(sys.argv) > 2 else "drop") == "keep"
K = 20
held = []
import warnings
warnings.simplefilter("ignore")
for i in range(N):
    env, ag, pops = bench.build_world(riab, bench.CONFIGS["cfg2"], 0)
    ag._time_rate_kernel = True
    ag._timed_population = pops[0]
    ts = []
    for r in range(70):
        ag.reset_history(); pops[0].reset_history(); ag.preallocate_history(K)
        torch.cuda.synchronize(); t0 = time.perf_counter(); ag.simulate(K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    st = ag._ctrl[8:16].cpu().view(torch.int64).numpy().astype(np.float64)   # STAMPS (rate start, end), TRAJ_STAMPS (start, end)
    base = st[2]
    d = ag.diagnostics
    print("world %2d: median %6.1f us (min %6.1f) | traj 0 .. %5.1f | rate kernel %5.1f .. %5.1f | serialised %d of 70 | %s" % (
        i, 1e6 * np.median(ts[10:]), 1e6 * min(ts[10:]), (st[3] - base) / 100, (st[0] - base) / 100, (st[1] - base) / 100,
        d["pipeline_serialised"], ag.pipeline_info()), flush=True)
    if keep:
        held.append((env, ag, pops))
    else:
        del env, ag, pops
        torch.cuda.empty_cache()

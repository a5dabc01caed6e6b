"""Per-row time line of the flag-coupled pipeline at the driver's shape (4096 agents x 1024 PlaceCells, K steps per
call) from a -DRIAB_PIPE_PROFILE build (tools/build_pipe_profile.sh): when trajectory workgroups 0 and 63 published
each row, when the rate kernel's first workgroup of a row was running, when it had its row, when the row's last
workgroup had its stores acknowledged — device constant clock (100 MHz), relative to the trajectory kernel's start.

    RIAB_HIP_LIB=tools/exp/libpipe_prof.so python tools/pipe_profile.py [K]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
assert 1 <= K <= 63, "the profile build keeps 64 stamps per kind"
np.random.seed(0)
env = riab.Environment({})
ag = riab.Agent(env, {"n_agents": 4096, "dt": 0.01, "seed": 1234})
pcs = riab.PlaceCells(ag, {"n": 1024, "widths": 0.2, "save_spikes": False})
ag.simulate(K)
torch.cuda.synchronize()
big = torch.zeros(8192, dtype=torch.int32, device="cuda")
big[:ag._ctrl.numel()] = ag._ctrl          # (the started count and the progress words go on from where they are)
ag._ctrl = big
ag._snap = None
ag._run_cache = None
dbg = big[2048:2048 + 2 * 5 * 64].view(torch.int64)
rows = []
tot = []
for rep in range(12):
    ag.reset_history(); pcs.reset_history()
    dbg.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ag.simulate(K)
    torch.cuda.synchronize()
    tot.append((time.perf_counter() - t0) * 1e6)
    d = dbg.cpu().numpy().reshape(5, 64).astype(np.float64)
    rows.append(d)
d = np.stack(rows[2:])
base = d[:, 0, 0][:, None]                      # trajectory workgroup 0's tail wave starts
us = lambda x: (x - base) / 100.0                # 100 MHz -> us
pub0, run, got, done, pub63 = (np.median(us(d[:, k, :]), 0) for k in (0, 1, 2, 3, 4))
print("host region: median %.1f us" % np.median(tot[2:]))
print("row | published by wg0 / wg63 | rate: first wg running / has its row | row's last wg done | lag done - published")
f = lambda x: "%7.1f" % x if abs(x) < 1e6 else "      -"   # ('-': published together with the next stamped row)
for t in range(K):
    print("%3d | %s %s | %7.1f %7.1f | %7.1f | %s" % (t, f(pub0[t + 1]), f(pub63[t + 1]), run[t], got[t], done[t], f(done[t] - pub0[t + 1])))

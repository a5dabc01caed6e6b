"""The trajectory kernel alone (no rate stage): one riab_agent_step launch of T steps for 4096 agents in the open box,
timed with events on the launch stream.  `RIAB_HIP_LIB=<variant .so>` selects a build (tools/build_variants.sh: the
ablation builds of riab_traj4_kernel.h), RIAB_TRAJ2=1 the two-wave kernel of round 1."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
L = riab._lib
np.random.seed(0)
walls = [] if len(sys.argv) < 2 or sys.argv[1] != "maze" else [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]
ag = riab.Agent(riab.Environment({"walls": walls}), {"n_agents": 4096, "dt": 0.01})
hist = torch.empty((1024, 8, 4096), dtype=torch.float32, device="cuda")
out = []
for T in (8, 20, 64, 256, 1024):
    s = torch.cuda.current_stream()
    raw = L.C.c_void_p(s.cuda_stream)
    for _ in range(3):
        ag._advance(T, None, None, 1, {}, hist_view=hist[:T], stream=raw)
    torch.cuda.synchronize()
    ms = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        ag._advance(T, None, None, 1, {}, hist_view=hist[:T], stream=raw)
        e1.record(s)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    out.append("T=%d %.1f us (%.2f/step)" % (T, 1e3 * np.median(ms), 1e3 * np.median(ms) / T))
print(os.environ.get("RIAB_HIP_LIB", "default")[-24:], "TRAJ2" if os.environ.get("RIAB_TRAJ2") else "", " | ".join(out), flush=True)

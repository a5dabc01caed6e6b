import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import bench, ratinabox_amd as riab
import torch.distributed as dist
K, R = 20, 300
use = os.environ.get("NCCL") == "1"
if use:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
env, ag, pops = bench.build_world(riab, bench.CONFIGS["cfg2"], 0)
ag._time_rate_kernel = False
def fresh():
    ag.reset_history()
    for p in pops: p.reset_history()
    ag.preallocate_history(K)
for _ in range(10):
    fresh(); ag.simulate(K)
torch.cuda.synchronize()
for mode in ("no barrier", "barrier"):
    a, b, c = [], [], []
    for _ in range(R):
        fresh(); torch.cuda.synchronize()
        if use and mode == "barrier":
            dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ag.simulate(K); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        a.append(t1 - t0); b.append(t2 - t1)
        t3 = time.perf_counter(); torch.cuda.synchronize(); c.append(time.perf_counter() - t3)
    tot = np.add(a, b)
    print("nccl=%s %-10s call %5.1f us sync %5.1f us total median %6.1f min %6.1f  idle sync %.1f us" % (use, mode, 1e6*np.median(a), 1e6*np.median(b), 1e6*np.median(tot), 1e6*tot.min(), 1e6*np.median(c)), flush=True)
if use: dist.destroy_process_group()

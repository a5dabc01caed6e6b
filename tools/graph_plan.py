"""Does a captured hipGraph beat the step plan's direct launches?  64 closed-loop steps (motion + PlaceCells rates at
cfg 2) captured once and replayed, against riab_plan_step.  [MI355X] 12.9 us/step replayed vs 12.5 us/step direct."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ratinabox_amd as riab
np.random.seed(0)
env = riab.Environment()
Ag = riab.Agent(env, {"n_agents": 4096, "dt": 0.01, "seed": 1, "save_history": False})
PCs = riab.PlaceCells(Ag, {"n": 1024})
PCs.save_history = False
plan = Ag.make_step_plan(capacity=1024)
def timeit(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
plan.step(64); torch.cuda.synchronize()
print("direct plan.step(1) x1024: %.2f us/step" % timeit(lambda: [plan.step(1) for _ in range(1024)], 1024))
print("direct plan.step(64) x16: %.2f us/step" % timeit(lambda: [plan.step(64) for _ in range(16)], 1024))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plan.step(64)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        plan.step(64)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("graph of 64 steps x16: %.2f us/step" % timeit(lambda: [g.replay() for _ in range(16)], 1024))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])

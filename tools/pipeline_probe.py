"""Where does the fused pipeline's time go?  Times simulate() with and without the rate
kernels, and the trajectory kernel's launches with events on its own stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab

def world():
    np.random.seed(0)
    env = riab.Environment()
    ag = riab.Agent(env, {"n_agents": 4096, "dt": 0.01, "seed": 1})
    pc = riab.PlaceCells(ag, {"n": 1024, "wall_geometry": "euclidean", "save_spikes": False})
    return ag, pc

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

K = 1024
for chunk in (64, 128, 256, 512):
    ag, pc = world()
    ag.simulate(128, chunk=chunk); torch.cuda.synchronize(); ag.reset_history(); pc.reset_history()
    ag.preallocate_history(K)
    a = timed(lambda: ag.simulate(K, chunk=chunk, neurons=[]))
    ag.reset_history(); pc.reset_history(); ag.preallocate_history(K)
    b = timed(lambda: ag.simulate(K, chunk=chunk))
    t0 = time.perf_counter(); 
    ag.reset_history(); pc.reset_history(); ag.preallocate_history(K)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ag.simulate(K, chunk=chunk); host = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
    print(f"chunk={chunk}: traj only {a:.2f} ms ({a/K*1e3:.2f} us/step), traj+rates {b:.2f} ms ({4096*K/b/1e3:.0f} M/s), host issue time {host:.2f} ms", flush=True)

import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench, ratinabox_amd as riab
L = riab._lib
env, ag, pops = bench.build_world(riab, bench.CONFIGS["cfg2"], 0)
N = pops[0]
p1 = N._population(); p2 = N._population()
print("same object:", p1 is p2, "cache:", "_pop_cache" in N.__dict__)
f = N._call(None, None); vals = tuple(f.values()); hit = N.__dict__.get("_pop_cache")
print("hit", hit is not None)
if hit is not None:
    print([ (type(a).__name__, (a is b) if torch.is_tensor(a) else (a == b)) for a, b in zip(hit[0], vals)], hit[1], (float(N.min_fr), float(N.max_fr)), N.noise_std)
def t(label, fn, reps=2000):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    print("%-40s %6.2f us" % (label, 1e6 * (time.perf_counter() - t0) / reps), flush=True)
t("_population", lambda: N._population())
t("_call", lambda: N._call(None, None))
t("tuple(f.values())", lambda: tuple(f.values()))
t("noise_std==0", lambda: N.noise_std == 0)
t("min/max", lambda: (float(N.min_fr), float(N.max_fr)))
t("all(...)", lambda: all(a is b if torch.is_tensor(a) else a == b for a, b in zip(hit[0], vals)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): N._population()
pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(12)

"""Where the timed region of `bench.py --steps K` goes: host time of Agent.simulate(K) (until the call returns, the
kernels are in flight) against the wait in torch.cuda.synchronize() that follows.  cfg 2."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench, ratinabox_amd as riab
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = bench.CONFIGS["cfg2"]
env, ag, pops = bench.build_world(riab, cfg, 0)
ag.preallocate_history(K * 400)
for _ in range(5):
    ag.simulate(K)
torch.cuda.synchronize()
a, b = [], []
for _ in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ag.simulate(K); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a.append(t1 - t0); b.append(t2 - t1)
print("K=%d  simulate() returns after %.1f us (min %.1f); synchronize() waits another %.1f us (min %.1f); total median %.1f us" % (
    K, 1e6 * np.median(a), 1e6 * min(a), 1e6 * np.median(b), 1e6 * min(b), 1e6 * np.median(np.add(a, b))))
# the native call itself inside simulate(): time before it, inside it, after it
real = riab._lib.lib.riab_simulate
marks = []
class _Lib:
    def __getattr__(self, k):
        return getattr(_orig, k)
def wrapped(*a):
    t = time.perf_counter(); rc = real(*a); marks.append((t, time.perf_counter())); return rc
_orig = riab._lib.lib
proxy = _Lib(); proxy.__dict__["riab_simulate"] = wrapped
riab._lib.lib = proxy
import ratinabox_amd.Agent as A
pre, nat, post = [], [], []
for _ in range(300):
    torch.cuda.synchronize(); marks.clear()
    t0 = time.perf_counter(); ag.simulate(K); t1 = time.perf_counter()
    pre.append(marks[0][0] - t0); nat.append(marks[0][1] - marks[0][0]); post.append(t1 - marks[0][1])
riab._lib.lib = _orig
print("inside simulate(): %.1f us before the native call, %.1f us in it, %.1f us after it" % (1e6 * np.median(pre), 1e6 * np.median(nat), 1e6 * np.median(post)))
# an empty synchronize and an empty kernel round trip, for scale
x = torch.zeros(1, device="cuda")
c = []
for _ in range(300):
    torch.cuda.synchronize(); t0 = time.perf_counter(); x.add_(1); torch.cuda.synchronize(); c.append(time.perf_counter() - t0)
print("one tiny torch kernel + synchronize: %.1f us" % (1e6 * np.median(c)))
L = riab._lib
c = []
for _ in range(300):
    torch.cuda.synchronize(); t0 = time.perf_counter(); L.lib.riab_fill(L.ptr(x), 0, 1.0, L.current_stream()); torch.cuda.synchronize(); c.append(time.perf_counter() - t0)
print("empty ctypes call + synchronize: %.1f us" % (1e6 * np.median(c)))

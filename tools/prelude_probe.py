"""Host-side cost of the pieces of Agent.simulate() in front of the native call (cfg 2), microseconds each."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench, ratinabox_amd as riab
L = riab._lib
cfg = bench.CONFIGS["cfg2"]
env, ag, pops = bench.build_world(riab, cfg, 0)
N = pops[0]
ag.preallocate_history(64)
ag.simulate(20); torch.cuda.synchronize()
def t(label, fn, reps=2000):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print("%-46s %6.2f us" % (label, 1e6 * (time.perf_counter() - t0) / reps), flush=True)
t("N._population()", lambda: N._population())
t("N._call(None, None)", lambda: N._call(None, None))
t("ag._motion(dt, False, 1, {})", lambda: ag._motion(ag.dt, False, 1, {}))
t("env.device_tables(device)", lambda: env.device_tables(ag._device))
t("_L.current_stream()", lambda: L.current_stream())
t("_L.env('RIAB_NO_NATIVE')", lambda: L.env("RIAB_NO_NATIVE"))
def rr():
    c, s = ag._hist.reserve_at(20); ag._hist.unreserve(20)
t("hist.reserve_at + unreserve", rr)
def rn():
    o = N._reserve_rows(20, ring=128); N._unreserve_rows(o, 20)
t("N._reserve_rows + unreserve", rn)
arr = (L.RiabPopulation * 1)()
pop = N._population()
t("memmove of one RiabPopulation", lambda: L.C.memmove(L.C.byref(arr, 0), L.C.byref(pop), L.POP_SIZE))
e, _w = env.device_tables(ag._device); m = ag._motion(ag.dt, False, 1, {})
t("C.pointer(env), C.pointer(m)", lambda: (L.C.pointer(e), L.C.pointer(m)))
run = L.RiabSimulate()
def fill():
    run.state, run.B, run.agent_id0 = ag._state.data_ptr(), ag._Bp, 0
    run.seed, run.n_pops = 1234, 1
    run.diag, run.ctrl, run.timed_pop = ag._diag.data_ptr(), ag._ctrl.data_ptr(), -1
    run.step0, run.T, run.hist = 5, 20, 12345
t("filling the RiabSimulate block", fill)
t("torch.cuda.synchronize() (idle)", lambda: torch.cuda.synchronize())
t("riab_fill of 0 bytes (one ctypes call, no launch?)", lambda: L.lib.riab_fill(None, 0, 1.0, L.current_stream()))
x = torch.zeros(16, device="cuda")
def fl():
    L.lib.riab_fill(L.ptr(x), 64, 1.0, L.current_stream())
t("riab_fill of 64 bytes (one launch), no sync", fl, reps=500)
torch.cuda.synchronize()
def fls():
    L.lib.riab_fill(L.ptr(x), 64, 1.0, L.current_stream()); torch.cuda.synchronize()
t("riab_fill of 64 bytes + synchronize", fls, reps=500)

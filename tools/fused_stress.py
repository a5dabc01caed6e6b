"""Stress of the flag-coupled pipelines: thousands of back-to-back simulate() calls at the bench shape; every call's
rates are reduced to a checksum on the device and compared, call by call, with the Python-driven chunked pipeline
(a row consumed before it was published, or not at all, changes the checksum).  python tools/fused_stress.py [calls]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab

def run(native, calls, K, B=4096, n=1024, two=False, idle=False):
    os.environ["RIAB_NO_FUSED"] = "0" if native else "1"
    os.environ["RIAB_NO_NATIVE"] = "0" if native else "1"
    np.random.seed(0)
    env = riab.Environment()
    ag = riab.Agent(env, {"n_agents": B, "dt": 0.01, "seed": 5})
    pops = [riab.PlaceCells(ag, {"n": n, "save_spikes": False})]
    if two == "bvc":     # a store-bound and an issue-bound population: the populations form of the rate stage
        pops.append(riab.BoundaryVectorCells(ag, {"n": 16, "save_spikes": True}))
    elif two:
        pops.append(riab.GridCells(ag, {"n": 256, "save_spikes": False}))
    sums = torch.zeros((calls, len(pops)), dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    for i in range(calls):
        per = max(1, 1200 // K)   # (about 20 GB of rate rows at a time)
        if i % per == 0:
            ag.reset_history()
            for p in pops:
                p.reset_history()
            ag.preallocate_history(per * K)
        if idle:   # the caller's stream idle at every call: the short-call road (two launches, the reserving shape)
            torch.cuda.synchronize()
        ag.simulate(K)
        for j, p in enumerate(pops):
            fr, _ = p.get_history_tensors()
            sums[i, j] = fr[-K:].sum(dtype=torch.float64)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d = ag.diagnostics
    return sums.cpu().numpy(), ag.state_tensor.cpu().numpy(), d, el, ag.last_rate_stage_form(), ag.pipeline_info()

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for K, two, c in ((20, False, calls), (64, False, calls // 4), (300, False, calls // 16), (1100, False, calls // 64),
                  (20, True, calls // 4), (300, True, calls // 16), (1100, True, calls // 64),
                  (20, "bvc", calls // 8), (300, "bvc", calls // 32), (1100, "bvc", calls // 128)):
    a, sa, da, ta, form, _info = run(True, c, K, two=two)
    b, sb, db, tb, _, _ = run(False, c, K, two=two)
    ok = np.array_equal(a, b) and np.array_equal(sa, sb)
    if not two and K <= 64:   # the same calls from an idle stream: the reserving (twelve-wave) shape, no started gate
        a2, sa2, da2, ta2, form2, info2 = run(True, c, K, two=two, idle=True)
        ok2 = np.array_equal(a2, b) and np.array_equal(sa2, sb)
        print("K=%-4d populations=place calls=%-5d idle stream: form=%s launches per call=%d identical=%s timeouts=%s serialised=%s (%.1f s)" % (
            K, c, form2, info2["launches_last_call"], ok2, da2.get("pipeline_timeouts"), da2.get("pipeline_serialised"), ta2), flush=True)
        assert ok2 and da2.get("pipeline_timeouts", 0) == 0 and info2["launches_last_call"] == 2
    print("K=%-4d populations=%s calls=%-5d form=%-11s identical=%s timeouts=%s  (%.1f s native, %.1f s chunked)" % (
        K, {False: "place", True: "place+grid", "bvc": "place+bvc"}[two], c, form, ok, da.get("pipeline_timeouts"), ta, tb), flush=True)
    assert ok and da.get("pipeline_timeouts", 0) == 0
print("OK")

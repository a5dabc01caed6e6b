"""Where the driver's 20-step region goes, variant by variant (cfg 2): host time of simulate(K) until it returns,
the wait in torch.cuda.synchronize() after it, and the total, for
  * the rate kernel's launch with / without its start / stop events (`timing`),
  * the started gate forced / skipped when the caller's stream is idle,
  * hipDeviceScheduleSpin (the host spins in synchronize() instead of blocking after 100 us).
Run on the GPU box: python tools/k20_probe.py [K] [repeats]"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import ratinabox_amd as riab  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 400
L = riab._lib


def world():
    cfg = bench.CONFIGS["cfg2"]
    env, ag, pops = bench.build_world(riab, cfg, 0)
    return ag, pops


def fresh(ag, pops):
    ag.reset_history()
    for p in pops:
        p.reset_history()
    ag.preallocate_history(K)


def measure(ag, pops, label):
    for _ in range(10):
        fresh(ag, pops)
        ag.simulate(K)
    torch.cuda.synchronize()
    a, b = [], []
    for _ in range(R):
        fresh(ag, pops)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ag.simulate(K)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        a.append(t1 - t0)
        b.append(t2 - t1)
    tot = np.add(a, b)
    kms = ag.last_rate_kernel_ms() if ag._time_rate_kernel else None
    print("%-44s call %5.1f us  sync %5.1f us  total median %6.1f  min %6.1f  p90 %6.1f  -> %.3f G/s  rate kernel %s ms" % (
        label, 1e6 * np.median(a), 1e6 * np.median(b), 1e6 * np.median(tot), 1e6 * tot.min(), 1e6 * np.percentile(tot, 90),
        4096 * K / np.median(tot) / 1e9, "%.4f" % kms if kms else "-"), flush=True)
    assert ag.diagnostics["pipeline_timeouts"] == 0


def main():
    for spin in (0, 1):
        if spin:
            rc = L.lib.riab_host_wait_spin(1)  # hipSetDeviceFlags(hipDeviceScheduleSpin)
            print("riab_host_wait_spin(1) ->", rc, flush=True)
        for timing in (True, "events", False):
            for gate in (1, 0):
                ag, pops = world()
                ag._time_rate_kernel = timing
                fresh(ag, pops)
                ag.simulate(K)
                L.lib.riab_streamer_configure(ag._streamer, L.STREAMER_OPT_GATE, 0 if gate else 1)
                measure(ag, pops, "spin=%d timing=%s gate=%s" % (spin, timing, "always" if gate else "when-busy"))
                del ag, pops
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

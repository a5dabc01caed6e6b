"""Round 4: where the driver's 20-step region goes, variant by variant (cfg 2, exactly bench.py's timed region: fresh
history rows, synchronize, simulate(K), synchronize).  Per variant: host time until simulate() returns, the wait in
synchronize(), the total (median / min / p90), the rate kernel by the device clock; then the same call split into
Python before the native call / the native call / Python after it.
    python tools/r04_probe.py [K] [repeats]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ratinabox_amd as riab  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 300
L = riab._lib


def world():
    env, ag, pops = bench.build_world(riab, bench.CONFIGS["cfg2"], 0)
    return ag, pops


def fresh(ag, pops):
    ag.reset_history()
    for p in pops:
        p.reset_history()
    ag.preallocate_history(K)


def measure(label, env=None, opts=None, timing=True):
    for k, v in (env or {}).items():
        os.environ[k] = v
    old = {k: L.set_option(k, v) for k, v in (opts or {}).items()}
    try:
        ag, pops = world()
        ag._time_rate_kernel = timing
        ag._timed_population = pops[0]
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
        kms = ag.last_rate_kernel_ms() if timing else None
        n_launch = L.lib.riab_streamer_info(ag._streamer, 3)
        # the call split: before / inside / after the native call
        real = L.lib.riab_simulate
        marks = []

        class Proxy:
            def __getattr__(self, k):
                return getattr(orig, k)

        def wrapped(*args):
            t = time.perf_counter()
            rc = real(*args)
            marks.append((t, time.perf_counter()))
            return rc
        orig = L.lib
        proxy = Proxy()
        proxy.__dict__["riab_simulate"] = wrapped
        L.lib = proxy
        pre, nat, post = [], [], []
        try:
            for _ in range(R):
                fresh(ag, pops)
                torch.cuda.synchronize()
                marks.clear()
                t0 = time.perf_counter()
                ag.simulate(K)
                t1 = time.perf_counter()
                pre.append(marks[0][0] - t0)
                nat.append(marks[0][1] - marks[0][0])
                post.append(t1 - marks[0][1])
        finally:
            L.lib = orig
        torch.cuda.synchronize()
        d = ag.diagnostics
        assert d["pipeline_timeouts"] == 0, d
        if d["pipeline_serialised"]:
            label += " [serialised %d]" % d["pipeline_serialised"]
        print("%-58s launches %d | call %5.1f sync %5.1f | total med %6.1f min %6.1f p90 %6.1f us -> %.3f G/s | kernel %s us | "
              "py-pre %4.1f native %4.1f py-post %4.1f" % (
                  label, n_launch, 1e6 * np.median(a), 1e6 * np.median(b), 1e6 * np.median(tot), 1e6 * tot.min(),
                  1e6 * np.percentile(tot, 90), 4096 * K / np.median(tot) / 1e9, "%.1f" % (1e3 * kms) if kms else "-",
                  1e6 * np.median(pre), 1e6 * np.median(nat), 1e6 * np.median(post)), flush=True)
        del ag, pops
        torch.cuda.empty_cache()
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
        for k, v in old.items():
            L.set_option(k, v)


def main():
    print("K = %d, %d regions per variant" % (K, R), flush=True)
    for rnd in range(2):   # twice: the boxes drift
        measure("A gate=always, content re-check (round 3's road)", {"RIAB_GATE": "always", "RIAB_NO_FAST_REPEAT": "1"})
        measure("B gate=always, watch list", {"RIAB_GATE": "always"})
        measure("C gate=reserved (12-wave shape, 2 launches), watch list", {"RIAB_GATE": "reserved"})
        measure("D gate=when_busy (4-wave shape, 2 launches), watch list", {"RIAB_GATE": "when_busy"})
        measure("E = C + first 8 rows published singly", {"RIAB_GATE": "reserved"}, {"pub_single_rows": 8})
        measure("F = C + poll sleep <= 16", {"RIAB_GATE": "reserved"}, {"poll_sleep": 16})
        measure("G = C + 8 single rows + poll sleep <= 16", {"RIAB_GATE": "reserved"}, {"pub_single_rows": 8, "poll_sleep": 16})
        measure("H = D + 8 single rows + poll sleep <= 16", {"RIAB_GATE": "when_busy"}, {"pub_single_rows": 8, "poll_sleep": 16})
        measure("I = C, untimed (no stamps read)", {"RIAB_GATE": "reserved"}, timing=False)
        if rnd == 0:
            measure("J = C + 20 single rows", {"RIAB_GATE": "reserved"}, {"pub_single_rows": 20})
            measure("K = C + poll sleep <= 32", {"RIAB_GATE": "reserved"}, {"poll_sleep": 32})


if __name__ == "__main__":
    main()

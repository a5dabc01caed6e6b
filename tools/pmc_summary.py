#!/usr/bin/env python
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (tools/prof_r02.sh: one counter per pass, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) -> a per-kernel summary (merged by hand into profiles/pmc_traffic.json), which bench.py reads for
`roofline.traffic`.

    python tools/pmc_summary.py profiles/r02_driver_pmc_WRITE_SIZE.csv profiles/r02_driver_pmc_FETCH_SIZE.csv \
        --kernel rate_kernel_gated --units-per-launch 81920 --out profiles/pmc_traffic_cfg2.json
    python tools/pmc_summary.py profiles/r02_default_pmc_WRITE_SIZE.csv profiles/r02_default_pmc_FETCH_SIZE.csv \
        --kernel rate_kernel_wide --units-per-thread 0.015625 --out profiles/r02_pmc_traffic_cfg2_default_run.json

Corrections (the guide's HBM section): FETCH_SIZE on gfx950 tallies 128-B read requests at 64 B -> doubled;
WRITE_SIZE is calibrated in the same pass on riab::fill_kernel dispatches of a known size (bench.py's store-ceiling
leg writes 1 GiB per dispatch)."""
import argparse
import csv
import json


def rows(path, counter):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                out.append((r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("write_csv")
    ap.add_argument("fetch_csv")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--units-per-launch", type=int, default=0)
    ap.add_argument("--units-per-thread", type=float, default=0.0,
                    help="agent-steps per thread of the grid (rate_kernel_wide at n cells: 16 / n; rate_kernel_gated with "
                         "8 cells per wave: 32 / n): launches of different sizes are then summed, bytes / units")
    ap.add_argument("--alg-bytes-kernel", type=int, default=4104)
    ap.add_argument("--alg-bytes-survey", type=int, default=4208)
    ap.add_argument("--source", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    w, f = rows(a.write_csv, "WRITE_SIZE"), rows(a.fetch_csv, "FETCH_SIZE")
    # the timed launches are the largest grids of that kernel (warm-up launches are shorter)
    def per_launch(rs):
        mine = [(g, v) for k, g, v in rs if a.kernel in k]
        gmax = max(g for g, _ in mine)
        vals = [v for g, v in mine if g == gmax]
        return sum(vals) / len(vals), len(vals), gmax
    w_kib, n_w, grid = per_launch(w)
    f_kib, n_f, _ = per_launch(f)
    if a.units_per_thread > 0:   # every launch of the kernel: totals
        mw = [(g, v) for k, g, v in w if a.kernel in k]
        mf = [(g, v) for k, g, v in f if a.kernel in k]
        assert [g for g, _ in mw] == [g for g, _ in mf], "the two passes must have seen the same launches"
        n_w = n_f = len(mw)
        w_kib, f_kib = sum(v for _, v in mw) / n_w, sum(v for _, v in mf) / n_f
        grid = sum(g for g, _ in mw) / n_w
        a.units_per_launch = grid * a.units_per_thread
    fills = [v for k, g, v in w if "fill_kernel" in k]
    calib = (sum(fills) / len(fills)) / (1 << 20) if fills else None   # reported KiB / true KiB (1 GiB = 2^20 KiB)
    w_true = w_kib / calib if calib else w_kib
    total = (w_true + 2.0 * f_kib) * 1024.0
    out = {
        "source": a.source or f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes): {a.write_csv}, {a.fetch_csv}",
        "kernel": a.kernel, "grid_threads": grid, "launches_averaged": [n_w, n_f],
        "units_per_launch": a.units_per_launch,
        "WRITE_SIZE_KiB_per_launch_raw": w_kib, "FETCH_SIZE_KiB_per_launch_raw": f_kib,
        "WRITE_SIZE_calibration": calib,
        "corrections": "FETCH_SIZE doubled (gfx950 counts 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM); WRITE_SIZE "
                       "divided by the factor measured in the same pass on riab::fill_kernel dispatches of exactly 1 GiB",
        "hbm_bytes_per_launch": total, "hbm_bytes_per_unit": total / a.units_per_launch,
        "algorithmic_bytes_per_unit_kernel": a.alg_bytes_kernel, "algorithmic_bytes_per_unit_survey": a.alg_bytes_survey,
        "traffic_over_algorithmic_kernel": total / a.units_per_launch / a.alg_bytes_kernel,
    }
    with open(a.out, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""gpurun_out/<round>_* -> profiles/: the summaries as they are, the large tables cut to what is cited.

* `<round>_driver_kernel_trace.csv`: the rows of the driver's own two kernels (`traj4_kernel`, `rate_kernel_gated`) — the
  secondary runs' 26 000 dispatches are in the per-kernel statistics.
* `<round>_{plan,task}_pmc_*.csv`: the calibration fills and the last 73 dispatches of the step kernel.
* every other `*kernel_trace.csv` stays in gpurun_out/ (scratch).
Usage: python tools/copy_profiles.py r06"""
import csv
import glob
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def rows(path):
    with open(path, newline="") as f:
        r = csv.reader(f)
        head = next(r)
        return head, list(r)


def write(path, head, body):
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(head)
        w.writerows(body)


for p in sorted(glob.glob(os.path.join(src, R + "_*"))):
    name = os.path.basename(p)
    out = os.path.join(dst, name)
    if name.endswith("_kernel_trace.csv"):
        if name != R + "_driver_kernel_trace.csv":
            continue
        head, body = rows(p)
        k = head.index("Kernel_Name")
        write(out, head, [r for r in body if "traj4_kernel" in r[k] or "rate_kernel_gated" in r[k]])
    elif "_pmc_" in name and name.endswith(".csv") and ("_plan_" in name or "_task_" in name):
        head, body = rows(p)
        k = head.index("Kernel_Name")
        fills = [r for r in body if "fill_kernel" in r[k]]
        steps = [r for r in body if "step1_" in r[k]][-73:]
        write(out, head, fills + steps)
    elif name.endswith((".json", ".csv", ".txt")):
        shutil.copyfile(p, out)
    else:
        continue
    print(name, os.path.getsize(out))

#!/usr/bin/env python
"""Build-container check (needs /root/reference): regenerate every golden fixture with
tests/golden/make_golden.py into a temporary directory and compare with the committed tests/golden/*.npz, array by
array, bit for bit.  Every section of the generator seeds its own noise stream, so `--sections a b` checks a subset
just as well.

    python tools/check_golden.py [--sections motion rates ...] [--keep]

Exit status 0: the documented recipe reproduces what is committed."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sections", nargs="*", default=[])
    ap.add_argument("--keep", action="store_true", help="keep the temporary directory")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="riab_golden_")
    cmd = [sys.executable, os.path.join(GOLDEN, "make_golden.py"), *a.sections, "--out", tmp]
    subprocess.run(cmd, check=True, env=dict(os.environ, MPLBACKEND="Agg"), stdout=subprocess.DEVNULL)
    bad = 0
    made = sorted(f for f in os.listdir(tmp) if f.endswith(".npz"))
    for f in made:
        ref_path = os.path.join(GOLDEN, f)
        if not os.path.exists(ref_path):
            print(f"{f}: not committed")
            bad += 1
            continue
        new, old = np.load(os.path.join(tmp, f)), np.load(ref_path)
        if sorted(new.files) != sorted(old.files):
            print(f"{f}: keys differ: {sorted(set(new.files) ^ set(old.files))}")
            bad += 1
            continue
        diff = [k for k in new.files if new[k].shape != old[k].shape or new[k].dtype != old[k].dtype or
                not np.array_equal(new[k], old[k], equal_nan=new[k].dtype.kind == "f")]
        if diff:
            print(f"{f}: {len(diff)} of {len(new.files)} arrays differ: {diff[:6]}")
            bad += 1
        else:
            print(f"{f}: identical ({len(new.files)} arrays)")
    if not a.sections:
        missing = sorted(set(f for f in os.listdir(GOLDEN) if f.endswith(".npz")) - set(made))
        if missing:
            print("committed but not regenerated:", missing)
            bad += 1
    if a.keep:
        print("kept", tmp)
    else:
        shutil.rmtree(tmp, ignore_errors=True)
    print("OK" if not bad else f"{bad} file(s) differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

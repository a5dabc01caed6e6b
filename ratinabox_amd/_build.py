"""Build libriab_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

`python -m ratinabox_amd._build` or `ratinabox_amd._build.build()`.  hipcc
cross-compiles for gfx950 without a GPU; the built library is git-ignored but
travels with the tree to the GPU box."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libriab_hip.so")
SOURCES = ["riab_rates.hip", "riab_agent.hip", "riab_bvc.hip", "riab_ff.hip", "riab_ovc.hip", "riab_plan.hip",
           "riab_task.hip", "riab_task_world.hip", "riab_env.hip", "riab_simulate.hip", "riab_step1.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libriab_hip.so (set HIPCC)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "riab_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libriab_hip.so.  Safe when several ranks
    import the package at once: one process builds under a file lock, into a private directory,
    and the library is moved into place atomically."""
    import fcntl
    import tempfile
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():  # another rank built it while we waited
                return LIB_PATH
            hipcc = _hipcc()
            # (leftovers of builds that were interrupted: they would travel to the GPU box with every push; this process
            # holds the build lock, so nobody is using them)
            for old in os.listdir(LIB_DIR):
                if old.startswith("build_"):
                    shutil.rmtree(os.path.join(LIB_DIR, old), ignore_errors=True)
            work = tempfile.mkdtemp(prefix="build_", dir=LIB_DIR)
            from concurrent.futures import ThreadPoolExecutor

            def compile_one(src):
                obj = os.path.join(work, src.replace(".hip", ".o"))
                cmd = [hipcc, *FLAGS, "-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.run(cmd, check=True)
                return obj

            try:
                # the translation units are independent: compile them side by side
                with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
                    objs = list(pool.map(compile_one, SOURCES))
                tmp_lib = os.path.join(work, "libriab_hip.so")
                cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib, *objs]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.run(cmd, check=True)
                os.replace(tmp_lib, LIB_PATH)
            finally:  # (a failed compile leaves no work directory behind either)
                shutil.rmtree(work, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

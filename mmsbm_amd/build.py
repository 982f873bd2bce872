"""Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "mmsbm_hip.hip")
CSRC = os.path.join(PKG_DIR, "csrc")
DEPS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + [
    os.path.join(os.path.dirname(PKG_DIR), "include", "mmsbm_hip.h")]
LIB = os.path.join(PKG_DIR, "libmmsbm_hip.so")
ARCH = "gfx950"
# MMSBM_HIPCC_FLAGS: extra compiler flags for experiments (e.g. "-mllvm -amdgpu-kernarg-preload-count=16")
EXTRA_FLAGS = os.environ.get("MMSBM_HIPCC_FLAGS", "").split()


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > built for d in DEPS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -shared -> mmsbm_amd/libmmsbm_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB
    tmp = f"{LIB}.{os.getpid()}.tmp"   # renamed into place when complete: nobody ever loads half a library
    cmd = [hipcc_path(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", *EXTRA_FLAGS, "-Wl,-rpath,/opt/rocm/lib", "-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    return LIB


def ensure_library() -> str:
    """Build the library if it is not there (a fresh clone: the .so is not tracked).  Safe to call from
    several ranks at once: one builds under a file lock, the others wait and find it built."""
    if os.path.exists(LIB):
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB):
                build_library(force=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))

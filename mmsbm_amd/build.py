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


def source_id() -> str:
    """First 16 hex digits of the SHA-256 over every file under csrc/ (name, NUL, contents; name order):
    compiled into the library as mmsbm_hip_build_id() and recorded with every profile (bench.py)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))):
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def built_id() -> str:
    """The source id the library on disk was built from ("" if there is none).  Read from the sidecar
    file the build writes next to it -- by content, not by time stamps: a copy of the tree (the GPU box
    gets one) does not keep them."""
    try:
        with open(LIB + ".srcid") as fh:
            rec = fh.read().split()
        return rec[0] if len(rec) == 2 and os.path.exists(LIB) and rec[1] == str(os.path.getsize(LIB)) else ""
    except OSError:
        return ""


def is_stale() -> bool:
    """No library, or one built from other sources (kernels or the C header) than those in the tree."""
    if not os.path.exists(LIB):
        return True
    return built_id() != source_id() or _header_changed()


def _header_changed() -> bool:
    try:
        with open(LIB + ".srcid.h") as fh:
            return fh.read() != _header_digest()
    except OSError:
        return True


def _header_digest() -> str:
    import hashlib
    with open(DEPS[-1], "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -shared -> mmsbm_amd/libmmsbm_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB
    sid = source_id()
    tmp = f"{LIB}.{os.getpid()}.tmp"   # renamed into place when complete: nobody ever loads half a library
    cmd = [hipcc_path(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", f'-DMMSBM_BUILD_ID="{sid}"', *EXTRA_FLAGS,
           "-Wl,-rpath,/opt/rocm/lib", "-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr.strip():
        print(res.stderr)
    os.replace(tmp, LIB)
    with open(LIB + ".srcid", "w") as fh:
        fh.write(f"{sid} {os.path.getsize(LIB)}\n")
    with open(LIB + ".srcid.h", "w") as fh:
        fh.write(_header_digest())
    return LIB


def ensure_library() -> str:
    """Build the library if it is missing OR was built from other sources than the tree holds (a fresh
    clone: the .so is not tracked; an edit under csrc/: bench.py and the tests must never measure the old
    binary).  Safe to call from several ranks at once: one builds under a file lock, the others wait and
    find it built."""
    if not is_stale():
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            build_library()   # (honours is_stale(): whoever got the lock first has built it)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))

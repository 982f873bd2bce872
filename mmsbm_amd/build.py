"""Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).

The library is eight translation units (csrc/prelude.hpp lists them) compiled SIDE BY SIDE -- one hipcc per unit, as
many at a time as the machine has cores -- and linked into one shared object: about 18 s on eight cores, where the
single unit of earlier rounds took a minute.  csrc/unity.hip is all of them as one unit, for the diagnostic builds
(`build_diagnostic`)."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# the translation units, longest compile first (tu_layout.hip carries rocPRIM's sorts)
UNITS = ("tu_layout", "tu_fused", "tu_once", "mmsbm_hip", "tu_seg", "tu_pair", "tu_etap", "tu_mfma")
OBJ_DIR = os.path.join(PKG_DIR, "_build")
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


def _compile_flags(sid: str) -> list:
    return ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function",
            f'-DMMSBM_BUILD_ID="{sid}"', *EXTRA_FLAGS]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950: csrc/<unit>.hip -> _build/<unit>.o side by side, then one link ->
    mmsbm_amd/libmmsbm_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB
    sid = source_id()
    hipcc = hipcc_path()
    tag = f"{os.getpid()}"
    os.makedirs(OBJ_DIR, exist_ok=True)

    def compile_unit(unit: str):
        obj = os.path.join(OBJ_DIR, f"{unit}.{tag}.o")
        cmd = [hipcc, *_compile_flags(sid), "-c", "-o", obj, os.path.join(CSRC, unit + ".hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        return unit, obj, subprocess.run(cmd, capture_output=True, text=True)

    workers = max(1, min(len(UNITS), os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        done = list(pool.map(compile_unit, UNITS))
    objs = [obj for _, obj, _ in done]
    try:
        failed = [(u, r) for u, _, r in done if r.returncode != 0]
        if failed:
            raise RuntimeError("hipcc failed:\n" + "\n".join(f"[{u}]\n{r.stdout}{r.stderr}" for u, r in failed))
        if verbose:
            for u, _, r in done:
                if r.stderr.strip():
                    print(f"[{u}]\n{r.stderr}")
        tmp = f"{LIB}.{tag}.tmp"   # renamed into place when complete: nobody ever loads half a library
        link = [hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared", "-Wl,-rpath,/opt/rocm/lib", "-o", tmp, *objs]
        if verbose:
            print(" ".join(link), flush=True)
        res = subprocess.run(link, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    finally:
        for obj in objs:
            if os.path.exists(obj):
                os.remove(obj)
    os.replace(tmp, LIB)
    with open(LIB + ".srcid", "w") as fh:
        fh.write(f"{sid} {os.path.getsize(LIB)}\n")
    with open(LIB + ".srcid.h", "w") as fh:
        fh.write(_header_digest())
    return LIB


def build_diagnostic(out: str, defines=(), verbose: bool = False) -> str:
    """A diagnostic build of the whole library as ONE translation unit (csrc/unity.hip) with extra -D switches
    (MMSBM_STAMPS: phase stamps; MMSBM_ABLATE: phase ablation through mmsbm_hip_time_stage) -> `out`.  Never the
    product: load it explicitly (mmsbm_amd._lib.load(path))."""
    cmd = [hipcc_path(), *_compile_flags(source_id() + "-diag"), *[f"-D{d}" for d in defines], "-shared",
           "-Wl,-rpath,/opt/rocm/lib", "-o", out, os.path.join(CSRC, "unity.hip")]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return out


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

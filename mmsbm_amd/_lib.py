"""ctypes binding of include/mmsbm_hip.h.  No fallback: if the shared library is missing
the import of anything that needs it fails loudly."""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# (MMSBM_HIP_LIBRARY: a diagnostic build of the same ABI, e.g. the phase-stamp build of scripts/)
LIB_PATH = os.environ.get("MMSBM_HIP_LIBRARY") or os.path.join(PKG_DIR, "libmmsbm_hip.so")

OK, E_INVALID, E_HIP, E_NODEVICE, E_UNSUPPORTED, E_TOOLARGE, E_INTERNAL = range(7)

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
c_f32p = C.POINTER(C.c_float)
c_intp = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/mmsbm_hip.h declares
SIGNATURES = {
    "mmsbm_hip_abi_version": (C.c_int, []),
    "mmsbm_hip_build_id": (C.c_char_p, []),
    "mmsbm_hip_last_error": (C.c_char_p, []),
    "mmsbm_hip_device_count": (C.c_int, [c_intp]),
    "mmsbm_hip_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, c_intp, c_i64p]),
    "mmsbm_hip_device_pci": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "mmsbm_hip_device_mem": (C.c_int, [C.c_int, c_i64p, c_i64p]),
    "mmsbm_hip_create": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, c_i32p, c_i32p, c_i32p, C.c_int,
                                   C.POINTER(C.c_void_p)]),
    "mmsbm_hip_destroy": (C.c_int, [C.c_void_p]),
    "mmsbm_hip_dims": (C.c_int, [C.c_void_p, c_i64p]),
    "mmsbm_hip_degrees": (C.c_int, [C.c_void_p, c_i64p, c_i64p]),
    "mmsbm_hip_set_params": (C.c_int, [C.c_void_p, c_f64p, c_f64p, c_f64p]),
    "mmsbm_hip_get_params": (C.c_int, [C.c_void_p, c_f64p, c_f64p, c_f64p]),
    "mmsbm_hip_init_params": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), c_f64p]),
    "mmsbm_hip_pcg64_doubles": (C.c_int, [C.POINTER(C.c_uint64), C.c_uint64, C.c_int64, c_f64p]),
    "mmsbm_hip_set_slots": (C.c_int, [C.c_void_p, C.c_int]),
    "mmsbm_hip_select_slot": (C.c_int, [C.c_void_p, C.c_int]),
    "mmsbm_hip_slots": (C.c_int, [C.c_void_p, c_intp, c_intp, c_i64p]),
    "mmsbm_hip_em_iterate": (C.c_int, [C.c_void_p, C.c_int]),
    "mmsbm_hip_synchronize": (C.c_int, [C.c_void_p]),
    "mmsbm_hip_update_coefficients": (C.c_int, [C.c_void_p, c_f64p, c_f64p, c_f64p]),
    "mmsbm_hip_likelihood": (C.c_int, [C.c_void_p, c_f64p]),
    "mmsbm_hip_result": (C.c_int, [C.c_void_p, c_f64p, c_f64p, c_f64p, c_f64p]),
    "mmsbm_hip_compute_omegas": (C.c_int, [C.c_void_p, c_f64p, C.c_int64]),
    "mmsbm_hip_prod_dist": (C.c_int, [C.c_void_p, C.c_int64, c_i32p, c_i32p, c_f64p]),
    "mmsbm_hip_predict_begin": (C.c_int, [C.c_void_p, C.c_int64, c_i32p, c_i32p, c_i32p, c_f64p]),
    "mmsbm_hip_predict_add": (C.c_int, [C.c_void_p, c_f64p]),
    "mmsbm_hip_predict_finish": (C.c_int, [C.c_void_p, c_f64p, c_f64p]),
    "mmsbm_hip_time_iterations": (C.c_int, [C.c_void_p, C.c_int, c_f32p]),
    "mmsbm_hip_kernel_count": (C.c_int, []),
    "mmsbm_hip_kernel_name": (C.c_char_p, [C.c_int]),
    "mmsbm_hip_profile_iterations": (C.c_int, [C.c_void_p, C.c_int, c_f32p, c_intp]),
    "mmsbm_hip_kernel_bytes": (C.c_int, [C.c_void_p, C.c_int, c_i64p, c_i64p]),
    "mmsbm_hip_time_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_f32p]),
    "mmsbm_hip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "mmsbm_hip_get_option": (C.c_int, [C.c_void_p, C.c_char_p, c_f64p]),
    "mmsbm_hip_set_graph_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "mmsbm_hip_layout_build": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p,
                                         c_i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    "mmsbm_hip_layout_array": (C.c_int, [C.c_void_p, C.c_int, c_i32p, C.c_int64, c_i64p]),
    "mmsbm_hip_layout_free": (C.c_int, [C.c_void_p]),
    "mmsbm_hip_layout_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_int, c_i32p, C.c_int64, c_i64p]),
    "mmsbm_hip_selftest_throw": (C.c_int, [C.c_int]),
}

_lib = None


class HipLibraryError(RuntimeError):
    """A C-ABI call returned a non-zero code."""

    def __init__(self, func, code, message):
        super().__init__(f"{func} failed (code {code}): {message}")
        self.func, self.code, self.message = func, code, message


def load():
    """Load libmmsbm_hip.so (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m mmsbm_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    except OSError as exc:
        raise ImportError(f"could not load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(func_name, code):
    if code != OK:
        msg = load().mmsbm_hip_last_error()
        raise HipLibraryError(func_name, code, msg.decode("utf-8", "replace") if msg else "")


def call(func_name, *args):
    check(func_name, getattr(load(), func_name)(*args))


def build_id() -> str:
    """Source identity the LOADED binary was compiled from (see mmsbm_hip_build_id in the header)."""
    return load().mmsbm_hip_build_id().decode()


def device_identity(device: int) -> dict:
    """{"name", "compute_units", "memory_bytes", "pci_bus_id"} of a HIP device, from the library's own runtime."""
    name, bus = C.create_string_buffer(256), C.create_string_buffer(64)
    cus, mem = C.c_int(0), C.c_int64(0)
    call("mmsbm_hip_device_info", int(device), name, 256, C.byref(cus), C.byref(mem))
    call("mmsbm_hip_device_pci", int(device), bus, 64)
    return {"name": name.value.decode(), "compute_units": int(cus.value), "memory_bytes": int(mem.value),
            "pci_bus_id": bus.value.decode()}


def device_count() -> int:
    """Number of HIP devices; 0 when there is none (never raises for "no GPU")."""
    n = C.c_int(0)
    code = load().mmsbm_hip_device_count(C.byref(n))
    return int(n.value) if code == OK else 0


def device_mem(device: int):
    """(free, total) bytes of a HIP device as its driver reports them now."""
    free, total = C.c_int64(0), C.c_int64(0)
    call("mmsbm_hip_device_mem", int(device), C.byref(free), C.byref(total))
    return int(free.value), int(total.value)


def worker_device(n_devices: int, env=None, identity=None) -> int:
    """The GPU a level-1 process (``kernels_hip``) works on.

    The reference runs every restart in its own spawned worker (``Pool(processes=sampling)``, src/mmsbm.py:182-185)
    and gives a backend no say in where (its own cupy backend puts every worker on GPU 0: README.md:186).  Rule:

    * ``MMSBM_HIP_DEVICE`` if set (an index; must be below the device count);
    * else, inside a ``multiprocessing`` child, (worker number - 1) mod device count -- Pool workers are numbered
      1, 2, ... in the order the parent started them, so ``sampling`` workers land on ``sampling`` GPUs round robin;
    * else (the main process) device 0.
    """
    env = os.environ if env is None else env
    n = max(int(n_devices), 1)
    forced = env.get("MMSBM_HIP_DEVICE", "").strip()
    if forced:
        try:
            d = int(forced)
        except ValueError:
            raise ValueError(f"MMSBM_HIP_DEVICE={forced!r}: an integer device index expected") from None
        if not 0 <= d < n:
            raise ValueError(f"MMSBM_HIP_DEVICE={d}: this machine shows {n} HIP device(s)")
        return d
    if identity is None:
        import multiprocessing
        identity = getattr(multiprocessing.current_process(), "_identity", ())
    return (int(identity[-1]) - 1) % n if identity else 0


def loaded_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process (two == trouble)."""
    out = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    out.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return sorted(out)

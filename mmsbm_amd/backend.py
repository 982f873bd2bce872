"""Backend selector with the semantics of the reference's src/backend.py:4-28.

``load_backend(name)`` returns ``(compute_omegas, update_coefficients, prod_dist, name)``.
This package ships exactly one backend, ``hip``; ``auto`` resolves to it.  Unknown or
unloadable names raise ``ImportError("Could not load any backend. Last error: ...")`` like
the reference does.  There is deliberately no numpy fallback here.
"""
from importlib import import_module


def load_backend(name: str = "auto"):
    order = ["hip"] if name == "auto" else [name]
    last_error = None
    for backend in order:
        try:
            mod = import_module(f"{__package__}.kernels_{backend}")
            return mod.compute_omegas, mod.update_coefficients, mod.prod_dist, backend
        except ModuleNotFoundError as e:
            last_error = e
        except ImportError as e:
            last_error = e
    raise ImportError(f"Could not load any backend. Last error: {last_error}")

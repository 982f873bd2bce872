"""mmsbm_amd -- MI355X-native EM core for the Mixed-Membership Stochastic Block Model.

Only the hot path of eudald-seeslab/mmsbm (the omega / theta / eta / p update) lives here:
hand-written HIP kernels for gfx950 behind a C ABI (``include/mmsbm_hip.h``), a ctypes
handle (``HipEM``), the reference's three-function backend contract (``kernels_hip``) and a
host class with the reference's ``MMSBM`` surface for ``backend='hip'``.
"""
from .core import HipEM, build_layout  # noqa: F401
from .mmsbm import MMSBM  # noqa: F401
from .backend import load_backend  # noqa: F401

__version__ = "0.1.0"

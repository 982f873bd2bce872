"""Put this directory on sys.path (or PYTHONPATH) and the reference's
``load_backend('hip')`` (src/backend.py:21: ``import_module(f"kernels_{backend}")``)
resolves to the HIP backend with no edit to the reference."""
from mmsbm_amd.kernels_hip import *  # noqa: F401,F403
from mmsbm_amd.kernels_hip import __all__  # noqa: F401

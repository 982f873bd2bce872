"""TEST INFRASTRUCTURE (build container only).  With this directory on PYTHONPATH every Python process of the
environment -- the spawned Pool workers of the reference's ``MMSBM.fit`` (src/mmsbm.py:182-185) included -- starts
with ``tests.fake_device.FakeHipEM`` (the oracle behind the device interface) standing in for the HIP library's
handle, so that the REAL reference caller can drive ``mmsbm_amd/plugin/kernels_hip.py`` on a machine without a GPU
(``tests/test_reference_caller_cpu.py``).  Nothing under ``mmsbm_amd/`` knows about this file."""
import os
import sys

_ROOT = os.environ.get("MMSBM_FAKE_SITE_ROOT")
if _ROOT:
    sys.path.insert(0, _ROOT)
    import mmsbm_amd._lib as _lib
    import mmsbm_amd.core as _core
    from tests.fake_device import FakeHipEM

    _lib.device_count = lambda: 1      # kernels_hip refuses to import without a device
    _core.HipEM = FakeHipEM            # `from .core import HipEM` in kernels_hip now gets the stand-in
    _log = os.environ.get("MMSBM_FAKE_SITE_LOG")
    if _log:
        with open(_log, "a") as fh:
            fh.write(f"{os.getpid()}\n")

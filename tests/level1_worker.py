"""Worker side of tests/test_gpu_level1_spawn.py: what ONE restart does in the reference's spawned Pool worker
(src/mmsbm.py:182-185 -> run_one_sampling, :243-256) -- the three backend callables resolved the way
src/backend.py:21 resolves them, plus the reference's numpy normalisations -- restated here (nothing of the reference
is imported: it does not exist on the GPU box)."""
import multiprocessing
import os
import sys

import numpy as np

EPS = np.finfo(float).eps


def run_restart(args):
    plugin_dir, train, theta, eta, pr, d_u, d_i, iterations = args
    if plugin_dir not in sys.path:
        sys.path.insert(0, plugin_dir)
    import importlib
    kernels_hip = importlib.import_module("kernels_" + "hip")   # src/backend.py:21
    import mmsbm_amd.kernels_hip as impl
    from mmsbm_amd import _lib
    for _ in range(iterations):                                 # src/mmsbm.py:243-250
        n_theta, n_eta, n_pr = kernels_hip.update_coefficients(train, theta, eta, pr)
        theta = n_theta / d_u[:, None]                          # normalize_with_d
        eta = n_eta / d_i[:, None]
        s = n_pr.sum(axis=2, keepdims=True)                     # normalize_with_self (zero rows divide by 1)
        s[s == 0] = 1
        pr = n_pr / s
    om = kernels_hip.compute_omegas(train, theta, eta, pr)     # src/expectation_maximization.py:157-167
    w = np.maximum(om, EPS)
    tot = np.maximum(om.sum(axis=(1, 2)), EPS)
    lik = float(np.sum(w * np.log(w) - w * np.log(tot)[:, None, None]))
    dist = kernels_hip.prod_dist(train, theta, eta, pr)
    dev = impl.device()
    free_before_close, _ = _lib.device_mem(dev)
    impl.clear_cache()
    return {"pid": os.getpid(), "identity": tuple(multiprocessing.current_process()._identity), "device": dev,
            "n_devices": _lib.device_count(), "theta": theta, "eta": eta, "pr": pr, "likelihood": lik,
            "dist_row_sums": dist.sum(axis=1), "module_file": os.path.abspath(kernels_hip.__file__),
            "free_before_close": free_before_close}

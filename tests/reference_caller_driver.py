"""Runs the REAL reference class with backend='hip' (build container only; started as a child process by
tests/test_reference_caller_cpu.py with mmsbm_amd/plugin, the reference's src/ and tests/fake_site on PYTHONPATH).
Nothing of the reference is copied: it is imported from where it lies."""
import os
import sys

import numpy as np
import pandas as pd


def mock_frame(seed, n=100):
    # the data recipe of the reference's tests/test_mmsbm.py:12-22 (as in tests/golden/make_golden.py)
    rng = np.random.default_rng(seed)
    return pd.DataFrame({
        "users": [f"user{rng.choice(list(range(5)))}" for _ in range(n)],
        "items": [f"item{rng.choice(list(range(10)))}" for _ in range(n)],
        "ratings": [rng.choice(list(range(1, 6))) for _ in range(n)],
    })


def main(out_path, sampling):
    from mmsbm import MMSBM  # the reference's own class (src/mmsbm.py)
    import kernels_hip       # resolved the way src/backend.py:21 resolves it

    mm = MMSBM(2, 2, iterations=10, sampling=sampling, seed=1, backend="hip")
    mm.fit(mock_frame(1), silent=True)          # spawn Pool, a fresh import of kernels_hip in every worker
    backend_name = mm.em._backend
    owners = sorted({f.__module__ for f in (mm.em._compute_omegas, mm.em._update_coeffs, mm.em._prod_dist)})
    pm = mm.predict(mock_frame(2))
    sc = mm.score(silent=True)
    out = {"backend": np.array(backend_name), "plugin_file": np.array(os.path.abspath(kernels_hip.__file__)),
           "prediction_matrix": pm, "stats_keys": np.array(list(sc["stats"].keys())),
           "stats_vals": np.array([float(np.sum(v)) for v in sc["stats"].values()]),
           "likelihoods": np.array([float(r["likelihood"]) for r in mm.results]), "pid": np.array(os.getpid()),
           "kernel_modules": np.array(owners)}
    for s, r in enumerate(mm.results):
        out[f"theta_{s}"], out[f"eta_{s}"], out[f"pr_{s}"] = r["theta"], r["eta"], r["pr"]
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))

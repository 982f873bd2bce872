"""Level 1 the way the reference really runs it, on the GPU: ``MMSBM.fit`` with ``sampling`` > 1 starts
``Pool(processes=sampling)`` from a spawn context and every worker imports ``kernels_<name>`` afresh
(/root/reference/src/mmsbm.py:182-185, src/backend.py:16-22) -- several processes, a HIP context each.  Here three
spawned workers import ``kernels_hip`` through mmsbm_amd/plugin and run the ten iterations of the reference's own
sampling = 3 test case (fixture G2, made by RUNNING the reference: tests/golden/make_golden.py)."""
import multiprocessing
import os

import numpy as np
import pytest

from conftest import ROOT, assert_elementwise, load_golden, rel_err
from oracle import mmsbm_oracle as orc

pytestmark = pytest.mark.gpu


def test_three_spawned_workers_reproduce_the_reference_sampling_run():
    from mmsbm_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("-m gpu tests need a GPU: no HIP device visible (no CPU fallback exists)")
    import level1_worker
    g = load_golden("g2_c1_sampling3")
    train = g["train"]
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    d_u, d_i = orc.degrees(train, n_u, n_i)
    tasks = []
    for seed in orc.child_seeds(1, 3):      # MMSBM(2, 2, iterations=10, sampling=3, seed=1): src/mmsbm.py:82-85,224-233
        theta, eta, pr = orc.init_params(seed, n_u, n_i, n_r, 2, 2, d_u, d_i)
        tasks.append((os.path.join(ROOT, "mmsbm_amd", "plugin"), train, theta, eta, pr,
                      d_u.astype(np.float64), d_i.astype(np.float64), 10))
    n_dev = _lib.device_count()
    free0 = [_lib.device_mem(d)[0] for d in range(n_dev)]
    # maxtasksperchild=1: a worker leaves after its restart, so the three restarts run in three processes whatever
    # the timing (the reference's Pool has as many workers as restarts and hands each one restart)
    with multiprocessing.get_context("spawn").Pool(processes=3, maxtasksperchild=1) as pool:
        res = pool.map(level1_worker.run_restart, tasks, chunksize=1)
        pool.close()
        pool.join()
    assert len({r["pid"] for r in res}) == 3 and os.getpid() not in {r["pid"] for r in res}
    for s, r in enumerate(res):
        assert r["module_file"] == os.path.join(ROOT, "mmsbm_amd", "plugin", "kernels_hip.py")
        # the device rule (mmsbm_amd/_lib.py: worker_device): worker number - 1, modulo the device count
        assert r["identity"] and r["device"] == (r["identity"][-1] - 1) % r["n_devices"], (r["identity"], r["device"])
        for nm in ("theta", "eta", "pr"):
            assert rel_err(r[nm], g[f"{nm}_{s}"]) < 1e-9, (s, nm)
            assert_elementwise(r[nm], g[f"{nm}_{s}"], f"restart {s} {nm}")
        assert abs(r["likelihood"] - g["likelihoods"][s]) <= 1e-9 * abs(g["likelihoods"][s])
        assert np.allclose(r["dist_row_sums"], 1.0, atol=1e-12)
    if n_dev > 1:   # (one GPU: all on it; more: the workers are spread)
        assert len({r["device"] for r in res}) == min(3, n_dev)
    # nothing left on the devices once the workers are gone (64 MiB: allocator granularity, other processes)
    free1 = [_lib.device_mem(d)[0] for d in range(n_dev)]
    for d in range(n_dev):
        assert free1[d] >= free0[d] - (64 << 20), (d, free0[d], free1[d])


def test_device_override_and_bad_values():
    from mmsbm_amd import _lib
    n = _lib.device_count()
    assert _lib.worker_device(n, env={"MMSBM_HIP_DEVICE": str(n - 1)}) == n - 1
    with pytest.raises(ValueError):
        _lib.worker_device(n, env={"MMSBM_HIP_DEVICE": str(n)})

"""The reference's OWN caller driving the plugin -- SURVEY 8(b) level 1, "zero edits to the reference".

Build container only (skipped where /root/reference is absent, i.e. on the GPU box): the REAL ``MMSBM(2, 2,
iterations=10, seed=1, backend='hip')`` of /root/reference/src/mmsbm.py is imported (never copied) in a child process
whose PYTHONPATH holds ``mmsbm_amd/plugin`` -- so that ``load_backend('hip')`` -> ``import_module("kernels_hip")``
(src/backend.py:16-22) finds this repo's module -- and ``tests/fake_site``, which puts the oracle-backed stand-in
behind ``mmsbm_amd.core.HipEM`` in EVERY process, the spawned Pool workers of ``fit`` included (src/mmsbm.py:182-185:
a fresh import of ``kernels_hip`` per worker; results pickled back).  ``fit`` / ``predict`` / ``score`` must give the
numbers of fixture G1 / G2, which the same class produced with backend='numpy'."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MMSBM_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "mmsbm.py")),
                                reason="the reference checkout only exists in the build container")


def run_reference(tmp_path, sampling):
    out = tmp_path / "out.npz"
    log = tmp_path / "pids.log"
    env = dict(os.environ)
    env.update(PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "fake_site"), os.path.join(ROOT, "mmsbm_amd", "plugin"),
                                           os.path.join(REF, "src"), ROOT]),
               PYTHONDONTWRITEBYTECODE="1", MMSBM_FAKE_SITE_ROOT=ROOT, MMSBM_FAKE_SITE_LOG=str(log))
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "reference_caller_driver.py"), str(out), str(sampling)],
                          cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout + done.stderr
    pids = [int(x) for x in log.read_text().split()]
    return np.load(out), pids


def test_the_references_mmsbm_runs_on_the_plugin_and_gives_g1(tmp_path):
    got, pids = run_reference(tmp_path, 1)
    g = np.load(os.path.join(GOLD, "g1_c1_mock.npz"))
    assert str(got["backend"]) == "hip"                                     # em._backend, src/backend.py:22
    assert str(got["plugin_file"]) == os.path.join(ROOT, "mmsbm_amd", "plugin", "kernels_hip.py")
    assert list(got["kernel_modules"]) == ["mmsbm_amd.kernels_hip"]        # the three callables are this repo's
    # a fresh interpreter per worker: the stand-in was installed in the parent AND in another process
    assert int(got["pid"]) in pids and len(set(pids)) >= 2
    for name in ("theta", "eta", "pr"):
        assert np.array_equal(got[name + "_0"], g["t_" + name]), name
    assert float(got["likelihoods"][0]) == float(g["t_likelihood"])
    assert np.array_equal(got["prediction_matrix"], g["t_prediction_matrix"])
    assert list(got["stats_keys"]) == list(g["t_stats_keys"])
    assert np.array_equal(got["stats_vals"], g["t_stats_vals"])


def test_three_restarts_in_three_spawned_workers_give_g2(tmp_path):
    got, pids = run_reference(tmp_path, 3)
    g = np.load(os.path.join(GOLD, "g2_c1_sampling3.npz"))
    assert len(set(pids)) >= 4                                              # parent + Pool(processes=3)
    for s in range(3):
        for name in ("theta", "eta", "pr"):
            assert np.array_equal(got[f"{name}_{s}"], g[f"{name}_{s}"]), (name, s)
    assert np.array_equal(got["likelihoods"], g["likelihoods"])
    assert np.array_equal(got["prediction_matrix"], g["prediction_matrix"])

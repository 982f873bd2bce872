"""Shared pytest plumbing: the `gpu` marker, repo-root imports, golden-fixture loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Launch log (mmsbm_amd/csrc/launch.hpp): on for every test run, so that the -m gpu suite leaves a record of WHICH kernel
# instantiations it launched, how often and under which test -- scripts/kernel_coverage.py diffs it against the compiled
# set (profiles/r6_kernel_coverage.csv).  The library reads the variable when it is loaded, so it is set here, before any
# test module imports it; child processes the tests start inherit it and append to the same file.  Host code only;
# without a GPU nothing is launched and no file appears.  MMSBM_HIP_LAUNCH_LOG= (empty) switches it off.
os.environ.setdefault("MMSBM_HIP_LAUNCH_LOG", os.path.join(ROOT, "gpurun_out", "launch_log.tsv"))
if os.environ["MMSBM_HIP_LAUNCH_LOG"]:
    os.makedirs(os.path.dirname(os.path.abspath(os.environ["MMSBM_HIP_LAUNCH_LOG"])), exist_ok=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")
    # a fresh clone has no libmmsbm_hip.so yet (it is not tracked): build it once (hipcc cross-compiles
    # without a GPU); a machine without hipcc just lets the tests that need the library say so
    try:
        from mmsbm_amd.build import ensure_library
        ensure_library()
    except Exception as exc:  # noqa: BLE001
        sys.stderr.write(f"[conftest] library not built: {exc}\n")


@pytest.fixture(autouse=True)
def _launch_tag(request):
    """The running test's id, recorded with the first launch of every kernel instantiation (launch log)."""
    os.environ["MMSBM_HIP_LAUNCH_TAG"] = request.node.nodeid
    yield
    os.environ["MMSBM_HIP_LAUNCH_TAG"] = "-"


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(got, want):
    """max |got - want| / max |want|  (the survey's B.4/B.5 error measure)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    denom = np.max(np.abs(want)) if want.size else 1.0
    return float(np.max(np.abs(got - want)) / (denom if denom > 0 else 1.0)) if want.size else 0.0


ELEMENT_FLOOR = 1e-290   # below this an entry is within a few binades of the subnormals: relative error means nothing
ELEMENT_RTOL = 1e-9      # north_star's bar is 1e-5 relative; every multiplicative update should hold far inside it


def elem_rel_err(got, want, floor=ELEMENT_FLOOR):
    """max over the entries with |want| > floor of |got - want| / |want| -- ELEMENT-WISE relative error,
    next to rel_err's max-norm (in which an entry of 1e-20 that is wrong by a factor of ten is invisible).
    Entries at or below the floor must agree to within the floor itself."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if not want.size:
        return 0.0
    big = np.abs(want) > floor
    small_ok = np.all(np.abs(got[~big] - want[~big]) <= floor)
    if not small_ok:
        return float("inf")
    if not big.any():
        return 0.0
    return float(np.max(np.abs(got[big] - want[big]) / np.abs(want[big])))


def assert_elementwise(got, want, what="", rtol=ELEMENT_RTOL):
    err = elem_rel_err(got, want)
    assert err <= rtol, f"{what}: element-wise relative error {err:.3e} > {rtol:.1e}"

#!/usr/bin/env python3
"""Calibrate the CPU baseline: time the REAL reference numpy backend against this repo's oracle
("port") on the same arrays, in the BUILD CONTAINER (the reference never travels to the GPU box).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/scripts/calibrate_cpu_baseline.py

One iteration = the body of src/mmsbm.py:244-250: kernels_numpy.update_coefficients +
normalize_with_d x 2 + normalize_with_self, on pre-encoded int64 triples, one core
(BASELINE.md section 4).  Workloads: C2 in full, C3 on its first 300,000 rows (the dense
N x K x L dataflow is linear in N; the full 1M rows need 7.3 GB and ~12 s per iteration here).
The two implementations are timed INTERLEAVED (reference, oracle, reference, ...) so that drift of
the shared host affects both alike.  Writes oracle/calibration.json, which bench.py attaches to
`cpu_baseline` as `port_over_reference` (= oracle seconds / reference seconds: > 1 means the port
is slower than the reference, so the GPU/CPU ratio bench.py prints is that much too flattering).
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MMSBM_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import kernels_numpy as ref_k  # noqa: E402  (the real reference)
from oracle import mmsbm_oracle as orc  # noqa: E402


def make_ref_em(n_rows, d_u, d_i, k, l, n_r):
    """The reference's EM driver on its numpy backend, given what MMSBM._prepare_objects gives it:
    the degrees as (U,K) / (I,L) integer arrays of repeated values (src/mmsbm.py:106-111); the
    index lists it also takes are dead (never read)."""
    from expectation_maximization import ExpectationMaximization
    dims = {"n_samples": n_rows, "n_user_groups": k, "n_item_groups": l, "n_ratings": n_r}
    norm = {"user": np.repeat(d_u[:, None], k, axis=1), "item": np.repeat(d_i[:, None], l, axis=1)}
    return ExpectationMaximization(dims, [], [], [], norm, backend="numpy")


def time_pair(cfg, rows, iters):
    n, u, i, r, k, l = cfg
    train = orc.synthetic_triples(n, u, i, r, seed=0)[:rows]
    n_u, n_i, n_r = (int(orc.synthetic_triples(n, u, i, r, seed=0)[:, j].max()) + 1 for j in range(3))
    d_u, d_i = orc.degrees(train, n_u, n_i)
    theta0, eta0, pr0 = orc.init_params(orc.child_seeds(0, 1)[0], n_u, n_i, n_r, k, l, d_u, d_i)
    em = make_ref_em(len(train), d_u, d_i, k, l, n_r)
    assert em._update_coeffs is ref_k.update_coefficients

    def ref_step(t, e, p):  # the loop body of src/mmsbm.py:244-250, the reference's own objects
        n_t, n_e, n_p = em.update_coefficients(train, t, e, p)
        return em.normalize_with_d(n_t, "user"), em.normalize_with_d(n_e, "item"), em.normalize_with_self(n_p)

    def port_step(t, e, p):
        return orc.em_step(train, t, e, p, d_u, d_i)

    state = {"ref": (theta0, eta0, pr0), "port": (theta0, eta0, pr0)}
    steps = {"ref": ref_step, "port": port_step}
    for nm in ("ref", "port"):  # warm-up (page faults)
        state[nm] = steps[nm](*state[nm])
    times = {"ref": [], "port": []}
    for _ in range(iters):
        for nm in ("ref", "port"):
            t0 = time.perf_counter()
            state[nm] = steps[nm](*state[nm])
            times[nm].append(time.perf_counter() - t0)
    for a, b in zip(state["ref"], state["port"]):  # same arithmetic: the two runs stay bit-identical
        assert np.array_equal(a, b), "oracle and reference diverged"
    med = {nm: float(np.median(v)) for nm, v in times.items()}
    return med, n / rows


def main():
    from mmsbm_amd.synthetic import CONFIGS
    out = {"host": os.uname().nodename, "cpus": os.cpu_count(), "numpy": np.__version__, "runs": {}}
    ratios = []
    for name, rows, iters in (("c2", 100_000, 10), ("c3", 300_000, 3)):
        med, scale = time_pair(CONFIGS[name], rows, iters)
        ratio = med["port"] / med["ref"]
        ratios.append(ratio)
        out["runs"][name] = {"rows": rows, "iterations": iters, "reference_s_per_iteration": med["ref"],
                             "port_s_per_iteration": med["port"], "port_over_reference": ratio,
                             "reference_full_size_it_per_s": 1.0 / (med["ref"] * scale),
                             "port_full_size_it_per_s": 1.0 / (med["port"] * scale)}
        print(f"{name}: rows {rows}  reference {med['ref']:.3f} s/it  port {med['port']:.3f} s/it  "
              f"port/reference {ratio:.3f}  (full size: reference {1 / (med['ref'] * scale):.3f} it/s, "
              f"port {1 / (med['port'] * scale):.3f} it/s)", flush=True)
    out["port_over_reference"] = out["runs"]["c3"]["port_over_reference"]
    out["note"] = ("oracle (port) seconds / real reference seconds per EM iteration, both on one core of the build "
                   "container, interleaved, same arrays: C3's first 300,000 rows "
                   f"{out['runs']['c3']['port_over_reference']:.3f}, C2 in full "
                   f"{out['runs']['c2']['port_over_reference']:.3f} (scripts/calibrate_cpu_baseline.py)")
    with open(os.path.join(ROOT, "oracle", "calibration.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

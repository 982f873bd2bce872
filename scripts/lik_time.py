"""Likelihood kernels: value and time per evaluation for the log-per-element form and the
log-table form with G lanes per triple."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples

for tag in sys.argv[1:] or ["c2", "c3"]:
    n, u, i, r, k, l = CONFIGS[tag]
    train = synthetic_triples(n, u, i, r, 0)
    mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
    ctx = mm._ctx(0); d_u, d_i = ctx.degrees()
    ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i)); done = 0
    def run(label):
        ctx.likelihood()
        reps = 5 if n > 2_000_000 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            v = ctx.likelihood()
        dt = (time.perf_counter() - t0) / reps
        print(f"{tag} {label:14s} {dt * 1e3:9.3f} ms   {v!r}", flush=True)
        return v
    for its in (30, 100, 400):   # early (everything live) ... late (concentrated memberships: dead and mixed rows)
        ctx.iterate(its - done); done = its
        t, e, p = ctx.get_params()
        print(f"{tag} after {its} iterations: theta entries < eps {np.mean(t < 2.2e-16):.3f}, eta {np.mean(e < 2.2e-16):.3f}, "
              f"p {np.mean(p < 2.2e-16):.3f}", flush=True)
        if n <= 2_000_000 or its == 30:
            ctx.set_option("lik_fast", 0); ref = run("log/element")
        ctx.set_option("lik_fast", 1); ctx.set_option("lik_g", 0)
        v = run("tables"); ref = v if n > 2_000_000 and its != 30 else ref
        print(f"      relative difference {abs(v - ref) / abs(ref):.2e}")
        ctx.set_option("lik_fast", 2)
        v = run("wave per pair")
        print(f"      relative difference {abs(v - ref) / abs(ref):.2e}")
    mm._release()

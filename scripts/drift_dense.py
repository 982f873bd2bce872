"""Drift against the oracle on DENSE data (every segment cut into work items; with
MMSBM_HIP_RANGES=p,u also the XCD-local lists): drift_dense.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM
from oracle import mmsbm_oracle as orc
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, u, i, r, k, l = 3_000_000, 5000, 3000, 5, 10, 10
rng = np.random.default_rng(0)
w = rng.lognormal(0, 1, u); w /= w.sum()
train = np.stack([rng.choice(u, n, p=w), rng.integers(0, i, n), rng.integers(0, r, n)], axis=1).astype(np.int64)
for c in range(3):
    train[:, c] = np.unique(train[:, c], return_inverse=True)[1]
nu, ni, nr = (int(train[:, j].max()) + 1 for j in range(3))
d_u, d_i = orc.degrees(train, nu, ni)
theta, eta, pr = orc.init_params(np.random.SeedSequence(3), nu, ni, nr, k, l, d_u, d_i)
em = HipEM(train, k, l, nu, ni, nr)
em.set_params(theta, eta, pr)
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
done, t0 = 0, time.time()
for stop in (1, 5, 15, 30, 60):
    if stop > iters: break
    em.iterate(stop - done)
    for _ in range(stop - done):
        theta, eta, pr = orc.em_step(train, theta, eta, pr, d_u, d_i)
    done = stop
    t, e, p = em.get_params()
    lik, lik_o = em.likelihood(), orc.compute_likelihood(train, theta, eta, pr)
    print(f"dense it {stop:3d}: theta {rel(t, theta):.2e} eta {rel(e, eta):.2e} p {rel(p, pr):.2e} "
          f"likelihood rel {abs(lik - lik_o) / abs(lik_o):.2e} [{time.time() - t0:.0f}s]", flush=True)

"""Two launches against four on data with uneven degrees (where the two-launch form is available at all)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM

def lognormal(n, u, i, r, sigma, seed=0):
    rng = np.random.default_rng(seed)
    pu, pi = rng.lognormal(0, sigma, u), rng.lognormal(0, sigma, i)
    return np.stack([rng.choice(u, n, p=pu / pu.sum()), rng.choice(i, n, p=pi / pi.sum()), rng.integers(0, r, n)], axis=1).astype(np.int64)

for n, u, i, r, k, l in ((100_000, 10_000, 5_000, 5, 10, 10), (300_000, 30_000, 10_000, 5, 10, 10), (100_000, 10_000, 5_000, 5, 20, 20),
                         (100_000, 943, 1_682, 5, 10, 10)):
    for sigma in (0.3, 0.6, 1.0, 1.5):
        data = lognormal(n, u, i, r, sigma)
        with HipEM(data, k, l) as em:
            em.init_params(1)
            avail, choice = em.get_option("fused"), int(em.get_option("launches"))
            out = {}
            for fused in ((0, 1) if avail else (0,)):
                em.set_option("fused", fused)
                em.iterate(20)
                out[fused] = min(em.time_iterations(300) for _ in range(3)) / 300 * 1e3
            print(f"{n:>8} x {u} x {i} K={k} sigma {sigma}: four {out[0]:7.2f} us  two " + (f"{out[1]:7.2f} us" if 1 in out else "   n/a   ") +
                  f"  (library: {choice}; work items {int(em.get_option('items_pairs'))} / {int(em.get_option('items_users'))})", flush=True)

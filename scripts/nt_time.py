"""Option nt_out (non-temporal rows between launches) on and off, microseconds per EM iteration, over row widths and
degree distributions.  usage: nt_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM
from mmsbm_amd.synthetic import synthetic_triples

def skewed(n, u, i, r, kind, seed=0, sigma=None):
    rng = np.random.default_rng(seed)
    if sigma is not None:
        pu, pi = rng.lognormal(0, sigma, u), rng.lognormal(0, sigma, i)
        uc, ic = rng.choice(u, n, p=pu / pu.sum()), rng.choice(i, n, p=pi / pi.sum())
    elif kind == "zipf":
        uc, ic = (rng.zipf(1.2, n) - 1) % u, (rng.zipf(1.2, n) - 1) % i
    else:  # log-normal popularity
        pu, pi = rng.lognormal(0, 1.2, u), rng.lognormal(0, 1.5, i)
        uc, ic = rng.choice(u, n, p=pu / pu.sum()), rng.choice(i, n, p=pi / pi.sum())
    return np.stack([uc, ic, rng.integers(0, r, n)], axis=1).astype(np.int64)

CASES = [("uniform 1M K=L=16", synthetic_triples(1_000_000, 100_000, 20_000, 5, 0), 16, 16),
         ("uniform 1M K=L=24", synthetic_triples(1_000_000, 100_000, 20_000, 5, 0), 24, 24),
         ("uniform 1M K=L=32", synthetic_triples(1_000_000, 100_000, 20_000, 5, 0), 32, 32),
         ("uniform 1M K=12 L=28", synthetic_triples(1_000_000, 100_000, 20_000, 5, 0), 12, 28),
         ("log-normal 1M K=L=20", skewed(1_000_000, 100_000, 20_000, 5, "lognormal"), 20, 20),
         ("zipf 1M K=L=20", skewed(1_000_000, 100_000, 20_000, 5, "zipf"), 20, 20),
         ("log-normal sigma 0.3", skewed(1_000_000, 100_000, 20_000, 5, "", sigma=0.3), 20, 20),
         ("log-normal sigma 0.5", skewed(1_000_000, 100_000, 20_000, 5, "", sigma=0.5), 20, 20),
         ("log-normal sigma 0.8", skewed(1_000_000, 100_000, 20_000, 5, "", sigma=0.8), 20, 20),
         ("log-normal sigma 1.0", skewed(1_000_000, 100_000, 20_000, 5, "", sigma=1.0), 20, 20),
         ("100k x 943 x 1682 K=L=10", synthetic_triples(100_000, 943, 1_682, 5, 0), 10, 10),
         ("2M x 200k x 40k K=L=20", synthetic_triples(2_000_000, 200_000, 40_000, 5, 0), 20, 20)]
for name, data, k, l in CASES:
    with HipEM(data, k, l) as em:
        em.init_params(3)
        line = f"{name:>28}:"
        forced = 7
        for nt in (7, 0, 15, 9, 10, 12):   # (8: whatever the data -- the library applies the hints only without work lists)
            em.set_option("nt_out", nt)
            em.iterate(20)
            iters = 200 if len(data) >= 500_000 else 1000
            us = min(em.time_iterations(iters) for _ in range(3)) * 1000 / iters
            line += f"  nt={nt} {us:8.2f}"
        em.set_option("nt_out", 7)
        print(line + f"  ({int(em.get_option('launches'))} launches; the library's own choice: nt_out = {int(em.get_option('nt_out'))}; work items {int(em.get_option('items_pairs'))} / {int(em.get_option('items_users'))})", flush=True)

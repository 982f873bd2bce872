"""EM iteration time at an arbitrary synthetic shape: shape_time.py N U I R K L [zipf] [swap0|swap1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM, _lib
n, u, i, r, k, l = (int(x) for x in sys.argv[1:7])
zipf = "zipf" in sys.argv[7:]
swap = 0 if "swap0" in sys.argv[7:] else (1 if "swap1" in sys.argv[7:] else -1)
rng = np.random.default_rng(0)
if zipf:
    uu = (rng.zipf(1.2, n) - 1) % u
    ii = (rng.zipf(1.2, n) - 1) % i
elif "lognorm" in sys.argv[7:]:  # MovieLens-like: log-normal activity / popularity (sigma = 1)
    def draw(m):
        w = rng.lognormal(0.0, 1.0, m)
        return rng.choice(m, size=n, p=w / w.sum())
    uu, ii = draw(u), draw(i)
else:
    uu, ii = rng.integers(0, u, n), rng.integers(0, i, n)
cols = [np.unique(c, return_inverse=True)[1] for c in (uu, ii, rng.integers(0, r, n))]
train = np.stack(cols, axis=1).astype(np.int64)
nu, ni, nr = (int(train[:, j].max()) + 1 for j in range(3))
em = HipEM(train, k, l, nu, ni, nr, swap_sides=swap)
em.init_params(np.random.SeedSequence(1))
em.iterate(20)
iters = 500 if n <= 2_000_000 else 50
us = min(em.time_iterations(iters) for _ in range(3)) * 1000 / iters
prof = em.profile_iterations(20)
rd = n * (12 + 8 * k + 8 * l) + 8 * k * l * nr
print(f"N={n} U={nu} I={ni} R={nr} K={k} L={l}{' zipf' if zipf else (' lognorm' if 'lognorm' in sys.argv[7:] else '')}: {us:8.2f} us/iteration = {1e6 / us:9.1f} it/s, "
      f"{rd / us / 1e3:7.1f} GB/s algorithmic ({rd / us / 1e3 / 80:.1f} % of 8 TB/s); pairs {em.n_pairs}, swapped {em.swapped}\n   " +
      "  ".join(f"{nm} {v[0]:.1f}x{v[1]}" for nm, v in prof.items()))

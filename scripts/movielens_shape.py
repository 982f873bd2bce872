"""The MovieLens-100k shape (100k ratings, 943 users x 1,682 items, log-normal popularity): microseconds per EM iteration and
per launch (the triple passes are timed as a stage there: seg_pass + the combine launch for split segments)."""
import os
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM
rng = np.random.default_rng(0)
n, u, i = 100_000, 943, 1682
pu, pi = rng.lognormal(0, 0.8, u), rng.lognormal(0, 1.2, i)
data = np.stack([rng.choice(u, n, p=pu / pu.sum()), rng.choice(i, n, p=pi / pi.sum()), rng.integers(0, 5, n)], axis=1).astype(np.int64)
for k in (10, 20):
    with HipEM(data, k, k) as em:
        em.init_params(1); em.iterate(20)
        us = min(em.time_iterations(500) for _ in range(3)) * 1000 / 500
        prof = em.profile_iterations(100)
        print(f"K=L={k}: {us:.2f} us per iteration; launches {int(em.get_option('launches'))}; items {int(em.get_option('items_pairs'))}/{int(em.get_option('items_users'))}")
        print("   " + "  ".join(f"{nm} {v[0]:.2f}x{v[1]}" for nm, v in prof.items() if v[1] > 0))

"""Round 4: where the two-launch form with whole-segment lists beats the separate launches on data with uneven degrees
(log-normal popularity, sigma 0.8 users / 1.2 items), by size.   usage: fused_split_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM


def lognormal(n, u, i, r, seed=0):
    rng = np.random.default_rng(seed)
    pu, pi = rng.lognormal(0, 0.8, u), rng.lognormal(0, 1.2, i)
    return np.stack([rng.choice(u, n, p=pu / pu.sum()), rng.choice(i, n, p=pi / pi.sum()), rng.integers(0, r, n)], axis=1).astype(np.int64)


for n, u, i, k in ((100_000, 943, 1_682, 10), (200_000, 2_000, 3_000, 10), (400_000, 3_000, 3_000, 10), (700_000, 5_000, 3_500, 10),
                   (1_000_000, 6_040, 3_706, 10), (300_000, 3_000, 3_000, 20), (600_000, 6_000, 4_000, 20), (1_000_000, 6_040, 3_706, 20),
                   (1_000_000, 100_000, 20_000, 10)):
    data = lognormal(n, u, i, 5)
    with HipEM(data, k, k) as em:
        em.init_params(1)
        split, choice = em.get_option("fused_split"), int(em.get_option("launches"))
        out = {}
        for fused in (0, 1):
            try:
                em.set_option("fused", fused)
            except Exception as exc:  # noqa: BLE001
                out[fused] = float("nan")
                continue
            em.iterate(20)
            out[fused] = min(em.time_iterations(300) for _ in range(3)) / 300 * 1e3
        print(f"{n:>8} x {u} x {i} K=L={k}: separate launches {out[0]:7.2f} us   two launches {out[1]:7.2f} us   "
              f"(whole-segment lists {int(split)}, splits {int(em.get_option('splits_pairs'))}/{int(em.get_option('splits_users'))}; library: {choice})", flush=True)

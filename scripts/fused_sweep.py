"""Two launches (fused_small.hpp) against four per EM iteration, over problem sizes: microseconds per iteration
(HIP events, 300 iterations, best of 3).  usage: python scripts/fused_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM
from mmsbm_amd.synthetic import synthetic_triples

CASES = [(100, 5, 10, 5, 2, 4), (10_000, 1_000, 500, 5, 10, 10), (100_000, 10_000, 5_000, 5, 10, 10),
         (200_000, 20_000, 8_000, 5, 10, 10), (300_000, 30_000, 10_000, 5, 10, 10), (500_000, 50_000, 12_000, 5, 10, 10),
         (100_000, 10_000, 5_000, 5, 20, 20), (300_000, 30_000, 10_000, 5, 20, 20), (600_000, 60_000, 15_000, 5, 20, 20),
         (1_000_000, 100_000, 20_000, 5, 20, 20), (400_000, 40_000, 11_000, 5, 16, 16), (600_000, 60_000, 15_000, 5, 4, 4),
         (1_000_000, 100_000, 20_000, 5, 2, 2), (1_400_000, 120_000, 25_000, 5, 3, 4)]
for n, u, i, r, k, l in CASES:
    train = synthetic_triples(n, u, i, r, 0)
    with HipEM(train, k, l, device=0) as em:
        em.init_params(np.random.SeedSequence(0).spawn(1)[0])
        out = {}
        choice = int(em.get_option("launches"))
        for fused in (0, 1):
            try:
                em.set_option("fused", fused)
            except Exception as exc:
                out[fused] = float("nan")
                continue
            em.iterate(20)
            out[fused] = min(em.time_iterations(300) for _ in range(3)) / 300 * 1e3
        print(f"{n:>9} ratings x {u} x {i}, R={r}, K={k}, L={l}: four launches {out[0]:7.2f} us, two {out[1]:7.2f} us "
              f"(library's choice: {'two' if choice == 2 else 'four'})", flush=True)

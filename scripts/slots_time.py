"""Restart slots (N1): device time per EM iteration per restart when S restarts share one
context's launches.  usage: slots_time.py [c1|c2|c3 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from mmsbm_amd import HipEM, MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples



def problem(tag):
    n, users, items, ratings, k, l = CONFIGS[tag]
    return synthetic_triples(n, users, items, ratings, seed=0), k, l


for tag in sys.argv[1:] or ["c1", "c2", "c3"]:
    train, k, l = problem(tag)
    mm = MMSBM(k, l, iterations=1, sampling=16, seed=0)
    mm._prepare_objects(train)
    base = None
    for slots in (1, 2, 4, 8, 16):
        em = HipEM(train, k, l, mm.p + 1, mm.m + 1, mm._dims["n_ratings"], slots=slots)
        d_u, d_i = em.degrees()
        for s in range(slots):
            em.select(s).set_params(*mm.init_params(mm.child_states[s], d_u, d_i))
        iters = 200 if tag == "c3" else 1000
        line = f"{tag} slots={slots:2d}"
        for fused in ((1, 0) if em.get_option("fused") else (0,)):   # small problems: two launches, and four for comparison
            em.set_option("fused", fused)
            em.iterate(20)
            ms = min(em.time_iterations(iters) for _ in range(3))
            us = ms * 1000 / iters
            base = base or us
            line += (f"  [{'two' if fused else 'four'} launches] {us:9.2f} us/iteration  {us / slots:8.2f} us per restart-iteration"
                     f"  x{base / (us / slots):5.2f}")
        print(line, flush=True)
        em.close()

"""Skinny rating tiles (one side below 16 groups, K x L > 1,024) against the square shape of the same K x L:
microseconds per EM iteration and per stage, with the library's own choice of pair-stage kernels and with each
family forced.  usage: python scripts/skinny_time.py [N U I R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM
from mmsbm_amd.synthetic import synthetic_triples

n, u, i, r = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1_000_000, 100_000, 20_000, 5)
train = synthetic_triples(n, u, i, r, 0)
SHAPES = [(600, 5), (55, 55), (300, 8), (49, 49), (8, 520), (64, 65), (1024, 3), (5, 600), (3, 1024), (16, 200), (200, 12)]
for k, l in SHAPES:
    line = f"K={k:5d} L={l:5d} (K x L = {k * l:6d}):"
    for label, opt in (("library", None), ("vector/wide", 0), ("matrix cores", 1), ("blocked matrix cores", 2)):
        try:
            with HipEM(train, k, l, device=0) as em:
                if opt is not None:
                    em.set_option("mfma", opt)
                    if em.get_option("mfma") != opt:
                        continue
                chosen = f"mfma={em.get_option('mfma'):g} wide={em.get_option('wide'):g}"
                em.init_params(np.random.SeedSequence(0).spawn(1)[0])
                em.iterate(3)
                it = min(em.time_iterations(10) for _ in range(3)) * 100
                st = [min(em.time_stage(s, 5) for _ in range(2)) for s in range(4)]
            line += f"\n      {label:22s} [{chosen}] iteration {it:9.1f} us   seg {st[0]:8.1f}  T+S {st[1]:8.1f}  eta_p {st[2]:8.1f}  A {st[3]:8.1f}"
        except Exception as exc:
            line += f"\n      {label:22s} failed: {str(exc)[:90]}"
    print(line, flush=True)

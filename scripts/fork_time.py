"""Round 4: the user pass of an iteration beside its dense chain (second stream, option "fork") against the serial
order: microseconds per EM iteration.   usage: fork_time.py [config | n,u,i,r,k,l] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mmsbm_amd import HipEM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples

for arg in sys.argv[1:] or ["c5", "c3"]:
    n, u, i, r, k, l = CONFIGS[arg] if arg in CONFIGS else (int(x) for x in arg.split(","))
    train = synthetic_triples(n, u, i, r, 0)
    with HipEM(train, k, l) as em:
        em.init_params(1)
        em.iterate(5)
        ref = None
        for fork in (0, 1, 0, 1):
            em.set_option("fork", fork)
            em.init_params(1)
            em.iterate(3)
            out = em.get_params()
            same = True if ref is None else all(np.array_equal(a, b) for a, b in zip(out, ref))
            ref = out
            em.iterate(5)
            reps = 30 if n >= 4_000_000 else 200
            us = min(em.time_iterations(reps) for _ in range(3)) * 1000 / reps
            print(f"{arg:>28s} fork={fork}: {us:9.2f} us per iteration  (bitwise the serial order: {same}; library's choice: "
                  f"{'on' if em.get_option('forked') and fork else ''})", flush=True)

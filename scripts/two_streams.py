"""Aggregate throughput of S independent restarts sharing one GPU (one context / stream each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM, HipEM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
n, u, i, r, k, l = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
train = synthetic_triples(n, u, i, r, 0)
for S in (1, 2, 3, 4):
    mm = MMSBM(k, l, iterations=1, sampling=S, seed=0); mm._prepare_objects(train)
    ctxs = [HipEM(train, k, l, mm.p + 1, mm.m + 1, r) for _ in range(S)]
    d_u, d_i = ctxs[0].degrees()
    for s, c in enumerate(ctxs):
        c.set_params(*mm.init_params(mm.child_states[s], d_u, d_i)); c.iterate(20)
    iters = 400
    best = 1e9
    for rep in range(3):
        for c in ctxs: c.synchronize()
        t0 = time.perf_counter()
        for chunk in range(iters // 20):          # interleave submissions so every stream stays fed
            for c in ctxs: c.iterate(20, sync=False)
        for c in ctxs: c.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{S} concurrent restarts: {best / iters * 1e6:8.2f} us per iteration-step of all, "
          f"{S * iters / best:9.1f} it/s aggregate")
    for c in ctxs: c.close()

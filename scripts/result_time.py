"""likelihood() + get_params() against result() (download overlapped with the likelihood kernels): result_time.py <config> [slots]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM, MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, u, i, r, k, l = CONFIGS[cfg]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, sampling=8, seed=0); mm._prepare_objects(train)
em = HipEM(train, k, l, mm.p + 1, mm.m + 1, mm._dims["n_ratings"], slots=slots)
for s in range(slots):
    em.select(s).init_params(mm.child_states[s])
em.iterate(5)
for rep in range(3):
    t0 = time.perf_counter()
    a = [(em.select(s).likelihood(),) + tuple(em.get_params()) for s in range(slots)]
    t1 = time.perf_counter()
    b = [em.select(s).result() for s in range(slots)]
    t2 = time.perf_counter()
    same = all(x[0] == y[0] and all(np.array_equal(p, q) for p, q in zip(x[1:], y[1:])) for x, y in zip(a, b))
    print(f"{cfg} slots={slots}: likelihood + get_params {1e3 * (t1 - t0) / slots:7.1f} ms per restart, result {1e3 * (t2 - t1) / slots:7.1f} ms  (identical: {same})", flush=True)
    del a, b

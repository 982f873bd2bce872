"""Per restart-iteration time with 1 and S restart slots at an arbitrary shape: slots_shape.py N U I R K L [S ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import HipEM, MMSBM
from mmsbm_amd.synthetic import synthetic_triples
n, u, i, r, k, l = (int(x) for x in sys.argv[1:7])
counts = [int(x) for x in sys.argv[7:]] or [1, 8]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, sampling=16, seed=0); mm._prepare_objects(train)
for slots in counts:
    em = HipEM(train, k, l, mm.p + 1, mm.m + 1, mm._dims["n_ratings"], slots=slots)
    for s in range(slots):
        em.select(s).init_params(mm.child_states[s])
    iters = 10
    em.iterate(3)
    us = min(em.time_iterations(iters) for _ in range(3)) * 1000 / iters
    print(f"N={n} U={u} I={i} R={r} K={k} L={l} slots={slots:2d}: {us / slots:10.2f} us per restart-iteration "
          f"(resident set per slot {em.bytes_per_slot / 2**20:.0f} MiB)", flush=True)
    em.close()

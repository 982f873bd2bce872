"""Restart slots with the multi-slot triple pass on / off: slots_opt.py <config> [slots ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import HipEM, MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
tag = sys.argv[1]
counts = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
n, u, i, r, k, l = CONFIGS[tag]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, sampling=16, seed=0); mm._prepare_objects(train)
for slots in counts:
    em = HipEM(train, k, l, mm.p + 1, mm.m + 1, mm._dims["n_ratings"], slots=slots)
    for s in range(slots):
        em.select(s).init_params(mm.child_states[s])
    iters = 200 if n <= 1_000_000 else 20
    row = []
    for sw in (1, 0):
        em.set_option("slot_waves", sw)
        em.iterate(5)
        us = min(em.time_iterations(iters) for _ in range(3)) * 1000 / iters
        prof = em.profile_iterations(10)
        row.append(f"slot_waves={sw}: {us / slots:9.2f} us per restart-iteration (seg_pass {prof['seg_pass_kernel'][0] / slots:8.2f})")
    print(f"{tag} slots={slots:2d}  " + "   ".join(row), flush=True)
    em.close()

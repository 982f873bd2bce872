"""Per-kernel time (HIP events) with S restart slots, per restart."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import HipEM, MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
tag = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, u, i, r, k, l = CONFIGS[tag]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, sampling=16, seed=0); mm._prepare_objects(train)
for slots in (1, 2, 4, 8):
    em = HipEM(train, k, l, mm.p + 1, mm.m + 1, mm._dims["n_ratings"], slots=slots)
    for s in range(slots):
        em.select(s).init_params(mm.child_states[s])
    em.iterate(20)
    prof = em.profile_iterations(30)
    tot = sum(v[0] for v in prof.values())
    print(f"{tag} slots={slots}: " + "  ".join(f"{nm} {v[0] / slots:6.2f}" for nm, v in prof.items()) +
          f"   sum {tot / slots:6.2f} us per restart", flush=True)
    em.close()

"""Microseconds per EM iteration on the library's own path.  usage: iter_time.py shape[:slots] ...
(shape = c1|c2|c3|c5 or N,U,I,R,K,L)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
for spec in sys.argv[1:] or ["c3"]:
    tag, _, slots = spec.partition(":")
    slots = int(slots or 1)
    n, u, i, r, k, l = CONFIGS[tag] if tag in CONFIGS else tuple(int(x) for x in tag.split(","))
    train = synthetic_triples(n, u, i, r, 0)
    mm = MMSBM(k, l, iterations=1, sampling=slots, seed=0); mm._prepare_objects(train)
    ctx = mm._ctx(0); ctx.set_slots(slots)
    for s in range(slots):
        ctx.select(s).init_params(mm.child_states[s])
    ctx.iterate(25)
    iters = 1000 if n * slots <= 300000 else (200 if n * slots <= 3000000 else 40)
    us = min(ctx.time_iterations(iters) for _ in range(3)) * 1000 / iters
    print(f"{spec:>34}: {us:9.2f} us per iteration ({int(ctx.get_option('launches'))} launches)  likelihood {ctx.select(0).likelihood():.6f}", flush=True)
    ctx.close()

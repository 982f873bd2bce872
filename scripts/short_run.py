"""Where a 20-step timed region loses time against the 1,000-step rate: fixed cost vs slope."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
n, u, i, r, k, l = CONFIGS["c3"]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.init_params(mm.child_states[0]); ctx.iterate(5)
def run(steps, tsync, idle):
    ctx.synchronize(); torch.cuda.synchronize()
    if idle: time.sleep(idle)
    t0 = time.perf_counter()
    ctx.iterate(steps, sync=False)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    if tsync: torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t0) * 1e6, (t3 - t0) * 1e6
for idle in (0, 0.002, 0.05, 1.0):
    for steps in (20, 40, 100, 1000):
        rs = [run(steps, True, idle) for _ in range(5)]
        e, s, t = (np.median([x[j] for x in rs]) for j in range(3))
        ev = ctx.time_iterations(steps) * 1000
        print(f"idle {idle:5.3f}s steps {steps:5d}: enqueue {e:8.1f} us, +ctx.sync {s:9.1f} us ({s / steps:7.2f}/step), +torch.sync {t:9.1f} us ({t / steps:7.2f}/step); events {ev / steps:7.2f}/step", flush=True)

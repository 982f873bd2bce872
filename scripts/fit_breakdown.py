"""Where the wall clock of a batched fit goes (host RNG, uploads, EM, likelihood, downloads)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples

n, u, i, r, k, l = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 8
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=iters, sampling=slots, seed=0)
t = [time.perf_counter()]
mm._prepare_objects(train); ctx = mm._ctx(0); t.append(time.perf_counter())
ctx.set_slots(slots); t.append(time.perf_counter())
d_u, d_i = ctx.degrees(); t.append(time.perf_counter())
params = [mm.init_params(s, d_u, d_i) for s in mm.child_states[:1]]; t.append(time.perf_counter())
for s in range(slots):
    ctx.select(s).init_params(mm.child_states[s])
t.append(time.perf_counter())
ctx.iterate(iters); t.append(time.perf_counter())
liks = [ctx.select(s).likelihood() for s in range(slots)]; t.append(time.perf_counter())
res = [ctx.select(s).get_params() for s in range(slots)]; t.append(time.perf_counter())
names = ["context", "set_slots", "degrees", "(host rng init of ONE restart, for comparison)", "device init_params", f"{iters} iterations", "likelihood", "get_params"]
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'c3'} slots={slots}: " +
      "  ".join(f"{nm} {1000 * (b - a):.1f} ms" for nm, a, b in zip(names, t, t[1:])) +
      f"   total {t[-1] - t[0]:.3f} s")
# second round on the warm context (pinned staging and allocations in place)
t = [time.perf_counter()]
for s in range(slots):
    ctx.select(s).set_params(*params[0])
t.append(time.perf_counter())
ctx.iterate(iters); t.append(time.perf_counter())
res = [ctx.select(s).get_params() for s in range(slots)]; t.append(time.perf_counter())
print("warm: " + "  ".join(f"{nm} {1000 * (b - a):.1f} ms" for nm, a, b in
                           zip(["set_params", f"{iters} iterations", "get_params"], t, t[1:])))

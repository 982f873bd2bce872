"""Effect of a library option (mmsbm_hip_set_option) on results and on the stage / iteration times.
usage: option_time.py <config> <option> [value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mmsbm_amd import MMSBM, _lib
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples

n, u, i, r, k, l = CONFIGS[sys.argv[1]]
opt = sys.argv[2]
values = [float(v) for v in sys.argv[3:]] or [0.0, 1.0]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0)
mm._prepare_objects(train)
ctx = mm._ctx(0)
d_u, d_i = ctx.degrees()
start = mm.init_params(mm.child_states[0], d_u, d_i)
lib = _lib.load()
outs = []
for v in values:
    ctx.set_option(opt, v)
    ctx.set_params(*start)
    ctx.iterate(5)
    outs.append(ctx.get_params())
    ctx.iterate(30)
    reps = 300 if n <= 1_000_000 else 30
    it = min(ctx.time_iterations(reps) for _ in range(3)) * 1000 / reps
    st = [ctx.time_stage(s, 100 if n <= 1_000_000 else 10) for s in range(4)]
    print(f"{opt}={v:g}: iteration {it:8.2f} us   stages " + "  ".join(f"{x:7.2f}" for x in st), flush=True)
print("bitwise identical to the first:", [all(np.array_equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:]])

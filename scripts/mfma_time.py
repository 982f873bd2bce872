"""Pair stage on the matrix cores vs the lane-per-pair kernels: stage and iteration times.
usage: mfma_time.py <config|K,L> [slots]   (MMSBM_HIP_LIBRARY selects a differently compiled library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mmsbm_amd import MMSBM, _lib
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples

arg = sys.argv[1]
if arg in CONFIGS:
    n, u, i, r, k, l = CONFIGS[arg]
else:
    k, l = (int(x) for x in arg.split(","))
    n, u, i, r = 4_000_000, 400_000, 50_000, 8
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0)
mm._prepare_objects(train)
ctx = mm._ctx(0)
d_u, d_i = ctx.degrees()
start = mm.init_params(mm.child_states[0], d_u, d_i)
lib = _lib.load()
print("library's choice: mfma =", ctx.get_option("mfma"), flush=True)
outs = []
for opts in ({"mfma": 0}, {"mfma": 1}):
    for key, v in opts.items():
        ctx.set_option(key, v)
    ctx.set_params(*start)
    ctx.iterate(3)
    outs.append(ctx.get_params())
    ctx.iterate(10)
    reps = 30
    it = min(ctx.time_iterations(reps) for _ in range(3)) * 1000 / reps
    st = [min(ctx.time_stage(s, 10) for _ in range(2)) for s in range(4)]
    print(f"{opts}: iteration {it:8.2f} us   stages " + "  ".join(f"{x:7.2f}" for x in st), flush=True)
err = [max(float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300))) for a, b in zip(outs[0], o)) for o in outs[1:]]
print("max relative difference to the lane-per-pair kernels after 3 iterations:", err)

"""Intrinsic time of every stage of the iteration, launched back to back on its own."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM, _lib
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
n, u, i, r, k, l = cfg
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); d_u, d_i = ctx.degrees()
ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i)); ctx.iterate(3)
lib = _lib.load()
for st in range(lib.mmsbm_hip_kernel_count()):
    print(f"{lib.mmsbm_hip_kernel_name(st).decode():28s} {ctx.time_stage(st, 100):8.2f} us back-to-back")
ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i))
for mode, nm in ((0, "eager"), (1, "graph")):
    ctx.set_graph_mode(mode); ctx.iterate(10)
    best = min(ctx.time_iterations(200) for _ in range(3))
    print(f"iteration, {nm:14s} {best * 5:8.2f} us")

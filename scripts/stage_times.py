"""Intrinsic time of every stage of the iteration, launched back to back on its own."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diagnostic_build
use_diagnostic_build("MMSBM_ABLATE")   # the stage bits 8+ (phase ablation) exist in this build only
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
for st in range(4):
    print(f"{lib.mmsbm_hip_kernel_name(st).decode():28s} {ctx.time_stage(st, 100):8.2f} us back-to-back")
ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i))
for st in range(4):
    print(f"{lib.mmsbm_hip_kernel_name(st).decode():28s} {ctx.time_stage(st, 100):8.2f} us back-to-back")
names = {1: "rows", 2: "eta rows", 4: "S", 8: "mat-vec", 16: "out copy", 32: "slab store"}
for st in (1, 3):
    for abl in (63, 63 - 1, 63 - 2, 63 - 4, 63 - 8, 63 - 16, 63 - 32):
        kept = "+".join(v for k, v in names.items() if not (abl & k)) or "nothing"
        print(f"  stage {st} with only [{kept}]: {ctx.time_stage(st | (abl << 8), 100):8.2f} us")
ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i))
def best(n=200, reps=3):
    ctx.iterate(10)
    return min(ctx.time_iterations(n) for _ in range(reps)) * 1000.0 / n
for mode, nm in ((0, "eager"), (1, "graph")):
    ctx.set_graph_mode(mode)
    print(f"iteration, {nm:6s} {best():8.2f} us")

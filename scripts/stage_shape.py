"""Stage times of one iteration at an arbitrary shape: stage_shape.py N U I R K L"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import MMSBM, _lib
from mmsbm_amd.synthetic import synthetic_triples
n, u, i, r, k, l = (int(x) for x in sys.argv[1:7])
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.init_params(mm.child_states[0]); ctx.iterate(3)
lib = _lib.load()
reps = 20
it = min(ctx.time_iterations(reps) for _ in range(3)) * 1000 / reps
st = [min(ctx.time_stage(s, 10) for _ in range(2)) for s in range(4)]   # (the four launches; 4, 5 = the two-launch form)
print(f"N={n} U={u} I={i} R={r} K={k} L={l}: mfma={ctx.get_option('mfma'):g} wide={ctx.get_option('wide'):g} "
      f"iteration {it:9.2f} us  stages " + "  ".join(f"{x:8.2f}" for x in st), flush=True)

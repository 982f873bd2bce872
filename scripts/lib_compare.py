"""seg_pass stage time and iteration time of the library selected by MMSBM_HIP_LIBRARY."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, u, i, r, k, l = CONFIGS[name]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0)
ctx.init_params(mm.child_states[0]); ctx.iterate(30)
reps = 300 if n <= 1_000_000 else 30
it = sorted(ctx.time_iterations(reps) * 1000 / reps for _ in range(5))
st = [ctx.time_stage(s, 100 if n <= 1_000_000 else 10) for s in range(4)]
print(f"{os.environ.get('MMSBM_HIP_LIBRARY', 'default'):60s} {name} iteration min {it[0]:8.2f} median {it[2]:8.2f}  stages " + " ".join(f"{x:7.2f}" for x in st), flush=True)

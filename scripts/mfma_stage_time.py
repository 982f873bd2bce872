"""Stage and iteration times of the library's own path (for A/B builds, scripts/ab_mfma.sh).  usage: mfma_stage_time.py <config|K,L>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
arg = sys.argv[1] if len(sys.argv) > 1 else "c5"
if arg in CONFIGS:
    n, u, i, r, k, l = CONFIGS[arg]
else:
    k, l = (int(x) for x in arg.split(","))
    n, u, i, r = 4_000_000, 400_000, 50_000, 8
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.init_params(mm.child_states[0]); ctx.iterate(5)
it = min(ctx.time_iterations(30) for _ in range(3)) * 1000 / 30
st = [min(ctx.time_stage(s, 10) for _ in range(3)) for s in range(4)]
print(f"{arg}: iteration {it:8.2f} us   stages " + "  ".join(f"{x:7.2f}" for x in st), flush=True)

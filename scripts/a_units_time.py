"""The matrix-core A launch on runs of A_UNITS 64-pair units per workgroup (option "a_units"; unset / 0: the library's own
choice, mmsbm_hip.hip: balanced_run_units): the launch back to back and the whole iteration.
usage: [A_UNITS=7] python scripts/a_units_time.py c5 | N,U,I,R,K,L"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
tag = sys.argv[1] if len(sys.argv) > 1 else "c5"
n, u, i, r, k, l = CONFIGS[tag] if tag in CONFIGS else tuple(int(x) for x in tag.split(","))
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.set_option("a_units", int(os.environ.get("A_UNITS", "0"))); ctx.init_params(mm.child_states[0]); ctx.iterate(5)
a = min(ctx.time_stage(3, 50) for _ in range(3))
ctx.init_params(mm.child_states[0]); ctx.iterate(5)
it = min(ctx.time_iterations(50) for _ in range(3)) * 1000.0 / 50
print(f"{tag} a_units={ctx.get_option('a_units'):.0f}: A launch {a:7.2f} us, iteration {it:8.2f} us; workgroups: T+S {ctx.get_option('n_chunks'):.0f}, A {ctx.get_option('a_chunks') or ctx.get_option('n_chunks'):.0f}", flush=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diagnostic_build
use_diagnostic_build("MMSBM_ABLATE")   # the stage bits 8+ (phase ablation) exist in this build only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
n, u, i, r, k, l = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.init_params(mm.child_states[0]); ctx.iterate(3)
for abl, nm in ((0, "both roles"), (64, "item_sum only"), (128, "p_update only"), (192, "neither (launch floor)")):
    print(f"eta_p {nm:24s} {ctx.time_stage(2 | (abl << 8), 200):7.2f} us")

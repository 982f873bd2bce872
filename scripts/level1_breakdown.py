"""Where a level-1 call (kernels_hip.update_coefficients: numpy in, numpy out, src/kernels_numpy.py:43-46) spends its time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import kernels_hip, MMSBM
from mmsbm_amd.core import data_key
from mmsbm_amd.mmsbm import normalize_with_self
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
n, u, i, r, k, l = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, seed=0); mm.p, mm.m = int(train[:, 0].max()), int(train[:, 1].max()); mm._dims = {"n_ratings": r}
d_u = np.bincount(train[:, 0]); d_i = np.bincount(train[:, 1])
theta, eta, pr = mm.init_params(mm.child_states[0], d_u, d_i)
kernels_hip.update_coefficients(train, theta, eta, pr)
ctx = kernels_hip._cache[0][1]
def best(f, reps=10):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3
print("digest of the id columns      %6.2f ms (median %6.2f)" % best(lambda: data_key(train)))
print("set_params (pack + upload + A) %6.2f ms (median %6.2f)" % best(lambda: ctx.set_params(theta, eta, pr)))
print("update_coefficients (ctx)      %6.2f ms (median %6.2f)" % best(lambda: ctx.update_coefficients()))
print("the whole level-1 call         %6.2f ms (median %6.2f)  [digest beside the GPU work]" % best(lambda: kernels_hip.update_coefficients(train, theta, eta, pr)))
nt, ne, npr = kernels_hip.update_coefficients(train, theta, eta, pr)
print("host normalisations (reference) %5.2f ms (median %6.2f)" % best(lambda: (nt / d_u[:, None], ne / d_i[:, None], normalize_with_self(npr))))

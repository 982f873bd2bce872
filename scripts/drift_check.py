"""Drift of the HIP path against the oracle over many EM iterations (a config tag or N,U,I,R,K,L):
max |diff| / max |want| for theta, eta, p at checkpoints, and argmax agreement of the
predictions under the tie rule (compare only where the oracle's top-2 gap exceeds 1e-9)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
from oracle import mmsbm_oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, u, i, r, k, l = CONFIGS[name] if name in CONFIGS else tuple(int(x) for x in name.split(","))  # or N,U,I,R,K,L
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=iters, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); d_u, d_i = ctx.degrees()
theta, eta, pr = mm.init_params(mm.child_states[0], d_u, d_i)
ctx.set_params(theta, eta, pr)
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
def elem(a, b, floor=1e-290):   # element-wise: max |diff| / |want| over the entries above the floor (tests/conftest.py)
    big = np.abs(b) > floor
    return float(np.max(np.abs(a[big] - b[big]) / np.abs(b[big]))) if big.any() else 0.0
done, t0 = 0, time.time()
for stop in (1, 10, 50, 100, 200, 400):
    if stop > iters: break
    ctx.iterate(stop - done)
    for _ in range(stop - done):
        theta, eta, pr = orc.em_step(train, theta, eta, pr, d_u, d_i)
    done = stop
    t, e, p = ctx.get_params()
    lik, lik_o = ctx.likelihood(), orc.compute_likelihood(train, theta, eta, pr)
    pd_h, pd_o = ctx.prod_dist(train), orc.prod_dist(train, theta, eta, pr)
    srt = np.sort(pd_o, axis=1); clear = (srt[:, -1] - srt[:, -2]) > 1e-9
    agree = float(np.mean(np.argmax(pd_h, 1)[clear] == np.argmax(pd_o, 1)[clear]))
    print(f"{name} it {stop:4d}: element-wise theta {elem(t, theta):.2e} eta {elem(e, eta):.2e} p {elem(p, pr):.2e} "
          f"(smallest entries {theta[theta > 0].min():.1e} / {eta[eta > 0].min():.1e} / {pr[pr > 0].min():.1e})")
    print(f"{name} it {stop:4d}: theta {rel(t, theta):.2e} eta {rel(e, eta):.2e} p {rel(p, pr):.2e} "
          f"likelihood rel {abs(lik - lik_o) / abs(lik_o):.2e}  argmax agreement {agree:.6f} on {clear.mean():.4f} of rows "
          f"[{time.time() - t0:.0f}s]", flush=True)

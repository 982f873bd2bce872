"""Iteration time on heavy-tailed (Zipf) data of C3's size: does one huge segment stall a pass?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM
rng = np.random.default_rng(0)
n = 1_000_000
u = (rng.zipf(1.15, n) - 1) % 100_000
i = (rng.zipf(1.15, n) - 1) % 20_000
data = np.stack([u, i, rng.integers(0, 5, n)], axis=1).astype(np.int64)
for c in (0, 1):
    data[:, c] = np.unique(data[:, c], return_inverse=True)[1]
mm = MMSBM(20, 20, iterations=1, seed=0); mm._prepare_objects(data)
ctx = mm._ctx(0); d_u, d_i = ctx.degrees()
print("users", len(d_u), "max user degree", d_u.max(), "items", len(d_i), "max item degree", d_i.max(), "pairs", ctx.n_pairs)
ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i)); ctx.iterate(10)
print(f"iteration {min(ctx.time_iterations(100) for _ in range(3)) * 10:.2f} us")

"""Where the time of building a context goes (MMSBM_HIP_TIMING=1): create_time.py N U I R K L"""
import os, sys, time
os.environ["MMSBM_HIP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM
n, u, i, r, k, l = (int(x) for x in sys.argv[1:7])
rng = np.random.default_rng(0)
train = np.stack([rng.integers(0, u, n), rng.integers(0, i, n), rng.integers(0, r, n)], axis=1).astype(np.int32)
train = np.asfortranarray(train)
for rep in range(2):
    t0 = time.perf_counter()
    em = HipEM(train, k, l, u, i, r)
    print(f"-- create #{rep}: {1e3 * (time.perf_counter() - t0):.1f} ms total", file=sys.stderr, flush=True)
    em.close()

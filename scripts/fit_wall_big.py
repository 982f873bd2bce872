"""fit() wall clock at a MovieLens-20M-like shape (ints in a DataFrame), phase by phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from mmsbm_amd import MMSBM
from mmsbm_amd.encode import Encoder
n, u, i, r, k, l = 20_000_000, 138_000, 27_000, 10, 20, 20
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
df = pd.DataFrame({"users": rng.integers(0, u, n), "items": rng.integers(0, i, n), "ratings": rng.integers(1, r + 1, n)})
t0 = time.perf_counter(); enc = Encoder(); train = enc.fit_transform(df); t1 = time.perf_counter()
print(f"encode {t1 - t0:.2f} s", flush=True)
os.environ["MMSBM_HIP_TIMING"] = "1"
mm = MMSBM(k, l, iterations=iters, sampling=1, seed=0)
t0 = time.perf_counter(); mm.fit(df, silent=True); t1 = time.perf_counter()
print(f"fit() total {t1 - t0:.2f} s for {iters} iterations (EM alone ~{iters * 1.22e-3:.2f} s)", flush=True)
t0 = time.perf_counter(); pm = mm.predict(df.iloc[:1_000_000]); t1 = time.perf_counter()
print(f"predict(1M rows) {t1 - t0:.2f} s  accuracy {mm.score(silent=True)['stats']['accuracy']:.4f}")

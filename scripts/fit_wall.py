"""End-to-end wall clock of MMSBM.fit / predict at a BASELINE config through the host class."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n, u, i, r, k, l = cfg
rng = np.random.default_rng(0)
df = pd.DataFrame({"users": rng.integers(0, u, n).astype(str), "items": rng.integers(0, i, n).astype(str),
                   "ratings": rng.integers(1, r + 1, n)})
mm = MMSBM(k, l, iterations=iters, sampling=1, seed=0)
t0 = time.perf_counter(); mm.data_handler = __import__("mmsbm_amd.encode", fromlist=["Encoder"]).Encoder()
train = mm.data_handler.fit_transform(df); t1 = time.perf_counter()
mm._prepare_objects(train); ctx = mm._ctx(0); t2 = time.perf_counter()
d_u, d_i = ctx.degrees(); params = mm.init_params(mm.child_states[0], d_u, d_i); t3 = time.perf_counter()
ctx.set_params(*params); t4 = time.perf_counter()
ctx.iterate(iters); t5 = time.perf_counter()
lik = ctx.likelihood(); t6 = time.perf_counter()
res = ctx.get_params(); t7 = time.perf_counter()
print(f"encode {t1-t0:.3f}s  context(sort+upload) {t2-t1:.3f}s  init rng {t3-t2:.3f}s  set_params {t4-t3:.3f}s  "
      f"{iters} iterations {t5-t4:.3f}s  likelihood {t6-t5:.4f}s  get_params {t7-t6:.3f}s   likelihood={lik:.3f}")
t0 = time.perf_counter(); mm2 = MMSBM(k, l, iterations=iters, sampling=1, seed=0); mm2.fit(df, silent=True); t1 = time.perf_counter()
pm = mm2.predict(df.iloc[:100000]); t2 = time.perf_counter()
print(f"fit() total {t1-t0:.3f}s   predict(100k rows) {t2-t1:.3f}s  accuracy {mm2.score(silent=True)['stats']['accuracy']:.4f}")
for samp in (8,):
    t0 = time.perf_counter(); mm3 = MMSBM(k, l, iterations=iters, sampling=samp, seed=0); mm3.fit(df, silent=True); t1 = time.perf_counter()
    pm = mm3.predict(df.iloc[:100000]); t2 = time.perf_counter()
    print(f"sampling={samp}: fit() {t1-t0:.3f}s   predict(100k rows, {samp} restarts) {t2-t1:.3f}s  "
          f"accuracy {mm3.score(silent=True)['stats']['accuracy']:.4f}")
    t0 = time.perf_counter(); mm4 = MMSBM(k, l, iterations=iters, sampling=samp, seed=0, restarts_per_launch=1); mm4.fit(df, silent=True); t1 = time.perf_counter()
    print(f"sampling={samp}, one restart at a time: fit() {t1-t0:.3f}s")

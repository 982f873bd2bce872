"""The whole user-level flow at a BASELINE config, from DataFrames: fit (sampling restarts as slots of one
context) -> predict -> score, wall clock per phase.
usage: user_flow.py <config> [iterations] [sampling] [test rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
n, u, i, r, k, l = CONFIGS[cfg]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sampling = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = int(sys.argv[4]) if len(sys.argv) > 4 else n // 10
frame = lambda a: pd.DataFrame({"users": a[:, 0], "items": a[:, 1], "ratings": a[:, 2]})
train, test = frame(synthetic_triples(n, u, i, r, 0)), frame(synthetic_triples(m, u, i, r, 1))
mm = MMSBM(k, l, iterations=iters, sampling=sampling, seed=0)
t0 = time.perf_counter(); mm.fit(train, silent=True); t1 = time.perf_counter()
print(f"{cfg}: fit {t1 - t0:.3f} s ({sampling} restarts x {iters} iterations = "
      f"{sampling * iters / (t1 - t0):.0f} restart-iterations/s end to end, encoder and context included)", flush=True)
t0 = time.perf_counter(); pm = mm.predict(test); t1 = time.perf_counter()
print(f"predict({m} rows, {sampling} restarts averaged) {t1 - t0:.3f} s", flush=True)
t0 = time.perf_counter(); st = mm.score(silent=True)["stats"]; t1 = time.perf_counter()
print(f"score {t1 - t0:.4f} s  accuracy {st['accuracy']:.4f} mae {st['mae']:.4f}", flush=True)
t0 = time.perf_counter(); mm.fit(train, silent=True); t1 = time.perf_counter()
print(f"second fit on the same frame {t1 - t0:.3f} s")

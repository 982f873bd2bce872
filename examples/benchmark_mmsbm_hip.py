#!/usr/bin/env python3
"""The reference's own benchmark (benchmark_mmsbm.py: fit + predict wall clock on a seeded random
data frame, same command-line flags and same data recipe) run through this package's host class
with backend='hip'.  Prints the three lines the reference prints, so the two can be put side by side:

    python examples/benchmark_mmsbm_hip.py --n_obs 200000 --user_groups 10 --item_groups 10
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

from mmsbm_amd import MMSBM  # noqa: E402


def frame(n_obs, seed=0, n_users=1_000, n_items=1_500, lo=1, hi=5):
    # benchmark_mmsbm.py:14-31 of the reference: users, items (as strings), ratings, in that draw order
    rng = np.random.default_rng(seed)
    users = rng.integers(0, n_users, size=n_obs).astype(str)
    items = rng.integers(0, n_items, size=n_obs).astype(str)
    return pd.DataFrame({"users": users, "items": items, "ratings": rng.integers(lo, hi + 1, size=n_obs)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n_obs", type=int, default=20_000)
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--user_groups", type=int, default=2)
    ap.add_argument("--item_groups", type=int, default=2)
    ap.add_argument("--sampling", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=2, help="the first run pays the one-off HIP start-up")
    args = ap.parse_args()
    data = frame(args.n_obs)
    for run in range(args.repeat):
        model = MMSBM(args.user_groups, args.item_groups, iterations=args.iterations,
                      sampling=args.sampling, seed=0, backend="hip", debug=True)
        t0 = time.perf_counter()
        model.fit(data, silent=True)
        t1 = time.perf_counter()
        model.predict(data)
        t2 = time.perf_counter()
        tag = "first run (includes HIP start-up)" if run == 0 and args.repeat > 1 else "run"
        print(f"[{tag}]\n Training time:   {t1 - t0:8.3f} s\nPrediction time: {t2 - t1:8.3f} s\n"
              f"Total time:      {t2 - t0:8.3f} s   accuracy {model.score(silent=True)['stats']['accuracy']:.4f}")
        model._release()


if __name__ == "__main__":
    main()

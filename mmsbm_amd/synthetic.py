"""Synthetic rating triples shaped like the reference's benchmark generator
(benchmark_mmsbm.py:14-31): uniform, independent columns drawn in the order users, items,
ratings from ``default_rng(seed)``, then densely re-encoded in numeric order."""
import numpy as np

# BASELINE.json configs: name -> (N, U, I, R, K, L)
CONFIGS = {
    "c1": (100, 5, 10, 5, 2, 4),
    "c2": (100_000, 10_000, 5_000, 5, 10, 10),
    "c3": (1_000_000, 100_000, 20_000, 5, 20, 20),
    "c5": (10_000_000, 1_000_000, 100_000, 10, 50, 50),
}


def synthetic_triples(n_obs, n_users, n_items, n_ratings, seed=0):
    rng = np.random.default_rng(seed)
    users = rng.integers(0, n_users, size=n_obs)
    items = rng.integers(0, n_items, size=n_obs)
    ratings = rng.integers(1, n_ratings + 1, size=n_obs)
    cols = [np.unique(c, return_inverse=True)[1].astype(np.int64) for c in (users, items, ratings)]
    return np.stack(cols, axis=1)


def algorithmic_bytes(n_obs, n_users, n_items, n_ratings, k, l):
    """SURVEY section 8d / BASELINE.md section 3: per EM iteration
    B_read = N (12 + 8K + 8L) + 8 K L R ;  B_write = 8 (U K + I L + K L R)."""
    rd = n_obs * (12 + 8 * k + 8 * l) + 8 * k * l * n_ratings
    wr = 8 * (n_users * k + n_items * l + k * l * n_ratings)
    return rd, wr

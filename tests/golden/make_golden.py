#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE REAL REFERENCE.

Build-container only: imports /root/reference/src (never copied into this repo,
never available on the GPU box).  Run from a scratch directory because the
reference's logger drops ``mmsbm.log`` into the cwd:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Everything stored is data: inputs and the reference's outputs for them.
Fixture ids follow SURVEY.md section 8c (G0..G5 + edge).
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MMSBM_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

import kernels_numpy as ref_k  # noqa: E402  (the parity target)
from backend import load_backend  # noqa: E402
from data_handler import DataHandler  # noqa: E402
from expectation_maximization import ExpectationMaximization  # noqa: E402
from mmsbm import MMSBM  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def mock_frame(seed, n=100):
    """The data recipe of the reference's tests/test_mmsbm.py:12-22."""
    rng = np.random.default_rng(seed)
    return pd.DataFrame({
        "users": [f"user{rng.choice(list(range(5)))}" for _ in range(n)],
        "items": [f"item{rng.choice(list(range(10)))}" for _ in range(n)],
        "ratings": [rng.choice(list(range(1, 6))) for _ in range(n)],
    })


def frame_cols(df, prefix):
    return {prefix + "_users": np.array([str(x) for x in df["users"]]),
            prefix + "_items": np.array([str(x) for x in df["items"]]),
            prefix + "_ratings": np.array([str(x) for x in df["ratings"]])}


def prepared(train, k, l, seed, iterations=1, sampling=1):
    mm = MMSBM(k, l, iterations=iterations, sampling=sampling, seed=seed, backend="numpy")
    mm._prepare_objects(train)
    return mm


def loop_with_snapshots(mm, train, child, iterations, snaps):
    """Drive the reference's own EM functions in the order of src/mmsbm.py:224-256."""
    rng = np.random.default_rng(child)
    k, l, r = mm._dims["n_user_groups"], mm._dims["n_item_groups"], mm._dims["n_ratings"]
    theta = mm.em.normalize_with_d(rng.random((mm.p + 1, k)), "user")
    eta = mm.em.normalize_with_d(rng.random((mm.m + 1, l)), "item")
    pr = mm.em.normalize_with_self(rng.random((k, l, r)))
    kept = {0: (theta, eta, pr)}
    first = None
    for j in range(iterations):
        n_theta, n_eta, n_pr = mm.em.update_coefficients(data=train, theta=theta, eta=eta, pr=pr)
        if j == 0:
            first = (n_theta, n_eta, n_pr)
        theta = mm.em.normalize_with_d(n_theta, "user")
        eta = mm.em.normalize_with_d(n_eta, "item")
        pr = mm.em.normalize_with_self(n_pr)
        if j + 1 in snaps:
            kept[j + 1] = (theta, eta, pr)
    lik = mm.em.compute_likelihood(train, theta, eta, pr)
    return kept, first, lik


def g0():
    """Inputs of the reference's tests/test_backends.py:21-35 and :48-60."""
    com, upd, prd, name = load_backend("numpy")
    data = np.array([[0, 0, 0], [1, 1, 1], [0, 1, 2]], dtype=np.int64)
    out = {"data": data}
    for tag, s in (("a", 0), ("b", 1)):
        rng = np.random.default_rng(s)
        theta = rng.random((2, 2)); eta = rng.random((2, 2)); pr = rng.random((2, 2, 3))
        theta /= theta.sum(axis=1, keepdims=True)
        eta /= eta.sum(axis=1, keepdims=True)
        pr /= pr.sum(axis=2, keepdims=True)
        nt, ne, npr = upd(data, theta, eta, pr)
        out.update({f"{tag}_theta": theta, f"{tag}_eta": eta, f"{tag}_pr": pr,
                    f"{tag}_omegas": com(data, theta, eta, pr),
                    f"{tag}_prod_dist": prd(data, theta, eta, pr),
                    f"{tag}_n_theta": nt, f"{tag}_n_eta": ne, f"{tag}_n_pr": npr})
    save("g0_backend_tests", **out)


def g1_g3():
    df1, df2 = mock_frame(1), mock_frame(2)
    dh = DataHandler()
    train = dh.format_train_data(df1.copy())
    test = dh.format_test_data(df2.copy())
    out = {"train": train, "test": test}
    out.update(frame_cols(df1, "train_raw")); out.update(frame_cols(df2, "test_raw"))
    out["dict_users_keys"] = np.array(list(dh.obs_dict.keys()))
    out["dict_users_vals"] = np.array(list(dh.obs_dict.values()))
    out["dict_items_keys"] = np.array(list(dh.items_dict.keys()))
    out["dict_items_vals"] = np.array(list(dh.items_dict.values()))
    out["dict_ratings_keys"] = np.array(list(dh.ratings_dict.keys()))
    out["dict_ratings_vals"] = np.array(list(dh.ratings_dict.values()))

    # --- C1 as BASELINE.json states it: K=2, L=4, 500 iterations, seed=1 ---------
    mm = prepared(train, 2, 4, seed=1)
    out["d_u"] = mm._normalization_factors["user"][:, 0]
    out["d_i"] = mm._normalization_factors["item"][:, 0]
    kept, first, lik = loop_with_snapshots(mm, train, mm.child_states[0], 500, {1, 10, 500})
    real = mm.run_one_sampling(train, mm.child_states[0], 0)  # iterations=1 ctor -> set below
    for it, (t, e, p) in kept.items():
        out[f"c1_theta_{it}"], out[f"c1_eta_{it}"], out[f"c1_pr_{it}"] = t, e, p
    out["c1_n_theta_1"], out["c1_n_eta_1"], out["c1_n_pr_1"] = first
    out["c1_likelihood_500"] = np.float64(lik)
    assert np.array_equal(real["theta"], kept[1][0])  # loop replica == the reference's own loop
    mm500 = prepared(train, 2, 4, seed=1, iterations=500)
    real500 = mm500.run_one_sampling(train, mm500.child_states[0], 0)
    assert np.array_equal(real500["theta"], kept[500][0]) and real500["likelihood"] == lik
    out["c1_likelihood_at"] = np.array(
        [mm.em.compute_likelihood(train, *kept[it]) for it in (0, 1, 10, 500)])
    out["c1_omegas_0"] = mm.em.compute_omegas(train, *kept[0])

    # --- the reference's own end-to-end test case: K=L=2, 10 iterations, seed=1 --
    m2 = MMSBM(2, 2, iterations=10, seed=1, backend="numpy")
    m2.fit(df1.copy(), silent=True)
    pm = m2.predict(df2.copy())
    sc = m2.score(silent=True)
    res = m2.results[0]
    out["t_theta"], out["t_eta"], out["t_pr"] = res["theta"], res["eta"], res["pr"]
    out["t_likelihood"] = np.float64(res["likelihood"])
    out["t_prediction_matrix"] = pm
    out["t_stats_keys"] = np.array(list(sc["stats"].keys()))
    out["t_stats_vals"] = np.array([float(np.sum(v)) for v in sc["stats"].values()])
    # G3: prod_dist, argmax, top-2 gap on the test pairs
    pd_ = ref_k.prod_dist(test, res["theta"], res["eta"], res["pr"])
    srt = np.sort(pd_, axis=1)
    out["t_prod_dist"], out["t_argmax"], out["t_gap"] = pd_, np.argmax(pd_, 1), srt[:, -1] - srt[:, -2]
    save("g1_c1_mock", **out)

    # --- G2: sampling=3, restarts are independent of `sampling` ---------------------
    m3 = MMSBM(2, 2, iterations=10, sampling=3, seed=1, backend="numpy")
    m3.fit(df1.copy(), silent=True)
    o2 = {"train": train}
    for s, rr in enumerate(m3.results):
        o2[f"theta_{s}"], o2[f"eta_{s}"], o2[f"pr_{s}"] = rr["theta"], rr["eta"], rr["pr"]
    o2["likelihoods"] = np.array([rr["likelihood"] for rr in m3.results])
    assert np.array_equal(m3.results[0]["theta"], res["theta"])
    pm3 = m3.predict(df2.copy())
    o2["prediction_matrix"] = pm3
    o2["test"] = test
    save("g2_c1_sampling3", **o2)


def uniform_triples(n, u, i, r, seed):
    """benchmark_mmsbm.py:14-31 draw order, dense re-encode in numeric order."""
    rng = np.random.default_rng(seed)
    users = rng.integers(0, u, size=n); items = rng.integers(0, i, size=n)
    ratings = rng.integers(1, r + 1, size=n)
    return np.stack([np.unique(c, return_inverse=True)[1].astype(np.int64)
                     for c in (users, items, ratings)], axis=1)


def g4():
    train = uniform_triples(2000, 200, 100, 5, 0)
    mm = prepared(train, 10, 10, seed=0)
    kept, first, lik = loop_with_snapshots(mm, train, mm.child_states[0], 50, {1, 50})
    out = {"train": train, "likelihood_50": np.float64(lik),
           "n_theta_1": first[0], "n_eta_1": first[1], "n_pr_1": first[2]}
    for it, (t, e, p) in kept.items():
        out[f"theta_{it}"], out[f"eta_{it}"], out[f"pr_{it}"] = t, e, p
    out["prod_dist_50"] = ref_k.prod_dist(train, *kept[50])
    save("g4_2k_k10", **out)


def g5():
    """C2 (100k ratings, 10k x 5k, R=5, K=L=10): sampled entries only."""
    train = uniform_triples(100_000, 10_000, 5_000, 5, 0)
    mm = MMSBM(10, 10, iterations=1, seed=0, backend="numpy")
    # skip the reference's O(U*N) _prepare_objects: build the EM object directly
    d_u = np.maximum(np.bincount(train[:, 0]), 1); d_i = np.maximum(np.bincount(train[:, 1]), 1)
    k = l = 10
    mm.p, mm.m = int(train[:, 0].max()), int(train[:, 1].max())
    mm._dims = {"n_samples": len(train), "n_user_groups": k, "n_item_groups": l,
                "n_ratings": 5}
    mm.em = ExpectationMaximization(
        dims=mm._dims, user_indices=None, item_indices=None, rating_indices=None,
        norm_factors={"user": np.repeat(d_u[:, None], k, 1), "item": np.repeat(d_i[:, None], l, 1)},
        backend="numpy")
    mm.train = train
    kept, first, lik = loop_with_snapshots(mm, train, mm.child_states[0], 50, {1, 10, 50})
    pick = np.random.default_rng(7)
    ut = pick.integers(0, mm.p + 1, 1000); kt = pick.integers(0, k, 1000)
    ie = pick.integers(0, mm.m + 1, 1000); le = pick.integers(0, l, 1000)
    out = {"n": 100_000, "u": 10_000, "i": 5_000, "r": 5, "gen_seed": 0, "model_seed": 0,
           "train_head": train[:64], "train_sum": train.sum(axis=0),
           "ut": ut, "kt": kt, "ie": ie, "le": le, "likelihood_50": np.float64(lik),
           "n_pr_1": first[2]}
    for it, (t, e, p) in kept.items():
        out[f"theta_s_{it}"] = t[ut, kt]; out[f"eta_s_{it}"] = e[ie, le]; out[f"pr_{it}"] = p
        out[f"theta_colsum_{it}"] = t.sum(axis=0); out[f"eta_colsum_{it}"] = e.sum(axis=0)
    out["likelihood_at"] = np.array([mm.em.compute_likelihood(train, *kept[it]) for it in (1, 10, 50)])
    pdist = ref_k.prod_dist(train, *kept[50])
    srt = np.sort(pdist, axis=1)
    out["argmax_50"] = np.argmax(pdist, 1).astype(np.int8)
    out["gap_50"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
    save("g5_c2_sampled", **out)


def c2_problem():
    """C2 with the EM object built directly (skipping the reference's O(U*N) _prepare_objects)."""
    train = uniform_triples(100_000, 10_000, 5_000, 5, 0)
    mm = MMSBM(10, 10, iterations=1, seed=0, backend="numpy")
    d_u = np.maximum(np.bincount(train[:, 0]), 1); d_i = np.maximum(np.bincount(train[:, 1]), 1)
    k = l = 10
    mm.p, mm.m = int(train[:, 0].max()), int(train[:, 1].max())
    mm._dims = {"n_samples": len(train), "n_user_groups": k, "n_item_groups": l, "n_ratings": 5}
    mm.em = ExpectationMaximization(
        dims=mm._dims, user_indices=None, item_indices=None, rating_indices=None,
        norm_factors={"user": np.repeat(d_u[:, None], k, 1), "item": np.repeat(d_i[:, None], l, 1)},
        backend="numpy")
    mm.train = train
    return mm, train


def g5_long():
    """C2 over the reference's DEFAULT run length (iterations=400, src/mmsbm.py:63-72): the same problem, start
    and sampled positions as G5, snapshots after 100, 200 and 400 iterations -- sampled theta / eta entries, column
    sums, full p, the likelihood, and the argmax prediction of every training row with a bit saying whether its
    top-2 gap exceeds 1e-9 (the tie rule of SURVEY 7.3 item 6).  About two minutes of reference time."""
    mm, train = c2_problem()
    k = l = 10
    snaps = (100, 200, 400)
    kept, _, lik = loop_with_snapshots(mm, train, mm.child_states[0], 400, set(snaps))
    pick = np.random.default_rng(7)
    ut = pick.integers(0, mm.p + 1, 1000); kt = pick.integers(0, k, 1000)
    ie = pick.integers(0, mm.m + 1, 1000); le = pick.integers(0, l, 1000)
    out = {"n": 100_000, "u": 10_000, "i": 5_000, "r": 5, "gen_seed": 0, "model_seed": 0,
           "train_sum": train.sum(axis=0), "ut": ut, "kt": kt, "ie": ie, "le": le,
           "snapshots": np.array(snaps), "likelihood_400": np.float64(lik)}
    for it in snaps:
        t, e, p = kept[it]
        out[f"theta_s_{it}"] = t[ut, kt]; out[f"eta_s_{it}"] = e[ie, le]; out[f"pr_{it}"] = p
        out[f"theta_colsum_{it}"] = t.sum(axis=0); out[f"eta_colsum_{it}"] = e.sum(axis=0)
        out[f"theta_min_{it}"] = np.float64(t[t > 0].min()) if (t > 0).any() else np.float64(0)
        pdist = ref_k.prod_dist(train, t, e, p)
        srt = np.sort(pdist, axis=1)
        out[f"argmax_{it}"] = np.argmax(pdist, 1).astype(np.int8)
        out[f"clear_{it}"] = np.packbits((srt[:, -1] - srt[:, -2]) > 1e-9)
    out["likelihood_at"] = np.array([mm.em.compute_likelihood(train, *kept[it]) for it in snaps])
    save("g5_c2_400", **out)


def uneven_frames():
    """A rating table shaped like a real one: string user / item ids, a few busy users, popular items (log-normal
    popularity on both sides) -- 20,000 ratings of 150 users x 260 items, and a 2,000-row test sample of it."""
    rng = np.random.default_rng(5)
    n, n_u, n_i = 20_000, 150, 260
    pu, pi = rng.lognormal(0, 0.9, n_u), rng.lognormal(0, 1.3, n_i)
    df = pd.DataFrame({"users": [f"u{x:03d}" for x in rng.choice(n_u, n, p=pu / pu.sum())],
                       "items": [f"film-{x}" for x in rng.choice(n_i, n, p=pi / pi.sum())],
                       "ratings": rng.integers(1, 6, n)})
    return df, df.sample(2000, random_state=1)


def g7_uneven():
    """The reference's own fit -> predict -> score on that table (K = 6, L = 7, 40 iterations, sampling = 2, seed = 3):
    every restart's theta / eta / pr and likelihood, the prediction matrix, the scores."""
    df, test_df = uneven_frames()
    mm = MMSBM(6, 7, iterations=40, sampling=2, seed=3, backend="numpy")
    mm.fit(df.copy(), silent=True)
    pm = mm.predict(test_df.copy())
    sc = mm.score(silent=True)
    out = {"train": mm.train, "test": mm.test, "prediction_matrix": pm,
           "stats_keys": np.array(list(sc["stats"].keys())),
           "stats_vals": np.array([float(np.sum(v)) for v in sc["stats"].values()]),
           "likelihoods": np.array([float(r["likelihood"]) for r in mm.results])}
    for s_, r in enumerate(mm.results):
        out[f"theta_{s_}"], out[f"eta_{s_}"], out[f"pr_{s_}"] = r["theta"], r["eta"], r["pr"]
    out.update(frame_cols(df, "train_raw")); out.update(frame_cols(test_df, "test_raw"))
    out["test_index"] = np.asarray(test_df.index)
    save("g7_uneven_strings", **out)


def edge():
    com, upd, prd, _ = load_backend("numpy")
    norm = ExpectationMaximization.normalize_with_self
    out = {}
    rng = np.random.default_rng(11)
    # (i) a user group that nobody belongs to -> all-zero (k, l, :) rows of n_pr
    data = np.array([[0, 0, 0], [0, 1, 2], [1, 1, 1], [2, 0, 2], [2, 2, 0], [1, 2, 2]], dtype=np.int64)
    theta = rng.random((3, 3)); theta[:, 1] = 0.0
    eta = rng.random((3, 2)); pr = norm(rng.random((3, 2, 3)))
    nt, ne, npr = upd(data, theta, eta, pr)
    out.update(zero_data=data, zero_theta=theta, zero_eta=eta, zero_pr=pr,
               zero_n_theta=nt, zero_n_eta=ne, zero_n_pr=npr, zero_pr_norm=norm(npr))
    # (ii) duplicate (user, item) rows, also with different ratings; a rating id (1) with one row
    data = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 2], [1, 1, 1], [1, 0, 0], [1, 0, 0], [0, 1, 2]],
                    dtype=np.int64)
    theta = rng.random((2, 4)); eta = rng.random((2, 3)); pr = norm(rng.random((4, 3, 3)))
    nt, ne, npr = upd(data, theta, eta, pr)
    out.update(dup_data=data, dup_theta=theta, dup_eta=eta, dup_pr=pr,
               dup_n_theta=nt, dup_n_eta=ne, dup_n_pr=npr,
               dup_prod_dist=prd(data, theta, eta, pr), dup_omegas=com(data, theta, eta, pr))
    # (iii) s_n far below eps -> max(s, eps) branch (cupy/numba's s + eps would differ)
    data = np.array([[0, 0, 0], [1, 1, 1], [0, 1, 1], [1, 0, 0]], dtype=np.int64)
    theta = rng.random((2, 2)) * 1e-110; eta = rng.random((2, 2)) * 1e-110
    pr = rng.random((2, 2, 2)) * 1e-110
    nt, ne, npr = upd(data, theta, eta, pr)
    mm = prepared(data, 2, 2, seed=0)
    out.update(tiny_data=data, tiny_theta=theta, tiny_eta=eta, tiny_pr=pr,
               tiny_n_theta=nt, tiny_n_eta=ne, tiny_n_pr=npr,
               tiny_likelihood=np.float64(mm.em.compute_likelihood(data, theta, eta, pr)))
    # (iv) mixed: one row underflows, the others do not
    theta = rng.random((2, 2)); theta[0] *= 1e-200
    eta = rng.random((2, 2)); eta[0] *= 1e-150
    pr = norm(rng.random((2, 2, 2)))
    nt, ne, npr = upd(data, theta, eta, pr)
    out.update(mix_data=data, mix_theta=theta, mix_eta=eta, mix_pr=pr,
               mix_n_theta=nt, mix_n_eta=ne, mix_n_pr=npr,
               mix_likelihood=np.float64(mm.em.compute_likelihood(data, theta, eta, pr)))
    save("edge_cases", **out)


if __name__ == "__main__" and not os.environ.get("MMSBM_GOLDEN_ONLY"):
    g0(); g1_g3(); g4(); g5(); edge()


def g6_cv():
    """The reference's cv_fit test case (tests/test_mmsbm.py:30-34,57-61): folds=2 on mock_data(1)."""
    df1 = mock_frame(1)
    mm = MMSBM(2, 2, iterations=10, seed=1, backend="numpy")
    acc = mm.cv_fit(df1.copy(), folds=2)
    out = {"accuracies": np.array(acc, dtype=np.float64)}
    out.update(frame_cols(df1, "raw"))
    out["best_prediction_matrix"] = mm.prediction_matrix
    save("g6_cv_fit", **out)


if __name__ == "__main__" and os.environ.get("MMSBM_GOLDEN_ONLY", "g6") == "g6":
    g6_cv()


if __name__ == "__main__" and os.environ.get("MMSBM_GOLDEN_ONLY", "g5long") == "g5long":
    g5_long()


if __name__ == "__main__" and os.environ.get("MMSBM_GOLDEN_ONLY", "g7") == "g7":
    g7_uneven()


# ------------------------------------------------------------------------------------------------------------
# The two LONG fixtures (opt-in: MMSBM_GOLDEN_ONLY=g8 / =g9; never part of the default run, which stays ~2 min)
# ------------------------------------------------------------------------------------------------------------
def direct_problem(n, u, i, r, k, l, gen_seed=0, model_seed=0):
    """A uniform problem with the EM object built directly (the reference's _prepare_objects is O(U*N) dead work
    at these sizes, src/mmsbm.py:100-122); degrees by bincount, identical values."""
    train = uniform_triples(n, u, i, r, gen_seed)
    mm = MMSBM(k, l, iterations=1, seed=model_seed, backend="numpy")
    d_u = np.maximum(np.bincount(train[:, 0]), 1); d_i = np.maximum(np.bincount(train[:, 1]), 1)
    mm.p, mm.m = int(train[:, 0].max()), int(train[:, 1].max())
    mm._dims = {"n_samples": len(train), "n_user_groups": k, "n_item_groups": l,
                "n_ratings": int(train[:, 2].max()) + 1}
    mm.em = ExpectationMaximization(
        dims=mm._dims, user_indices=None, item_indices=None, rating_indices=None,
        norm_factors={"user": np.repeat(d_u[:, None], k, 1), "item": np.repeat(d_i[:, None], l, 1)},
        backend="numpy")
    mm.train = train
    return mm, train


def long_run(name, n, u, i, r, k, l, snaps, n_pick):
    """The reference's own EM functions over its DEFAULT run length on one restart (model seed 0, restart 0):
    at each snapshot the sampled theta / eta entries, column sums, all of p, the likelihood, the argmax prediction
    of every training row and the bit 'top-2 gap > 1e-9'.  Progress goes to stderr (the big one runs for over an hour)."""
    import time
    mm, train = direct_problem(n, u, i, r, k, l)
    pick = np.random.default_rng(7)
    ut = pick.integers(0, mm.p + 1, n_pick); kt = pick.integers(0, k, n_pick)
    ie = pick.integers(0, mm.m + 1, n_pick); le = pick.integers(0, l, n_pick)
    out = {"n": n, "u": u, "i": i, "r": r, "k": k, "l": l, "gen_seed": 0, "model_seed": 0,
           "train_sum": train.sum(axis=0), "train_head": train[:64], "ut": ut, "kt": kt, "ie": ie, "le": le,
           "snapshots": np.array(snaps)}
    rng = np.random.default_rng(mm.child_states[0])
    theta = mm.em.normalize_with_d(rng.random((mm.p + 1, k)), "user")
    eta = mm.em.normalize_with_d(rng.random((mm.m + 1, l)), "item")
    pr = mm.em.normalize_with_self(rng.random((k, l, mm._dims["n_ratings"])))
    out["theta_s_0"] = theta[ut, kt]; out["eta_s_0"] = eta[ie, le]; out["pr_0"] = pr
    liks, t0 = [], time.time()
    for j in range(max(snaps)):
        n_theta, n_eta, n_pr = mm.em.update_coefficients(data=train, theta=theta, eta=eta, pr=pr)
        theta = mm.em.normalize_with_d(n_theta, "user")
        eta = mm.em.normalize_with_d(n_eta, "item")
        pr = mm.em.normalize_with_self(n_pr)
        it = j + 1
        if it % 10 == 0:
            print(f"{name}: iteration {it}  {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        if it in snaps:
            out[f"theta_s_{it}"] = theta[ut, kt]; out[f"eta_s_{it}"] = eta[ie, le]; out[f"pr_{it}"] = pr
            out[f"theta_colsum_{it}"] = theta.sum(axis=0); out[f"eta_colsum_{it}"] = eta.sum(axis=0)
            pdist = ref_k.prod_dist(train, theta, eta, pr)
            srt = np.sort(pdist, axis=1)
            out[f"argmax_{it}"] = np.argmax(pdist, 1).astype(np.int8)
            out[f"clear_{it}"] = np.packbits((srt[:, -1] - srt[:, -2]) > 1e-9)
            liks.append(mm.em.compute_likelihood(train, theta, eta, pr))
            out["likelihood_at"] = np.array(liks)
            np.savez_compressed(os.path.join(os.environ.get("TMPDIR", "/tmp"), name + "_partial.npz"), **out)
    out["likelihood_at"] = np.array(liks)
    save(name, **out)


def g8_c3_long():
    """C3 itself -- BASELINE.json's headline config: 1M ratings, 100k x 20k, R=5, K=L=20, seed 0 -- over the reference's
    default iterations=400 (src/mmsbm.py:63-72), snapshots at 100 / 200 / 400.  About 11 s per iteration on one core,
    7 GB resident: ~75 minutes."""
    long_run("g8_c3_400", 1_000_000, 100_000, 20_000, 5, 20, 20, (100, 200, 400), 4000)


def g9_k50_long():
    """A C5-shaped problem the dense reference can hold: 100k ratings of 10k users x 1k items (C5's 10 ratings per
    user, 100 per item), R=10, K=L=50 -- the matrix-core kernel family -- 400 iterations, snapshots at 100 / 200 / 400."""
    long_run("g9_k50_400", 100_000, 10_000, 1_000, 10, 50, 50, (100, 200, 400), 4000)


def g10_k80_long():
    """The BLOCKED matrix-core kernels (a side beyond 64 groups: mfma_rows_kernel + mfma_slab_kernel): 40k ratings of 4k users
    x 400 items, R = 8, K = L = 80, 200 iterations, snapshots at 50 / 100 / 200.  About 20 minutes."""
    long_run("g10_k80_200", 40_000, 4_000, 400, 8, 80, 80, (50, 100, 200), 4000)


if __name__ == "__main__" and os.environ.get("MMSBM_GOLDEN_ONLY") == "g10":
    g10_k80_long()

if __name__ == "__main__" and os.environ.get("MMSBM_GOLDEN_ONLY") == "g8":
    g8_c3_long()

if __name__ == "__main__" and os.environ.get("MMSBM_GOLDEN_ONLY") == "g9":
    g9_k50_long()

if __name__ == "__main__" and os.environ.get("MMSBM_GOLDEN_ONLY") == "glong_selftest":
    long_run("glong_selftest", 5_000, 500, 100, 5, 6, 7, (2, 4, 6), 50)

"""Parity of the HIP path (through the C ABI) with the oracle and the reference-generated
golden vectors.  Needs a real MI355X: run with ``-m gpu``."""
import numpy as np
import pytest

from conftest import assert_elementwise, load_golden, rel_err
from oracle import mmsbm_oracle as orc

pytestmark = pytest.mark.gpu

# Tolerances (float64 everywhere; the HIP path re-associates the sums):
TOL_STEP = 1e-12   # one update_coefficients call, relative to max |want|
TOL_LOOP = 1e-9    # after tens..hundreds of EM iterations (north_star bar: 1e-5)


@pytest.fixture(scope="module")
def hip():
    from mmsbm_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("-m gpu tests need a GPU: no HIP device visible (no CPU fallback exists)")
    import mmsbm_amd
    return mmsbm_amd


def make_ctx(hip, data, theta, eta, pr, **kw):
    em = hip.HipEM(data, theta.shape[1], eta.shape[1], theta.shape[0], eta.shape[0], pr.shape[2], **kw)
    em.set_params(theta, eta, pr)
    return em


def test_reference_backend_test_vectors(hip):
    """The inputs of the reference's tests/test_backends.py, at its tolerance (atol 1e-8) and ours."""
    g = load_golden("g0_backend_tests")
    from mmsbm_amd import kernels_hip
    for tag in "ab":
        args = (g["data"], g[f"{tag}_theta"], g[f"{tag}_eta"], g[f"{tag}_pr"])
        om = kernels_hip.compute_omegas(*args)
        assert om.shape == (3, 2, 2) and np.allclose(om, g[f"{tag}_omegas"], atol=1e-8)
        assert np.array_equal(om, g[f"{tag}_omegas"])  # same association order -> bit exact
        pdist = kernels_hip.prod_dist(*args)
        assert np.allclose(pdist, g[f"{tag}_prod_dist"], atol=1e-8)
        assert np.allclose(pdist, g[f"{tag}_prod_dist"], rtol=1e-14, atol=1e-16)
        for got, nm in zip(kernels_hip.update_coefficients(*args), ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, g[f"{tag}_{nm}"]) < TOL_STEP, nm
    kernels_hip.clear_cache()


def test_level1_contract_semantics(hip):
    from mmsbm_amd import kernels_hip, load_backend
    fns = load_backend("hip")
    assert fns[3] == "hip" and fns[1] is kernels_hip.update_coefficients
    assert load_backend("auto")[3] == "hip"
    g = load_golden("g4_2k_k10")
    data = g["train"]
    theta, eta, pr = g["theta_0"].copy(), g["eta_0"].copy(), g["pr_0"].copy()
    keep = [a.copy() for a in (data, theta, eta, pr)]
    strided = np.asfortranarray(data)  # the reference passes whatever pandas gave it
    out1 = kernels_hip.update_coefficients(strided, theta, eta, pr)
    out2 = kernels_hip.update_coefficients(data, theta, eta, pr)
    for a, b, nm in zip(out1, out2, ("n_theta", "n_eta", "n_pr")):
        assert np.array_equal(a, b)  # bitwise reproducible, cached context or not
        assert a.flags.owndata or a.base is None
        assert rel_err(a, g[f"{nm}_1"]) < TOL_STEP
    for a, b in zip((data, theta, eta, pr), keep):
        assert np.array_equal(a, b)  # inputs never mutated
    with pytest.raises(ValueError):
        kernels_hip.update_coefficients(data, theta, eta, pr[:, :-1])
    kernels_hip.clear_cache()


@pytest.mark.parametrize("tag", ["zero", "dup", "tiny", "mix"])
def test_edge_cases(hip, tag):
    g = load_golden("edge_cases")
    data, theta, eta, pr = (g[f"{tag}_{x}"] for x in ("data", "theta", "eta", "pr"))
    with make_ctx(hip, data, theta, eta, pr) as em:
        for got, nm in zip(em.update_coefficients(), ("n_theta", "n_eta", "n_pr")):
            assert np.allclose(got, g[f"{tag}_{nm}"], rtol=1e-12, atol=1e-300), nm
        if tag == "zero":  # zero-row guard of normalize_with_self
            em.iterate(1)
            p1 = em.get_params()[2]
            assert np.allclose(p1, g["zero_pr_norm"], rtol=1e-12, atol=0)
            assert np.all(p1[1] == 0)
        if tag in ("tiny", "mix"):
            lik = em.likelihood()
            assert lik == pytest.approx(float(g[f"{tag}_likelihood"]), rel=1e-12)
        if tag == "dup":
            assert np.array_equal(em.compute_omegas(), g["dup_omegas"])
            assert np.allclose(em.prod_dist(data), g["dup_prod_dist"], rtol=1e-13, atol=1e-16)


def test_c1_500_iterations_vs_reference(hip):
    """BASELINE config 0: 100 ratings, K=2, L=4, 500 iterations, seed 1."""
    g = load_golden("g1_c1_mock")
    train = g["train"]
    with make_ctx(hip, train, g["c1_theta_0"], g["c1_eta_0"], g["c1_pr_0"]) as em:
        d_u, d_i = em.degrees()
        assert np.array_equal(d_u, g["d_u"]) and np.array_equal(d_i, g["d_i"])
        assert np.array_equal(em.compute_omegas(), g["c1_omegas_0"])
        assert em.likelihood() == pytest.approx(float(g["c1_likelihood_at"][0]), rel=1e-12)
        for got, nm in zip(em.update_coefficients(), ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, g[f"c1_{nm}_1"]) < TOL_STEP
            assert_elementwise(got, g[f"c1_{nm}_1"], nm, rtol=1e-12)
        done = 0
        for it in (1, 10, 500):
            em.iterate(it - done)
            done = it
            for got, nm in zip(em.get_params(), ("theta", "eta", "pr")):
                assert rel_err(got, g[f"c1_{nm}_{it}"]) < TOL_LOOP, (it, nm)
                # every entry on its own scale (p entries of this run go down to 1e-251):
                # max-norm alone would not see an entry of 1e-20 that is off by a factor of ten
                assert_elementwise(got, g[f"c1_{nm}_{it}"], f"{nm} after {it} iterations")
        assert em.likelihood() == pytest.approx(-9.470339454833308, rel=1e-9)


def test_g4_50_iterations_and_swapped_sides(hip):
    g = load_golden("g4_2k_k10")
    outs = []
    for swap in (0, 1):
        with make_ctx(hip, g["train"], g["theta_0"], g["eta_0"], g["pr_0"], swap_sides=swap) as em:
            assert em.swapped == bool(swap)
            for got, nm in zip(em.update_coefficients(), ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, g[f"{nm}_1"]) < TOL_STEP, (swap, nm)
            em.iterate(50)
            params = em.get_params()
            for got, nm in zip(params, ("theta", "eta", "pr")):
                assert rel_err(got, g[f"{nm}_50"]) < TOL_LOOP, (swap, nm)
            assert em.likelihood() == pytest.approx(float(g["likelihood_50"]), rel=1e-10)
            pdist = em.prod_dist(g["train"])
            assert np.allclose(pdist, g["prod_dist_50"], rtol=1e-9, atol=1e-14)
            outs.append(params)
            if swap:
                om = em.compute_omegas()
                assert np.array_equal(om, orc.compute_omegas(g["train"], *params))


@pytest.mark.parametrize("k,l,r", [(1, 1, 2), (3, 5, 4), (7, 2, 3), (16, 17, 5), (33, 50, 6),
                                   (70, 9, 3), (130, 20, 2), (20, 140, 2), (260, 6, 2)])
def test_padded_and_wide_group_shapes(hip, k, l, r):
    """Odd K/L (zero padding), every (lanes-per-row, vector width) instantiation."""
    rng = np.random.default_rng(k * 1000 + l)
    data = orc.synthetic_triples(1500, 120, 60, r, seed=k + l)
    n_u, n_i, n_r = (int(data[:, j].max()) + 1 for j in range(3))
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(rng.integers(1 << 30), n_u, n_i, n_r, k, l, d_u, d_i)
    with make_ctx(hip, data, theta, eta, pr) as em:
        want = orc.update_coefficients(data, theta, eta, pr)
        for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, w) < TOL_STEP, nm
        em.iterate(3)
        for _ in range(3):
            theta, eta, pr = orc.em_step(data, theta, eta, pr, d_u, d_i)
        for got, w, nm in zip(em.get_params(), (theta, eta, pr), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, nm
        assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, theta, eta, pr)), rel=1e-11)
        assert np.allclose(em.prod_dist(data[:100]), orc.prod_dist(data[:100], theta, eta, pr),
                           rtol=1e-11, atol=1e-15)


def test_skewed_degrees_and_empty_ids(hip):
    """One user holds a third of the rows; some users / items / ratings have no rows at all."""
    rng = np.random.default_rng(5)
    n = 6000
    u = np.where(rng.random(n) < 0.33, 7, rng.integers(0, 400, n))
    i = np.where(rng.random(n) < 0.2, 3, rng.integers(0, 50, n))
    r = rng.integers(0, 4, n)
    r[r == 2] = 3  # rating 2 never observed
    data = np.stack([u, i, r], axis=1).astype(np.int64)
    n_u, n_i, n_r, k, l = 410, 55, 5, 6, 12  # ids 400..409 / 50..54 / rating 4 unused
    theta = rng.random((n_u, k)); eta = rng.random((n_i, l))
    pr = orc.normalize_with_self(rng.random((k, l, n_r)))
    with make_ctx(hip, data, theta, eta, pr) as em:
        want = orc.update_coefficients(data, theta, eta, pr)
        for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, w) < TOL_STEP, nm
        d_u, d_i = em.degrees()
        assert d_u[405] == 1 and d_u[7] == np.sum(u == 7)
        em.iterate(2)
        du = np.maximum(np.bincount(u, minlength=n_u), 1); di = np.maximum(np.bincount(i, minlength=n_i), 1)
        for _ in range(2):
            theta, eta, pr = orc.em_step(data, theta, eta, pr, du, di)
        for got, w, nm in zip(em.get_params(), (theta, eta, pr), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, nm


def test_bitwise_reproducible(hip):
    g = load_golden("g4_2k_k10")
    runs = []
    for _ in range(2):
        with make_ctx(hip, g["train"], g["theta_0"], g["eta_0"], g["pr_0"]) as em:
            em.iterate(20)
            runs.append(em.get_params() + (em.likelihood(),))
    for a, b in zip(*runs):
        assert np.array_equal(a, b)


def test_host_class_reference_end_to_end_case(hip):
    """MMSBM(2, 2, iterations=10, seed=1) on mock_data(1) / mock_data(2): the numbers the
    reference's tests/test_mmsbm.py:53-102 assert, and the exact reference outputs."""
    g = load_golden("g1_c1_mock")
    import pandas as pd
    def frame(prefix):
        return pd.DataFrame({"users": g[prefix + "_users"], "items": g[prefix + "_items"],
                             "ratings": g[prefix + "_ratings"]})
    mm = hip.MMSBM(2, 2, iterations=10, seed=1, backend="hip")
    mm.fit(frame("train_raw"), silent=True)
    assert mm._backend == "hip"
    res = mm.results[0]
    assert set(res) == {"likelihood", "pr", "theta", "eta"}
    for nm in ("theta", "eta", "pr"):
        assert rel_err(res[nm], g[f"t_{nm}"]) < TOL_LOOP, nm
    assert float(res["likelihood"]) == pytest.approx(-13.773187406968459, rel=1e-9)
    pm = mm.predict(frame("test_raw"))
    assert pm.sum() == pytest.approx(100, 0.01)
    assert np.allclose(pm, g["t_prediction_matrix"], rtol=1e-8, atol=1e-12)
    assert np.array_equal(np.argmax(pm, 1), g["t_argmax"])  # identical argmax predictions
    sc = mm.score(silent=True)
    want = dict(zip(g["t_stats_keys"].tolist(), g["t_stats_vals"].tolist()))
    assert sc["stats"]["accuracy"] == pytest.approx(0.13, 0.01)
    assert sc["stats"]["one_off_accuracy"] == pytest.approx(0.55, 0.01)
    assert sc["stats"]["mae"] == pytest.approx(0.78, 0.01)
    assert sc["stats"]["s2"] == want["s2"] == 153
    assert sc["stats"]["s2pond"] == pytest.approx(129.4766730930339, rel=1e-9)
    assert sc["objects"]["theta"].sum(axis=0)[0] == pytest.approx(2.11, 0.1)
    assert sc["objects"]["eta"].sum(axis=0)[0] == pytest.approx(5.93, 0.1)
    assert set(sc["objects"]["pr"].keys()) == {"1", "2", "3", "4", "5"}


def test_host_class_on_a_rating_table_with_string_ids_and_uneven_degrees(hip):
    """The reference's front door on data shaped like a real rating table (src/data_handler.py:27-61 into the loop at
    src/mmsbm.py:243-256): string user / item ids, a few busy users, popular items -- long segments on both sides, which
    the library cuts into pieces and (round 4) still runs as two launches per iteration.  Against what the REAL
    reference gave for the same frames (tests/golden/g7_uneven_strings.npz, make_golden.py: g7_uneven; K = 6, L = 7, 40
    iterations, sampling = 2, seed = 3): the encoding, every restart's theta / eta / pr and likelihood, the prediction
    matrix with its argmax, the scores; and the context really ran the whole-segment form."""
    import pandas as pd
    g = load_golden("g7_uneven_strings")

    def frame(prefix, index=None):
        return pd.DataFrame({"users": g[prefix + "_users"], "items": g[prefix + "_items"],
                             "ratings": g[prefix + "_ratings"].astype(np.int64)}, index=index)
    mm = hip.MMSBM(6, 7, iterations=40, sampling=2, seed=3, backend="hip")
    mm.fit(frame("train_raw"), silent=True)
    assert np.array_equal(mm.train, g["train"])                      # the reference's DataHandler encoding
    ctx = mm._ctx(0)
    assert ctx.get_option("splits_pairs") > 0 and ctx.get_option("splits_users") > 0
    assert ctx.get_option("fused_split") == 3.0 and ctx.get_option("launches") == 2.0
    for s_, got in enumerate(mm.results):
        for nm in ("theta", "eta", "pr"):
            assert rel_err(got[nm], g[f"{nm}_{s_}"]) < TOL_LOOP, (s_, nm)
            assert_elementwise(got[nm], g[f"{nm}_{s_}"], f"{nm} of restart {s_}")
        assert float(got["likelihood"]) == pytest.approx(float(g["likelihoods"][s_]), rel=1e-10)
    pm = mm.predict(frame("test_raw", index=g["test_index"]))
    assert np.array_equal(mm.test, g["test"])
    ref = g["prediction_matrix"]
    assert np.allclose(pm, ref, rtol=1e-9, atol=1e-300)
    srt = np.sort(ref, axis=1)
    clear = srt[:, -1] - srt[:, -2] > 1e-9
    assert clear.mean() > 0.99 and np.array_equal(np.argmax(pm, 1)[clear], np.argmax(ref, 1)[clear])
    want = dict(zip(g["stats_keys"].tolist(), g["stats_vals"].tolist()))
    st = mm.score(silent=True)["stats"]
    assert st["accuracy"] == pytest.approx(want["accuracy"], abs=1e-12) and st["s2"] == want["s2"]
    assert st["one_off_accuracy"] == pytest.approx(want["one_off_accuracy"], abs=1e-12)
    assert st["mae"] == pytest.approx(want["mae"], abs=1e-12) and st["s2pond"] == pytest.approx(want["s2pond"], rel=1e-9)
    mm._release()


def test_host_class_sampling3_matches_reference(hip):
    g = load_golden("g2_c1_sampling3")
    mm = hip.MMSBM(2, 2, iterations=10, sampling=3, seed=1, backend="auto")
    mm.fit_encoded(g["train"])
    liks = np.array([r["likelihood"] for r in mm.results])
    assert np.allclose(liks, g["likelihoods"], rtol=1e-9)
    assert mm.best_by_likelihood == int(np.argmax(g["likelihoods"]))
    for s in range(3):
        assert rel_err(mm.results[s]["theta"], g[f"theta_{s}"]) < TOL_LOOP
    one = hip.MMSBM(2, 2, iterations=10, sampling=1, seed=1).fit_encoded(g["train"])
    assert np.array_equal(one.results[0]["theta"], mm.results[0]["theta"])  # independent of sampling


def test_c2_50_iterations_sampled_entries_and_argmax(hip):
    """BASELINE config 1 (100k ratings, 10k x 5k, K=L=10) against entries sampled from the
    reference's run; argmax parity with the tie rule of SURVEY section 7.3 item 6."""
    g = load_golden("g5_c2_sampled")
    train = orc.synthetic_triples(int(g["n"]), int(g["u"]), int(g["i"]), int(g["r"]), int(g["gen_seed"]))
    assert np.array_equal(train[:64], g["train_head"])
    mm = hip.MMSBM(10, 10, iterations=50, seed=int(g["model_seed"]))
    mm._prepare_objects(train)
    ctx = mm._ctx(0)
    d_u, d_i = ctx.degrees()
    ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i))
    done = 0
    for it in (1, 10, 50):
        ctx.iterate(it - done)
        done = it
        t, e, p = ctx.get_params()
        assert rel_err(t[g["ut"], g["kt"]], g[f"theta_s_{it}"]) < TOL_LOOP, it
        assert rel_err(e[g["ie"], g["le"]], g[f"eta_s_{it}"]) < TOL_LOOP, it
        assert rel_err(p, g[f"pr_{it}"]) < TOL_LOOP, it
        assert_elementwise(t[g["ut"], g["kt"]], g[f"theta_s_{it}"], f"theta entries after {it}")
        assert_elementwise(e[g["ie"], g["le"]], g[f"eta_s_{it}"], f"eta entries after {it}")
        assert_elementwise(p, g[f"pr_{it}"], f"p after {it}")
        assert rel_err(t.sum(0), g[f"theta_colsum_{it}"]) < TOL_LOOP
        assert ctx.likelihood() == pytest.approx(float(g["likelihood_at"][(1, 10, 50).index(it)]), rel=1e-9)
    pdist = ctx.prod_dist(train)
    clear = g["gap_50"] > 1e-9
    assert np.array_equal(np.argmax(pdist, 1)[clear], g["argmax_50"][clear])
    assert clear.mean() > 0.99


def test_c2_400_iterations_the_references_default_run_length(hip):
    """the reference's DEFAULT run length (iterations=400, src/mmsbm.py:63-72) in the suite.
    C2 from the reference's own start, against snapshots of the reference's run after 100, 200 and 400 iterations
    (tests/golden/g5_c2_400.npz, made by make_golden.py: g5_long): sampled theta / eta entries and all of p
    ELEMENT-WISE to 1e-6 (north_star's bar: 1e-5; measured 1.6e-8 at 400 -- an entry that has decayed through 200
    orders of magnitude is a product of hundreds of ratios), max-norm 1e-9, likelihood 1e-9, and the argmax
    prediction of every training row whose top-2 gap in the reference exceeds 1e-9."""
    g = load_golden("g5_c2_400")
    train = orc.synthetic_triples(int(g["n"]), int(g["u"]), int(g["i"]), int(g["r"]), int(g["gen_seed"]))
    assert np.array_equal(train.sum(0), g["train_sum"])
    mm = hip.MMSBM(10, 10, iterations=400, seed=int(g["model_seed"]))
    mm._prepare_objects(train)
    ctx = mm._ctx(0)
    d_u, d_i = ctx.degrees()
    ctx.set_params(*mm.init_params(mm.child_states[0], d_u, d_i))
    snaps = [int(x) for x in g["snapshots"]]
    assert snaps == [100, 200, 400]
    done = 0
    for j, it in enumerate(snaps):
        ctx.iterate(it - done)
        done = it
        t, e, p = ctx.get_params()
        for got, want, nm in ((t[g["ut"], g["kt"]], g[f"theta_s_{it}"], "theta entries"),
                              (e[g["ie"], g["le"]], g[f"eta_s_{it}"], "eta entries"), (p, g[f"pr_{it}"], "p")):
            assert rel_err(got, want) < TOL_LOOP, (nm, it)
            assert_elementwise(got, want, f"{nm} after {it} iterations", rtol=1e-6)
        assert rel_err(t.sum(0), g[f"theta_colsum_{it}"]) < TOL_LOOP and rel_err(e.sum(0), g[f"eta_colsum_{it}"]) < TOL_LOOP
        assert ctx.likelihood() == pytest.approx(float(g["likelihood_at"][j]), rel=1e-9)
        clear = np.unpackbits(g[f"clear_{it}"])[:len(train)].astype(bool)
        assert clear.mean() > 0.99
        assert np.array_equal(np.argmax(ctx.prod_dist(train), 1)[clear], g[f"argmax_{it}"][clear]), it
    assert float(g["likelihood_at"][2]) == float(g["likelihood_400"])


def test_c3_full_size_invariants(hip):
    """BASELINE config 2 (1M ratings, 100k x 20k, R=5, K=L=20): size-independent properties.
    sum_kl inc = 1 per triple  =>  rows of n_theta sum to d_u, of n_eta to d_i, n_p sums to N;
    after an iteration theta / eta rows and p[k,l,:] sum to 1; a 3,000-triple slice of the
    SAME parameters agrees with the oracle on the rows it touches."""
    train = orc.synthetic_triples(1_000_000, 100_000, 20_000, 5, seed=0)
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    assert n_u == 99_997
    mm = hip.MMSBM(20, 20, iterations=3, seed=0)
    mm._prepare_objects(train)
    ctx = mm._ctx(0)
    d_u, d_i = ctx.degrees()
    assert np.array_equal(d_u, np.bincount(train[:, 0])) and np.array_equal(d_i, np.bincount(train[:, 1]))
    theta, eta, pr = mm.init_params(mm.child_states[0], d_u, d_i)
    ctx.set_params(theta, eta, pr)
    n_t, n_e, n_p = ctx.update_coefficients()
    assert np.allclose(n_t.sum(1), d_u, rtol=1e-12)
    assert np.allclose(n_e.sum(1), d_i, rtol=1e-12)
    assert n_p.sum() == pytest.approx(len(train), rel=1e-12)
    assert np.allclose(n_p.sum(axis=(0, 1)), np.bincount(train[:, 2]), rtol=1e-12)
    # oracle on one user block: all rows of users < 300 (n_theta of those users is complete)
    sub = train[train[:, 0] < 300]
    w_t, _, _ = orc.update_coefficients(sub, theta, eta, pr)
    assert rel_err(n_t[:300], w_t[:300]) < TOL_STEP
    ctx.iterate(3)
    t, e, p = ctx.get_params()
    assert np.allclose(t.sum(1), 1, atol=1e-13) and np.allclose(e.sum(1), 1, atol=1e-13)
    assert np.allclose(p.sum(2), 1, atol=1e-13)
    assert np.isfinite(ctx.likelihood())


def test_plugin_module_name_resolution(hip):
    """src/backend.py:21 does import_module('kernels_' + name): with mmsbm_amd/plugin on sys.path the
    name 'kernels_hip' must resolve to the HIP backend (what INTEGRATION.md level 1 relies on)."""
    import importlib, os, sys
    from conftest import ROOT
    plug = os.path.join(ROOT, "mmsbm_amd", "plugin")
    sys.path.insert(0, plug)
    try:
        mod = importlib.import_module("kernels_hip")
    finally:
        sys.path.remove(plug)
    assert mod.__all__ == ["compute_omegas", "update_coefficients", "prod_dist"]
    g = load_golden("g0_backend_tests")
    out = mod.compute_omegas(g["data"], g["a_theta"], g["a_eta"], g["a_pr"])
    assert np.array_equal(out, g["a_omegas"])
    from mmsbm_amd import kernels_hip
    assert mod.update_coefficients is kernels_hip.update_coefficients
    kernels_hip.clear_cache()


def test_c5_shape_small(hip):
    """K=L=50, R=10 (the C5 shape: 32-lane groups, 2 slots per thread in the slab phase) on a
    problem small enough for the dense oracle."""
    data = orc.synthetic_triples(4000, 300, 150, 10, seed=50)
    n_u, n_i, n_r = (int(data[:, j].max()) + 1 for j in range(3))
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(123, n_u, n_i, n_r, 50, 50, d_u, d_i)
    with make_ctx(hip, data, theta, eta, pr) as em:
        want = orc.update_coefficients(data, theta, eta, pr)
        for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, w) < TOL_STEP, nm
            assert_elementwise(got, w, nm, rtol=1e-12)
        em.iterate(5)
        for _ in range(5):
            theta, eta, pr = orc.em_step(data, theta, eta, pr, d_u, d_i)
        for got, w, nm in zip(em.get_params(), (theta, eta, pr), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, nm
            assert_elementwise(got, w, nm)
        assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, theta, eta, pr)), rel=1e-11)


@pytest.mark.parametrize("slots", [1, 3])
def test_launch_modes_give_identical_results(hip, slots):
    """hipGraph replay changes scheduling only: bitwise-identical output (with restart slots too)."""
    g = load_golden("g4_2k_k10")
    outs = []
    for use in (False, True):
        with make_ctx(hip, g["train"], g["theta_0"], g["eta_0"], g["pr_0"]) as em:
            if slots > 1:
                em.set_slots(slots)
                for s in range(slots):
                    em.select(s).set_params(g["theta_0"] * (1 + 0.01 * s), g["eta_0"], g["pr_0"])
            if use:
                em.set_graph_mode(1)
            em.iterate(7)
            outs.append([em.select(s).get_params() for s in range(slots)])
    for ra, rb in zip(*outs):
        for a, b in zip(ra, rb):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", ["c1", "g4", "k20", "k12x24", "ragged", "busy", "c2", "ml10", "ml20", "bigsplit", "pairsplit"])
def test_two_launch_iteration_is_bitwise_the_four_launch_one(hip, shape):
    """fused_small.hpp: for small problems an iteration is pairs_fused_kernel (A, the pair pass and T + S
    of a 64-pair unit in one workgroup, C never leaving LDS) + tail_fused_kernel (user pass || p_update ||
    item_sum).  Every output keeps its arithmetic and association order: numerators after one step and the
    parameters after 1, 2 and 9 iterations are bit for bit those of the four launches; restart slots, graph
    replay, the likelihood (which needs A of the NEW parameters) and a switch between the forms in mid-run."""
    if shape == "c1":
        g = load_golden("g1_c1_mock")
        data, start, k, l = g["train"], (g["c1_theta_0"], g["c1_eta_0"], g["c1_pr_0"]), 2, 4
    elif shape == "g4":
        g = load_golden("g4_2k_k10")
        data, start, k, l = g["train"], (g["theta_0"], g["eta_0"], g["pr_0"]), 10, 10
    else:
        rng = np.random.default_rng(len(shape))
        if shape == "c2":
            data, k, l = orc.synthetic_triples(100_000, 10_000, 5_000, 5, seed=0), 10, 10
        elif shape == "ragged":   # unequal rating counts (units of 1..64 pairs, an almost empty rating), absent ids
            n = 9_000
            r_col = np.minimum(rng.geometric(0.5, n) - 1, 5)
            i_col = rng.integers(0, 400, n) * 3 % 401
            u_col = rng.integers(0, 900, n) * 2
            data, k, l = np.stack([u_col, i_col, r_col], axis=1).astype(np.int64), 7, 13
        elif shape in ("ml10", "ml20"):   # the MovieLens-100k shape in small: few users with ~100 ratings each, popular
            n, n_uu, n_ii = 30_000, 300, 500   # items -- segments cut into pieces on BOTH sides, every piece of a
            pu, pi = rng.lognormal(0, 0.8, n_uu), rng.lognormal(0, 1.2, n_ii)   # segment inside one workgroup (round 4)
            data = np.stack([rng.choice(n_uu, n, p=pu / pu.sum()), rng.choice(n_ii, n, p=pi / pi.sum()), rng.integers(0, 5, n)],
                            axis=1).astype(np.int64)
            k, l = (10, 10) if shape == "ml10" else (20, 16)
        elif shape == "bigsplit":  # a user with 40 % of the rows (225 pieces: more than a workgroup has groups, strided
            n = 24_000             # combine order) and an (item, rating) pair with 12 % (45 pieces of 64)
            u_col = np.where(rng.random(n) < 0.4, 5, rng.integers(0, 900, n))
            hot = rng.random(n) < 0.12
            data = np.stack([u_col, np.where(hot, 2, rng.integers(0, 400, n)), np.where(hot, 1, rng.integers(0, 5, n))],
                            axis=1).astype(np.int64)
            k, l = 7, 13
        elif shape == "pairsplit":  # popular items among many ordinary users: only the PAIR side has cut segments
            n = 20_000
            pi = rng.lognormal(0, 2.0, 150)
            data = np.stack([rng.integers(0, 4_000, n), rng.choice(150, n, p=pi / pi.sum()), rng.integers(0, 3, n)],
                            axis=1).astype(np.int64)
            k, l = 10, 6
        elif shape == "busy":     # 20 users with 50 ratings each among 780 with ~11: segments of two steps of rows in flight
            u_col = np.concatenate([np.repeat(np.arange(20), 50), rng.integers(20, 800, 8_600)])
            data = np.stack([u_col, rng.integers(0, 300, u_col.size), rng.integers(0, 4, u_col.size)], axis=1).astype(np.int64)
            k, l = 9, 11
        else:
            data = orc.synthetic_triples(20_000, 2_000, 500, 4, seed=9)
            k, l = (20, 20) if shape == "k20" else (12, 24)
        n_u, n_i, n_r = (int(data[:, j].max()) + 1 for j in range(3))
        start = orc.init_params(5, n_u, n_i, n_r, k, l, *orc.degrees(data, n_u, n_i))
    n_u, n_i, n_r = start[0].shape[0], start[1].shape[0], start[2].shape[2]
    runs = {}
    for fused in (1, 0):
        with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=2) as em:
            if fused:
                assert em.get_option("fused") == 1.0     # the library's own choice for a problem of this size
                if shape in ("ml10", "ml20", "bigsplit"):
                    assert em.get_option("splits_pairs") > 0 and em.get_option("splits_users") > 0
                    assert em.get_option("fused_split") == 3.0 and em.get_option("launches") == 2.0
                if shape == "pairsplit":
                    assert em.get_option("splits_pairs") + em.get_option("splits_users") > 0
                    assert em.get_option("fused_split") in (1.0, 2.0) and em.get_option("launches") == 2.0
            em.set_option("fused", fused)
            em.select(0).set_params(*start)
            em.select(1).set_params(start[0] * 0.5 + 0.01, start[1], start[2])
            out = [em.select(0).update_coefficients()]
            for its in (1, 1, 7):
                em.iterate(its)
                out.append([em.select(s).get_params() for s in range(2)])
                out.append(em.select(1).likelihood())
            em.set_graph_mode(1)
            em.iterate(4)
            em.set_graph_mode(0)
            out.append([em.select(s).get_params() for s in range(2)])
            em.set_option("fused", 1 - fused)            # ... and on in the other form
            em.iterate(3)
            out.append([em.select(s).get_params() for s in range(2)])
            out.append(em.select(0).likelihood())
            runs[fused] = out

    def same(a, b):
        if isinstance(a, (list, tuple)):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                same(x, y)
        else:
            assert np.array_equal(np.asarray(a), np.asarray(b))
    same(runs[1], runs[0])
    # and right: the fused numerators against the oracle
    for got, want, nm in zip(runs[1][0], orc.update_coefficients(data, *start), ("n_theta", "n_eta", "n_pr")):
        assert rel_err(got, want) < TOL_STEP, nm


def test_graph_replay_keeps_track_of_whether_a_is_current(hip):
    """a hipGraph replay runs none of the host code of an iteration, so the note "atab[cur] holds A of
    the current parameters" has to be kept by the replay loop.  Two-launch form captured, a one-step
    update_coefficients in between (which refreshes A and says so), replays, then the four-launch form -- whose triple
    passes READ atab[cur] -- must recompute A: bitwise an eager four-launch run of the same iterations."""
    g = load_golden("g4_2k_k10")
    start = (g["theta_0"], g["eta_0"], g["pr_0"])
    outs = []
    for graph in (1, 0):
        with make_ctx(hip, g["train"], *start) as em:
            assert em.get_option("launches") == 2.0
            em.set_graph_mode(graph)
            em.iterate(4)                       # (graph: captured and replayed twice)
            em.update_coefficients()            # not committed: A of the current parameters is refreshed on its way
            em.iterate(4)                       # replays: A in atab[cur] is stale again afterwards
            em.set_graph_mode(0)
            em.set_option("fused", 0)
            em.iterate(3)                       # four launches: needs A of the current parameters
            outs.append((em.get_params(), em.likelihood()))
        with make_ctx(hip, g["train"], *start) as em:   # the same in a context that never ran the two-launch form
            em.set_option("fused", 0)
            em.iterate(11)
            outs.append((em.get_params(), em.likelihood()))
    for params, lik in outs[1:]:
        for a, b in zip(params, outs[0][0]):
            assert np.array_equal(a, b)
        assert lik == outs[0][1]


def test_two_launch_iteration_is_chosen_by_size_and_refused_where_it_does_not_apply(hip):
    big = orc.synthetic_triples(400_000, 30_000, 6_000, 5, seed=2)
    with hip.HipEM(big, 20, 20) as em:                   # ratings x slots x (K + L) beyond 14M: four launches ...
        assert em.get_option("fused") == 1.0 and em.get_option("launches") == 4.0
        em.set_option("fused", 1)                        # ... unless asked for
        assert em.get_option("launches") == 2.0
    with hip.HipEM(big, 10, 10, slots=2) as em:          # the size that counts is that of the launch: all its slots
        assert em.get_option("launches") == 4.0
        em.set_slots(1)
        assert em.get_option("launches") == 2.0
    small = orc.synthetic_triples(5_000, 500, 200, 5, seed=2)
    with hip.HipEM(small, 50, 50) as em:                 # tile beyond the scalar cache: not this kernel
        assert em.get_option("fused") == 0.0
        with pytest.raises(Exception, match="fused"):
            em.set_option("fused", 1)
    with hip.HipEM(small, 28, 8) as em:                  # rows of more than 24 groups
        assert em.get_option("fused") == 0.0
    rng = np.random.default_rng(0)                       # one very busy user: its segment is cut into work items
    n = 9_000
    skew = np.stack([np.where(rng.random(n) < 0.3, 3, rng.integers(0, 900, n)), rng.integers(0, 400, n),
                     rng.integers(0, 5, n)], axis=1).astype(np.int64)
    with hip.HipEM(skew, 10, 10) as em:                  # ... its pieces all sit in one workgroup of the tail launch (round 4)
        assert em.get_option("items_users") > 0 and em.get_option("splits_users") > 0
        assert em.get_option("fused") == 1.0 and em.get_option("fused_split") == 2.0 and em.get_option("launches") == 2.0
    hot = skew.copy()                                    # an (item, rating) pair of 4,500 triples: more than the 64 pieces a
    hot[:4500, 1:] = (7, 2)                              # 64-pair unit may hold -- this data keeps the separate launches
    hot[:4500, 0] = np.arange(4500) % 900
    with hip.HipEM(hot, 10, 10) as em:
        assert em.get_option("splits_pairs") > 0 and em.get_option("fused") == 0.0 and em.get_option("launches") == 4.0
        with pytest.raises(Exception, match="fused"):
            em.set_option("fused", 1)


def test_non_temporal_output_rows_change_nothing_but_the_cache_policy(hip):
    """T, A and theta' rows go out as non-temporal stores for one restart per launch and rows of up to 32 groups
    (stages.hpp: nt_on): the same bits as with plain stores, in the four-launch and in the two-launch form."""
    data = orc.synthetic_triples(30_000, 3_000, 700, 5, seed=3)
    for k, l, fused in ((20, 20, 0), (20, 20, 1), (10, 7, 1), (32, 9, 0)):
        runs = []
        for nt in (1, 0):
            with hip.HipEM(data, k, l) as em:
                assert em.get_option("nt_out") == 7.0    # T and A rows, theta' rows, own-row loads
                em.set_option("nt_out", 7 * nt)
                assert em.get_option("nt_out") == 7.0 * nt
                em.set_option("fused", fused)
                em.init_params(11)
                em.iterate(6)
                runs.append(em.get_params() + (em.likelihood(),))
        for a, b in zip(*runs):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    with hip.HipEM(data, 20, 20, slots=2) as em:     # several restarts per launch, or long rows: plain stores
        assert em.get_option("nt_out") == 0.0
    with hip.HipEM(data[:3000], 40, 8) as em:
        assert em.get_option("nt_out") == 0.0
    rng = np.random.default_rng(1)                   # heavy-tailed degrees (work lists): rows are re-used, plain stores
    skew = np.stack([(rng.zipf(1.3, 20_000) - 1) % 2_000, (rng.zipf(1.3, 20_000) - 1) % 500, rng.integers(0, 5, 20_000)],
                    axis=1).astype(np.int64)
    with hip.HipEM(skew, 20, 20) as em:
        assert em.get_option("items_users") > 0 and em.get_option("nt_out") == 0.0


def test_cv_fit_matches_reference(hip):
    """The reference's cv_fit test case (tests/test_mmsbm.py:30-34,57-61): folds=2, accuracies 0.125, 0.16."""
    import pandas as pd
    g = load_golden("g6_cv_fit")
    df = pd.DataFrame({"users": g["raw_users"], "items": g["raw_items"], "ratings": g["raw_ratings"]})
    mm = hip.MMSBM(2, 2, iterations=10, seed=1)
    acc = mm.cv_fit(df, folds=2)
    assert acc == pytest.approx(g["accuracies"].tolist(), rel=1e-9)
    assert acc[0] == pytest.approx(0.125, 0.01) and acc[1] == pytest.approx(0.16, 0.01)
    assert np.allclose(mm.prediction_matrix, g["best_prediction_matrix"], rtol=1e-8, atol=1e-12)


def test_c_abi_error_paths(hip):
    """Status codes + last_error instead of exceptions across the ABI; nothing silently succeeds."""
    from mmsbm_amd import _lib
    from mmsbm_amd.core import HipEM
    good = np.array([[0, 0, 0], [1, 1, 1], [1, 0, 1]], dtype=np.int64)
    with pytest.raises(_lib.HipLibraryError) as e:
        HipEM(np.array([[0, 0, 0], [7, 0, 0]]), 2, 2, n_users=2, n_items=1, n_ratings=1)
    assert e.value.code == _lib.E_INVALID and "out of range" in e.value.message
    with pytest.raises(_lib.HipLibraryError) as e:
        HipEM(good, 9000, 9000)                             # a rating tile of more than 2^26 entries
    assert e.value.code == _lib.E_UNSUPPORTED and "2^26" in e.value.message
    with HipEM(good, 1025, 2) as em:                        # beyond 64 lanes x 16 doubles per row: runs since round 3
        assert em.n_pairs == 3
    with HipEM(good, 200, 200) as em:                       # beyond the 64-pair LDS stage: wide-row kernels,
        assert em.get_option("wide") == 1.0                 # no longer refused
    with pytest.raises(_lib.HipLibraryError) as e:
        HipEM(good, 2, 2, device=99)
    assert e.value.code == _lib.E_INVALID
    with HipEM(good, 2, 3) as em:
        with pytest.raises(_lib.HipLibraryError) as e:
            em.iterate(1)                                   # no parameters yet
        assert e.value.code == _lib.E_INVALID and "set_params" in e.value.message
        with pytest.raises(ValueError):
            em.set_params(np.ones((2, 2)), np.ones((2, 2)), np.ones((2, 3, 2)))   # eta has L=2, not 3
        em.set_params(np.full((2, 2), 0.5), np.full((2, 3), 1 / 3), np.full((2, 3, 2), 0.5))
        with pytest.raises(_lib.HipLibraryError) as e:
            em.prod_dist(np.array([[0, 5]]))                # unknown item
        assert e.value.code == _lib.E_INVALID
        import ctypes as C
        buf = np.empty(4)
        rc = _lib.load().mmsbm_hip_compute_omegas(em._h, buf.ctypes.data_as(C.POINTER(C.c_double)), 4)
        assert rc == _lib.E_TOOLARGE and b"larger" in _lib.load().mmsbm_hip_last_error()
        assert em.prod_dist(np.zeros((0, 2), dtype=np.int64)).shape == (0, 2)
        em.iterate(0)
        assert np.isfinite(em.likelihood())


def test_heavy_tailed_degrees_split_segments(hip):
    """Zipf-like degrees (one user with ~40 % of the rows, one pair with ~10 %): the long segments
    run as work items + ordered combine; results still match the oracle and are reproducible."""
    rng = np.random.default_rng(21)
    n = 30000
    u = np.where(rng.random(n) < 0.4, 5, (rng.zipf(1.3, n) - 1) % 2000)
    i = np.where(rng.random(n) < 0.3, 2, rng.integers(0, 300, n))
    data = np.stack([u, i, rng.integers(0, 5, n)], axis=1).astype(np.int64)
    data[:, 0] = np.unique(data[:, 0], return_inverse=True)[1]
    data[:, 1] = np.unique(data[:, 1], return_inverse=True)[1]
    n_u, n_i, n_r, k, l = int(data[:, 0].max()) + 1, int(data[:, 1].max()) + 1, 5, 20, 12
    d_u, d_i = orc.degrees(data, n_u, n_i)
    assert d_u.max() > 5000
    theta, eta, pr = orc.init_params(77, n_u, n_i, n_r, k, l, d_u, d_i)
    outs = []
    for _ in range(2):
        with make_ctx(hip, data, theta, eta, pr) as em:
            want = orc.update_coefficients(data, theta, eta, pr)
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, nm
            em.iterate(4)
            outs.append(em.get_params())
    t, e, p = theta, eta, pr
    for _ in range(4):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for got, w, nm in zip(outs[0], (t, e, p), ("theta", "eta", "pr")):
        assert rel_err(got, w) < 1e-11, nm
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_c5_full_size_invariants(hip):
    """BASELINE config 4's shape on one GPU (10M ratings, 1M x 100k, R=10, K=L=50): same
    size-independent properties as C3, plus a user-block slice against the oracle."""
    train = orc.synthetic_triples(10_000_000, 1_000_000, 100_000, 10, seed=0)
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    mm = hip.MMSBM(50, 50, iterations=2, seed=0)
    mm._prepare_objects(train)
    ctx = mm._ctx(0)
    d_u, d_i = ctx.degrees()
    assert np.array_equal(d_u, np.bincount(train[:, 0])) and np.array_equal(d_i, np.bincount(train[:, 1]))
    theta, eta, pr = mm.init_params(mm.child_states[0], d_u, d_i)
    ctx.set_params(theta, eta, pr)
    n_t, n_e, n_p = ctx.update_coefficients()
    assert np.allclose(n_t.sum(1), d_u, rtol=1e-12)
    assert np.allclose(n_e.sum(1), d_i, rtol=1e-12)
    assert np.allclose(n_p.sum(axis=(0, 1)), np.bincount(train[:, 2]), rtol=1e-11)
    sub = train[train[:, 0] < 40]
    w_t, _, _ = orc.update_coefficients(sub, theta, eta, pr)
    assert rel_err(n_t[:40], w_t[:40]) < TOL_STEP
    ctx.iterate(2)
    t, e, p = ctx.get_params()
    assert np.allclose(t.sum(1), 1, atol=1e-13) and np.allclose(e.sum(1), 1, atol=1e-13)
    assert np.allclose(p.sum(2), 1, atol=1e-13)
    assert np.isfinite(ctx.likelihood())


def test_randomised_shapes_against_oracle(hip):
    """40 seeded random problems (K, L in 1..48, R in 1..12, uniform or skewed degrees, either side
    paired with the ratings): numerators after one step and parameters after 3 iterations."""
    rng = np.random.default_rng(2024)
    for trial in range(40):
        k, l, r = int(rng.integers(1, 49)), int(rng.integers(1, 49)), int(rng.integers(1, 13))
        n_u, n_i = int(rng.integers(1, 400)), int(rng.integers(1, 200))
        n = int(rng.integers(1, 4000))
        u = rng.integers(0, n_u, n); i = rng.integers(0, n_i, n)
        if trial % 4 == 1:
            u = np.where(rng.random(n) < 0.5, 0, u)
        if trial % 4 == 2:
            i = np.where(rng.random(n) < 0.5, n_i - 1, i)
        data = np.stack([u, i, rng.integers(0, r, n)], axis=1).astype(np.int64)
        theta = rng.random((n_u, k)); eta = rng.random((n_i, l))
        pr = orc.normalize_with_self(rng.random((k, l, r)))
        d_u = np.maximum(np.bincount(u, minlength=n_u), 1); d_i = np.maximum(np.bincount(i, minlength=n_i), 1)
        tag = (trial, k, l, r, n_u, n_i, n)
        with make_ctx(hip, data, theta, eta, pr, swap_sides=int(trial % 3 == 0)) as em:
            want = orc.update_coefficients(data, theta, eta, pr)
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, (tag, nm)
            em.iterate(3)
            t, e, p = theta, eta, pr
            for _ in range(3):
                t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
            for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-10, (tag, nm)
            lik, lik_o = em.likelihood(), float(orc.compute_likelihood(data, t, e, p))
            assert lik == pytest.approx(lik_o, rel=1e-10, abs=1e-12), tag


def test_plain_c_program_matches_python_path(hip, tmp_path):
    """examples/abi_demo.c (C99, no Python) and the ctypes path give the same numbers on the same
    LCG-generated problem."""
    import re, subprocess
    from test_abi_cpu import _build_demo
    exe = _build_demo(tmp_path)
    n, U, I, R, K, L, iters = 5000, 300, 120, 5, 6, 9, 25
    run = subprocess.run([str(exe)] + [str(x) for x in (n, U, I, R, K, L, iters)], capture_output=True,
                         text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    m = re.search(r"likelihood (\S+) max\|rowsum-1\| (\S+) checksum (\S+)", run.stdout)
    lik_c, worst, chk_c = float(m.group(1)), float(m.group(2)), float(m.group(3))
    state = 88172645463325252
    def lcg():
        nonlocal state
        state = (state * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return state >> 33
    trip = np.array([[lcg() % U, lcg() % I, lcg() % R] for _ in range(n)], dtype=np.int64)
    unit = lambda: (lcg() + 1.0) / 2147483649.0
    theta = np.array([unit() for _ in range(U * K)]).reshape(U, K)
    eta = np.array([unit() for _ in range(I * L)]).reshape(I, L)
    pr = np.array([unit() for _ in range(K * L * R)]).reshape(K, L, R)
    with hip.HipEM(trip, K, L, U, I, R) as em:
        em.set_params(theta, eta, pr)
        em.iterate(iters)
        t = em.get_params()[0]
        assert em.likelihood() == pytest.approx(lik_c, rel=1e-12)
        assert float(((np.arange(U) % 7 + 1)[:, None] * t).sum()) == pytest.approx(chk_c, rel=1e-12)
    assert worst < 1e-12


def test_degenerate_sizes(hip):
    """No triples at all, a single triple, and ids that never occur: defined results, no launch of
    an empty grid, same numbers as the oracle."""
    rng = np.random.default_rng(3)
    theta = rng.random((3, 2)); eta = rng.random((2, 3)); pr = orc.normalize_with_self(rng.random((2, 3, 2)))
    empty = np.zeros((0, 3), dtype=np.int64)
    with hip.HipEM(empty, 2, 3, n_users=3, n_items=2, n_ratings=2) as em:
        em.set_params(theta, eta, pr)
        for got in em.update_coefficients():
            assert not got.any()
        em.iterate(2)
        t, e, p = em.get_params()
        assert not t.any() and not e.any() and not p.any()      # 0/1 and the zero-row guard
        assert em.likelihood() == 0.0 and em.compute_omegas().shape == (0, 2, 3)
    one = np.array([[2, 1, 1]], dtype=np.int64)
    with hip.HipEM(one, 2, 3, n_users=3, n_items=2, n_ratings=2) as em:
        em.set_params(theta, eta, pr)
        want = orc.update_coefficients(one, theta, eta, pr)
        for got, w in zip(em.update_coefficients(), want):
            assert np.allclose(got, w, rtol=1e-13, atol=0)
        em.iterate(1)
        d_u, d_i = np.array([1, 1, 1]), np.array([1, 1])
        for got, w in zip(em.get_params(), orc.em_step(one, theta, eta, pr, d_u, d_i)):
            assert np.allclose(got, w, rtol=1e-13, atol=0)


def _slot_problem(skew):
    rng = np.random.default_rng(5 if skew else 6)
    n = 20000
    if skew:  # long segments: work items + ordered combine, per slot
        u = np.where(rng.random(n) < 0.3, 3, rng.integers(0, 1500, n))
        i = np.where(rng.random(n) < 0.2, 1, rng.integers(0, 200, n))
    else:
        u, i = rng.integers(0, 3000, n), rng.integers(0, 400, n)
    data = np.stack([u, i, rng.integers(0, 5, n)], axis=1).astype(np.int64)
    data[:, 0] = np.unique(data[:, 0], return_inverse=True)[1]
    data[:, 1] = np.unique(data[:, 1], return_inverse=True)[1]
    return data, int(data[:, 0].max()) + 1, int(data[:, 1].max()) + 1, 5


@pytest.mark.parametrize("skew,k,l", [(False, 20, 20), (True, 7, 13), (False, 50, 36)])
def test_restart_slots_are_independent_and_bit_identical(hip, skew, k, l):
    """N1 (README.md:188 TODO): S restarts advance in one set of launches.  Slot s must hold,
    bit for bit, what a one-slot context gives for the same start -- and match the oracle."""
    data, n_u, n_i, n_r = _slot_problem(skew)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    starts = [orc.init_params(100 + s, n_u, n_i, n_r, k, l, d_u, d_i) for s in range(3)]
    test = data[:500]
    single = []
    for st in starts:
        with hip.HipEM(data, k, l, n_u, n_i, n_r) as em:
            em.set_params(*st)
            num = em.update_coefficients()
            em.iterate(5)
            single.append((num, em.get_params(), em.likelihood(), em.prod_dist(test)))
    with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=3) as em:
        assert em.slots == 3 and em.selected == 0
        for s in (2, 0, 1):  # any order
            em.select(s).set_params(*starts[s])
        for s in range(3):
            for a, b in zip(em.select(s).update_coefficients(), single[s][0]):
                assert np.array_equal(a, b)
        em.iterate(3)
        em.iterate(2)
        for s in range(3):
            em.select(s)
            assert em.selected == s
            for a, b in zip(em.get_params(), single[s][1]):
                assert np.array_equal(a, b)
            assert em.likelihood() == single[s][2]
            assert np.array_equal(em.prod_dist(test), single[s][3])
        # and against the oracle
        t, e, p = starts[1]
        for _ in range(5):
            t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
        for got, w, nm in zip(em.select(1).get_params(), (t, e, p), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, nm
        # shrinking back to one slot gives a fresh, working context
        em.set_slots(1)
        em.set_params(*starts[0])
        em.iterate(5)
        for a, b in zip(em.get_params(), single[0][1]):
            assert np.array_equal(a, b)


def test_restart_slot_errors(hip):
    from mmsbm_amd import _lib
    data, n_u, n_i, n_r = _slot_problem(False)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    st = orc.init_params(1, n_u, n_i, n_r, 4, 4, d_u, d_i)
    with hip.HipEM(data, 4, 4, n_u, n_i, n_r, slots=2) as em:
        em.set_params(*st)
        with pytest.raises(_lib.HipLibraryError) as e:
            em.iterate(1)  # slot 1 has no parameters yet
        assert e.value.code == _lib.E_INVALID and "slot 1" in e.value.message
        with pytest.raises(_lib.HipLibraryError):
            em.select(1).likelihood()
        for bad in (-1, 2):
            with pytest.raises(_lib.HipLibraryError) as e:
                em.select(bad)
            assert e.value.code == _lib.E_INVALID
        with pytest.raises(_lib.HipLibraryError):
            em.set_slots(0)
        em.select(1).set_params(*st)
        em.iterate(2)
        a, b = em.select(0).get_params(), em.select(1).get_params()
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_host_class_batches_restarts_without_changing_them(hip):
    """fit() runs the restarts of a GPU as slots of one context; batch size must not matter."""
    g = load_golden("g4_2k_k10")
    runs = {}
    for per in (1, 2, 5):
        mm = hip.MMSBM(10, 10, iterations=12, sampling=5, seed=3, restarts_per_launch=per)
        mm.fit_encoded(g["train"])
        runs[per] = mm.results
        assert mm.best_by_likelihood == int(np.argmax([r["likelihood"] for r in mm.results]))
        mm._release()
    for per in (2, 5):
        for a, b in zip(runs[1], runs[per]):
            assert a["likelihood"] == b["likelihood"]
            for key in ("theta", "eta", "pr"):
                assert np.array_equal(a[key], b[key])
    want = orc.fit(g["train"], 10, 10, 12, 5, 3)
    for got, w in zip(runs[5], want):
        assert abs(got["likelihood"] - w["likelihood"]) < 1e-9 * abs(w["likelihood"])
        assert rel_err(got["theta"], w["theta"]) < TOL_LOOP


def _score_problem():
    rng = np.random.default_rng(11)
    n_u, n_i, n_r, k, l = 300, 120, 5, 6, 9
    data = np.stack([rng.integers(0, n_u, 6000), rng.integers(0, n_i, 6000),
                     rng.integers(0, n_r, 6000)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    test = np.stack([rng.integers(0, n_u, 3001), rng.integers(0, n_i, 3001),
                     rng.integers(0, n_r, 3001)], axis=1).astype(np.int64)
    return data, test, (n_u, n_i, n_r, k, l), d_u, d_i


@pytest.mark.parametrize("swap", [0, 1])
def test_device_predict_score_matches_host_formulas(hip, swap):
    """N2: prod_dist for every restart, mean over restarts, argmax and the indicators of
    src/mmsbm.py:488-539 evaluated on the device."""
    data, test, (n_u, n_i, n_r, k, l), d_u, d_i = _score_problem()
    weights = np.arange(n_r, dtype=np.float64)  # the reference's self.ratings (rating indices)
    with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=3, swap_sides=swap) as em:
        for s in range(3):
            em.select(s).set_params(*orc.init_params(40 + s, n_u, n_i, n_r, k, l, d_u, d_i))
        em.iterate(6)
        # a user with an all-zero membership row: its test rows have an all-zero distribution
        t, e, p = em.select(2).get_params()
        t[test[0, 0]] = 0.0
        em.set_params(t, e, p)
        rats = [em.select(s).prod_dist(test) for s in range(3)]
        assert (rats[2].sum(1) == 0).any()
        em.predict_begin(test, weights)
        per = [em.select(s).predict_add() for s in range(3)]
        mean, raw = em.predict_finish()
    assert np.array_equal(mean, np.array(rats).mean(axis=0))  # numpy adds the restarts in order too
    for rat, st in list(zip(rats, per)) + [(mean, raw)]:
        want = orc.score_stats(rat, test[:, 2], list(range(n_r)))
        got = hip.HipEM.final_stats(st)
        assert st[0] == (rat.sum(1) != 0).sum()
        for key in ("accuracy", "one_off_accuracy", "mae"):
            assert got[key] == want[key], key
        assert got["s2"] == want["s2"]
        assert abs(got["s2pond"] - want["s2pond"]) <= 1e-12 * want["s2pond"]


@pytest.mark.parametrize("k,l,n_i", [(3, 4, 40), (20, 20, 150), (50, 50, 90), (8, 70, 60), (200, 24, 50)])
def test_predict_through_the_item_rating_table(hip, k, l, n_i):
    """prod_dist / predict as P[m, r] = theta_u . (p_r eta_i): the inner vectors for every (item,
    rating) combination come from the A launch's mat-vec (lane-per-pair, matrix-core and wide-row
    forms), a test row then costs R dot products (predict_rows_kernel).  Same distributions as the
    per-row kernels (option predict_fast = 0) and the oracle, same indicator sums; taken only when
    the rows are not far fewer than the items."""
    rng = np.random.default_rng(k * 31 + l)
    n_u, n_r = 120, 4
    data = np.stack([rng.integers(0, n_u, 3000), rng.integers(0, n_i, 3000), rng.integers(0, n_r, 3000)], axis=1).astype(np.int64)
    test = np.stack([rng.integers(0, n_u, 1001), rng.integers(0, n_i, 1001), rng.integers(0, n_r, 1001)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    weights = np.arange(n_r, dtype=np.float64)
    for swap in (0, 1):
        with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=2, swap_sides=swap) as em:
            for s_ in range(2):
                em.select(s_).set_params(*orc.init_params(7 + s_, n_u, n_i, n_r, k, l, d_u, d_i))
            em.iterate(2)
            got = {}
            for fast in (1, 0):
                em.set_option("predict_fast", fast)
                assert em.get_option("predict_fast") == float(fast)
                rats = [em.select(s_).prod_dist(test) for s_ in range(2)]
                em.predict_begin(test, weights)
                per = [em.select(s_).predict_add() for s_ in range(2)]
                mean, raw = em.predict_finish()
                assert np.array_equal(mean, np.array(rats).mean(axis=0))
                got[fast] = (rats, per, mean, raw)
            em.set_option("predict_fast", 1)
            few = em.select(0).prod_dist(test[:3])           # 3 rows, many items: the per-row kernel either way
            ref = [orc.prod_dist(test, *em.select(s_).get_params()) for s_ in range(2)]
        for s_ in range(2):
            assert np.allclose(got[1][0][s_], ref[s_], rtol=1e-11, atol=1e-300)
            assert np.allclose(got[1][0][s_], got[0][0][s_], rtol=1e-12, atol=1e-300)
            assert np.array_equal(got[1][1][s_][:5], got[0][1][s_][:5])                     # counts: exact
            assert got[1][1][s_][5] == pytest.approx(got[0][1][s_][5], rel=1e-12)
        assert np.array_equal(got[1][3][:5], got[0][3][:5])
        assert np.allclose(few, ref[0][:3], rtol=1e-11, atol=1e-300)


def test_device_predict_session_errors(hip):
    from mmsbm_amd import _lib
    data, test, (n_u, n_i, n_r, k, l), d_u, d_i = _score_problem()
    with hip.HipEM(data, k, l, n_u, n_i, n_r) as em:
        em.set_params(*orc.init_params(1, n_u, n_i, n_r, k, l, d_u, d_i))
        with pytest.raises(_lib.HipLibraryError) as e:
            em.predict_add()
        assert e.value.code == _lib.E_INVALID and "predict_begin" in e.value.message
        em.predict_begin(test, np.arange(n_r))
        with pytest.raises(_lib.HipLibraryError):
            em.predict_finish()                      # nothing added yet
        bad = test.copy()
        bad[5, 2] = n_r
        with pytest.raises(_lib.HipLibraryError) as e:
            em.predict_begin(bad, np.arange(n_r))
        assert e.value.code == _lib.E_INVALID
        st = em.predict_add()                        # a rejected begin leaves the open session alone
        assert st[0] == len(test)
        with pytest.raises(ValueError):
            em.predict_begin(test, np.arange(n_r + 1))
        em.predict_begin(test[:0], np.arange(n_r))   # empty test set
        st = em.predict_add()
        mean, raw = em.predict_finish()
        assert mean.shape == (0, n_r) and not st.any() and not raw.any()


def test_host_predict_uses_resident_slots_or_uploads(hip):
    """predict() after fit() reads the restarts straight from the context's slots; after the
    slots were reused (or with results from elsewhere) it uploads -- same answer either way."""
    import pandas as pd
    rng = np.random.default_rng(3)
    df = pd.DataFrame({"u": rng.integers(0, 60, 1500), "i": rng.integers(0, 40, 1500),
                       "r": rng.integers(1, 6, 1500)})
    test = df.iloc[:400]
    mm = hip.MMSBM(3, 4, iterations=15, sampling=4, seed=2)
    mm.fit(df.iloc[400:], silent=True)
    assert mm._resident[(0, 0)] == [0, 1, 2, 3]
    a = mm.predict(test).copy()
    sa = mm.score(silent=True)["stats"]
    mm.compute_likelihood(mm.train, mm.results[0]["theta"], mm.results[0]["eta"], mm.results[0]["pr"])
    assert (0, 0) not in mm._resident
    b = mm.predict(test)
    sb = mm.score(silent=True)["stats"]
    assert np.array_equal(a, b) and sa == sb
    rats = []
    ctx = mm._ctx(0)
    for res in mm.results:
        ctx.set_params(res["theta"], res["eta"], res["pr"])
        rats.append(ctx.prod_dist(mm.test))
    assert np.array_equal(a, np.array(rats).mean(axis=0))
    host = mm._compute_stats(a)                      # the host-side formulas on the same matrix
    for key in ("accuracy", "one_off_accuracy", "mae", "s2"):
        assert sa[key] == host[key], key
    assert abs(sa["s2pond"] - host["s2pond"]) < 1e-12 * host["s2pond"]
    accs = [mm._compute_stats(r)["accuracy"] for r in rats]
    assert [s["accuracy"] for s in mm.run_stats] == accs


def test_cv_folds_run_concurrently_with_the_same_result(hip):
    """N4: with several entries in `devices` the folds are independent jobs on separate contexts
    (here two lanes on the one GPU); accuracies and the kept objects equal the sequential run."""
    import pandas as pd
    g = load_golden("g6_cv_fit")
    df = pd.DataFrame({"users": g["raw_users"], "items": g["raw_items"], "ratings": g["raw_ratings"]})
    seq = hip.MMSBM(2, 2, iterations=10, sampling=2, seed=1)
    acc_seq = seq.cv_fit(df, folds=3)
    par = hip.MMSBM(2, 2, iterations=10, sampling=2, seed=1, devices=[0, 0])
    acc_par = par.cv_fit(df, folds=3)
    assert acc_seq == acc_par
    assert np.array_equal(seq.prediction_matrix, par.prediction_matrix)
    assert seq.theta.equals(par.theta) and seq.eta.equals(par.eta)
    for a, b in zip(seq.cv_results, par.cv_results):
        assert a["stats"] == b["stats"]


def test_convergence_monitor_stops_early_and_matches_fixed_length_run(hip):
    """`tol` stops a batch once every restart's likelihood moved by less than tol (relative)
    between two checks; what it ran is exactly the fixed-length run of that many iterations."""
    g = load_golden("g4_2k_k10")
    mm = hip.MMSBM(10, 10, iterations=4000, sampling=2, seed=3, tol=1e-4, check_every=25)
    mm.fit_encoded(g["train"])
    ran = mm.iterations_run[0]
    assert ran == mm.iterations_run[1] and 50 <= ran < 4000 and ran % 25 == 0
    ref = hip.MMSBM(10, 10, iterations=ran, sampling=2, seed=3)
    ref.fit_encoded(g["train"])
    assert ref.iterations_run == {0: ran, 1: ran}
    for a, b in zip(mm.results, ref.results):
        assert a["likelihood"] == b["likelihood"] and np.array_equal(a["theta"], b["theta"])
    shorter = hip.MMSBM(10, 10, iterations=ran - 25, sampling=2, seed=3)
    shorter.fit_encoded(g["train"])
    for a, b in zip(mm.results, shorter.results):  # the stopping rule, checked from outside
        assert abs(a["likelihood"] - b["likelihood"]) <= 1e-4 * abs(b["likelihood"])


@pytest.mark.parametrize("k,l", [(1, 1), (3, 5), (7, 2), (16, 17), (20, 24), (33, 50), (20, 61),
                                 (6, 130), (5, 200)])
def test_likelihood_kernels_agree_with_the_reference_formula(hip, k, l):
    """src/expectation_maximization.py:157-167 through both device forms: a logarithm per element
    (the reference's order) and the logarithm tables (1..8 lanes per triple; rows wider than
    160 use the first form).  Zero memberships give -inf table entries that must never be used,
    tiny ones exercise the eps clamp."""
    rng = np.random.default_rng(k * 1000 + l)
    n_u, n_i, n_r = 90, 70, 4
    data = np.stack([rng.integers(0, n_u, 1500), rng.integers(0, n_i, 1500),
                     rng.integers(0, n_r, 1500)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(9, n_u, n_i, n_r, k, l, d_u, d_i)
    theta[::7, 0] = 0.0                      # exact zeros
    eta[::5, -1] = 1e-300                    # far below eps
    pr[0, 0, :] = 0.0                        # a zero tile entry for every rating
    want = float(orc.compute_likelihood(data, theta, eta, pr))
    for swap in (0, 1):
        with make_ctx(hip, data, theta, eta, pr, swap_sides=swap) as em:
            em.set_option("lik_fast", 0)
            slow = em.likelihood()
            em.set_option("lik_fast", 1)
            fast = em.likelihood()
            assert np.isfinite(fast)
            assert slow == pytest.approx(want, rel=1e-12)
            assert fast == pytest.approx(want, rel=1e-12)
            for g in (1, 2, 4, 8):
                em.set_option("lik_g", g)
                assert em.likelihood() == pytest.approx(want, rel=1e-12), g
            em.set_option("lik_g", 0)
            em.set_option("lik_fast", 2)       # the default: a wave per pair where rows have more than 32 groups
            assert em.get_option("lik_fast") == 2.0
            assert em.likelihood() == pytest.approx(want, rel=1e-12)
            for g in (1, 2, 4, 8):
                em.set_option("lik_g", g)
                assert em.likelihood() == pytest.approx(want, rel=1e-12), g


@pytest.mark.parametrize("k,l,stage", [(50, 50, "early"), (50, 50, "late"), (50, 50, "border"), (7, 70, "late"),
                                       (70, 9, "border"), (130, 40, "late"), (40, 100, "border"), (12, 150, "late"),
                                       (33, 64, "border"), (64, 33, "late"), (20, 20, "late"), (3, 5, "tiny"),
                                       (50, 50, "tiny")])
def test_likelihood_a_wave_per_pair(hip, k, l, stage):
    """lik_fact.hpp: for rows of more than 32 groups a wave takes one (item, rating) pair and walks its triples
    four at a time, theta through scalar loads, the clamp as two maxima, padding taken off in closed form
    (1, 2 or 3 columns per lane; either side paired with the rating; narrower rows keep the round-2 kernel).
    early = after one EM step (nothing clamped); late = concentrated memberships, most elements far below eps;
    border = products scattered a few decades either side of eps, exact zeros among them; tiny = every s_n
    below eps.  Against the dense oracle's src/expectation_maximization.py:157-167 at 1e-12 and against the
    two element-wise device forms."""
    rng = np.random.default_rng(k * 131 + l)
    n_u, n_i, n_r, n = 120, 80, 5, 4000
    data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(3, n_u, n_i, n_r, k, l, d_u, d_i)
    if stage == "early":
        theta, eta, pr = orc.em_step(data, theta, eta, pr, d_u, d_i)
    elif stage == "late":       # memberships concentrated on a few groups, the rest decayed towards zero
        theta = theta ** rng.integers(1, 40, theta.shape)
        eta = eta ** rng.integers(1, 40, eta.shape)
        theta /= theta.sum(1, keepdims=True); eta /= eta.sum(1, keepdims=True)
        pr = orc.normalize_with_self(pr ** rng.integers(1, 12, pr.shape))
    elif stage == "border":     # products theta eta p scattered over a few decades either side of eps
        theta = 10.0 ** rng.uniform(-9, -3, theta.shape)
        eta = 10.0 ** rng.uniform(-9, -3, eta.shape)
        theta[rng.random(theta.shape) < 0.1] = 0.0
        eta[rng.random(eta.shape) < 0.05] = 0.0
        pr[rng.random(pr.shape) < 0.05] = 0.0
    else:                       # tiny: s_n < eps for every triple
        theta, eta = theta * 1e-110, eta * 1e-110
    want = float(orc.compute_likelihood(data, theta, eta, pr))
    clamped_share = float(np.mean(orc.compute_omegas(data, theta, eta, pr) < orc.EPS))
    if stage == "early":
        assert clamped_share == 0.0
    elif stage in ("late", "border"):
        assert 0.05 < clamped_share < 0.999, clamped_share
    for swap in (0, 1):
        with make_ctx(hip, data, theta, eta, pr, swap_sides=swap) as em:
            assert em.get_option("lik_fast") == 2.0
            got = em.likelihood()
            assert got == pytest.approx(want, rel=1e-12), (stage, swap, clamped_share)
            em.set_option("lik_fast", 1)
            assert em.likelihood() == pytest.approx(got, rel=1e-12)
            em.set_option("lik_fast", 0)
            assert em.likelihood() == pytest.approx(got, rel=1e-12)
            em.set_option("lik_fast", 2)
            assert em.likelihood() == got      # bitwise reproducible
    # two restart slots with different parameters: the selected slot's tables are the ones that are read
    theta2, eta2, pr2 = orc.init_params(4, n_u, n_i, n_r, k, l, d_u, d_i)
    with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=2) as em:
        em.select(0).set_params(theta2, eta2, pr2)
        em.select(1).set_params(theta, eta, pr)
        assert em.select(1).likelihood() == pytest.approx(want, rel=1e-12)
        assert em.select(0).likelihood() == pytest.approx(float(orc.compute_likelihood(data, theta2, eta2, pr2)), rel=1e-12)


@pytest.mark.parametrize("k,l,swap", [(2, 4, 0), (20, 20, 0), (20, 20, 1), (7, 13, 1), (50, 36, 0), (1, 1, 0)])
def test_device_side_random_start_is_numpys(hip, k, l, swap):
    """a7 (src/mmsbm.py:224-233): theta0 = rng.random((U,K))/d_u, eta0 = rng.random((I,L))/d_i,
    p0 = normalize_with_self(rng.random((K,L,R))) from default_rng(child seed) -- drawn on the
    device from the same PCG64 stream, bit for bit, including ids that never occur (degree 1)."""
    rng = np.random.default_rng(77)
    n_u, n_i, n_r = 1234, 321, 6
    data = np.stack([rng.integers(0, n_u - 3, 9000), rng.integers(0, n_i - 2, 9000),
                     rng.integers(0, n_r, 9000)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    seeds = orc.child_seeds(5, 3)
    with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=swap, slots=3) as em:
        for s in (1, 0, 2):
            p0 = em.select(s).init_params(seeds[s])
            want = orc.init_params(seeds[s], n_u, n_i, n_r, k, l, d_u, d_i)
            assert np.array_equal(p0, want[2])
            for got, w, nm in zip(em.get_params(), want, ("theta", "eta", "pr")):
                assert np.array_equal(got, w), (s, nm)
        em.iterate(3)                                   # and the run that follows is the usual one
        got = em.select(2).get_params()
    with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=swap) as em:
        em.set_params(*orc.init_params(seeds[2], n_u, n_i, n_r, k, l, d_u, d_i))
        em.iterate(3)
        for a, b in zip(got, em.get_params()):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("k,l", [(64, 64), (60, 56), (72, 80), (120, 120), (150, 40), (9, 158)])
def test_wide_group_counts_beyond_the_lds_tile(hip, k, l):
    """K, L up to ~150: the rating tile no longer fits in LDS beside the rows and is read through
    scalar loads; likelihood falls back from the table form when ITS tiles do not fit."""
    rng = np.random.default_rng(k + l)
    n_u, n_i, n_r = 150, 90, 3
    data = np.stack([rng.integers(0, n_u, 1200), rng.integers(0, n_i, 1200),
                     rng.integers(0, n_r, 1200)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(3, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(2):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for swap in (0, 1):
        with make_ctx(hip, data, theta, eta, pr, swap_sides=swap) as em:
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, nm
            em.iterate(2)
            for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-11, nm
            assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)
            assert np.allclose(em.prod_dist(data[:50]), orc.prod_dist(data[:50], t, e, p), rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("k,l,n", [(1500, 3, 700), (3, 1500, 700), (1100, 20, 500), (2100, 2, 400), (2, 2500, 400),
                                   (1040, 1030, 60), (2048, 3, 300)])
def test_more_than_1024_groups_per_side(hip, k, l, n):
    """The reference's numpy backend has no size limit (src/kernels_numpy.py:21-79).  Rows beyond the widest
    group-of-lanes instantiation (64 lanes x 16 doubles = 1,024 groups) run since round 3: the triple passes with 32
    doubles per lane up to 2,048 groups and as seg_wide_kernel beyond (a wave per segment, the row in blocks of
    1,024 columns, weights first, then the columns),
    item_sum and the prediction rows with one more trip per 1,024 columns, the pair stage on the blocked
    matrix-core kernels.  Segments longer than a wave's 64 triples (a busy user / a popular pair) included; same
    checks as every other shape, either side paired with the rating, two restart slots."""
    rng = np.random.default_rng(k * 7 + l)
    n_u, n_i, n_r = 30, 12, 3
    u_col = np.where(rng.random(n) < 0.3, 4, rng.integers(0, n_u, n))      # user 4: ~ n / 3 triples (> 64)
    i_col = np.where(rng.random(n) < 0.3, 2, rng.integers(0, n_i, n))      # item 2 likewise
    data = np.stack([u_col, i_col, rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(5, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(2):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for swap in (0, 1):
        with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=2, swap_sides=swap) as em:
            em.select(1).set_params(theta * 0.5, eta, pr)
            em.select(0).set_params(theta, eta, pr)
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, (nm, rel_err(got, w))
                assert_elementwise(got, w, nm, rtol=1e-11)
            em.iterate(2)
            for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-11, nm
            assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)
            assert np.allclose(em.prod_dist(data[:50]), orc.prod_dist(data[:50], t, e, p), rtol=1e-11, atol=1e-300)
            st = em.select(1).get_params()
            seen = np.bincount(data[:, 0], minlength=n_u) > 0
            assert np.all(np.isfinite(st[0])) and np.allclose(st[0].sum(1)[seen], 1.0, atol=1e-12)


def test_more_than_65535_rating_values(hip):
    """R = 70,000 distinct rating values (most (item, rating) pairs hold one triple): no limit on R either.
    Checked against the factorised checker (pinned to the dense oracle in tests/test_oracle_golden.py): the dense
    oracle restates the reference's per-rating mask loop, which is O(R N)."""
    from oracle import mmsbm_factorised as fac
    rng = np.random.default_rng(5)
    n, n_u, n_i, n_r = 80_000, 500, 300, 70_000
    r_col = np.concatenate([np.arange(n_r), rng.integers(0, n_r, n - n_r)])
    data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), r_col], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(5, n_u, n_i, n_r, 2, 3, d_u, d_i)
    pairs = fac.Pairs(data, n_u, n_i, n_r)
    want = fac.update_coefficients(data, theta, eta, pr, pairs)
    with make_ctx(hip, data, theta, eta, pr) as em:
        for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, w) < TOL_STEP, nm
        em.iterate(2)
        t, e, p = theta, eta, pr
        for _ in range(2):
            t, e, p = fac.em_step(data, t, e, p, d_u, d_i, pairs)
        for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, nm
        assert em.likelihood() == pytest.approx(float(fac.compute_likelihood(data, t, e, p, pairs)), rel=1e-11)


@pytest.mark.parametrize("k,l,n", [(200, 200, 3000), (170, 160, 1500), (300, 24, 1500), (8, 520, 1200), (600, 5, 900),
                                   (1024, 3, 400)])
def test_any_group_count_runs_wide_rows(hip, k, l, n):
    """K, L beyond the 64-pair LDS stage (K = L = 200 and up to 1,024 groups): the reference's numpy
    backend has no size limit (src/kernels_numpy.py:21-79); the library switches to its wide-row
    kernels instead of refusing.  Same checks as every other shape: numerators after one step,
    parameters + likelihood + predictions after a few iterations, either side paired with the rating."""
    rng = np.random.default_rng(k * 7 + l)
    n_u, n_i, n_r = 90, 40, 3
    data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(5, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(2):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for swap in (0, 1):
        with make_ctx(hip, data, theta, eta, pr, swap_sides=swap) as em:
            assert em.get_option("wide") == 1.0
            assert np.array_equal(em.compute_omegas(), orc.compute_omegas(data, theta, eta, pr))   # same inputs: bit exact
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, (nm, rel_err(got, w))
            em.iterate(2)
            for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-11, nm
            assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)
            assert np.allclose(em.prod_dist(data[:50]), orc.prod_dist(data[:50], t, e, p), rtol=1e-11, atol=1e-300)


def test_wide_row_kernels_agree_with_the_lds_stage(hip, monkeypatch):
    """The wide-row form forced on a shape the LDS stage handles (K=20, L=24, restart slots, long and
    empty segments): same results to rounding (the slab sums are associated differently), same
    mat-vec results bit for bit."""
    rng = np.random.default_rng(3)
    n_u, n_i, n_r, k, l = 400, 130, 5, 20, 24
    users = np.minimum((rng.pareto(1.2, 9000) * 4).astype(np.int64), n_u - 1)
    data = np.stack([users, rng.integers(0, n_i, 9000), rng.integers(0, n_r, 9000)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(9, n_u, n_i, n_r, k, l, d_u, d_i)
    outs = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("MMSBM_HIP_FORCE_WIDE", "1")
        with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=2) as em:
            assert em.get_option("wide") == (1.0 if force else 0.0)
            em.select(0).set_params(theta, eta, pr)
            em.select(1).set_params(theta[::-1].copy() / 1.0, eta, pr)
            em.iterate(3)
            outs.append([em.select(s).get_params() for s in range(2)])
    monkeypatch.delenv("MMSBM_HIP_FORCE_WIDE")
    for s in range(2):
        for a, b, nm in zip(outs[0][s], outs[1][s], ("theta", "eta", "pr")):
            assert rel_err(b, a) < 1e-12, (s, nm)
    t, e, p = theta, eta, pr
    for _ in range(3):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for got, w in zip(outs[1][0], (t, e, p)):
        assert rel_err(got, w) < 1e-11


@pytest.mark.parametrize("k,l,forced", [(50, 50, False), (40, 60, False), (33, 64, False), (64, 17, False), (36, 36, False),
                                        (20, 20, True), (7, 64, True), (3, 5, True), (16, 16, True), (48, 20, True)])
def test_pair_stage_on_the_matrix_cores(hip, k, l, forced):
    """pair_mfma_kernel (v_mfma_f64_16x16x4_f64): chosen by the library for K x L > 1024 with K, L <= 64
    (BASELINE's K = L = 50), forced here on smaller tiles too.  Ragged rating chunks (units of 1..64
    pairs, chunks of several units), restart slots, either side paired with the rating; numerators
    after one step and parameters / likelihood after three iterations against the oracle, and against
    the lane-per-pair kernels (same sums, associated differently)."""
    rng = np.random.default_rng(100 * k + l)
    n_u, n_i, n_r = 300, 170, 4
    n = 5000
    r_col = np.minimum(rng.geometric(0.45, n) - 1, n_r - 1)          # unequal rating counts: ragged chunk tails
    data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), r_col], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(11, n_u, n_i, n_r, k, l, d_u, d_i)
    theta2, eta2, pr2 = orc.init_params(12, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(3):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for swap in (0, 1):
        outs = {}
        for on in (1, 0):
            with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=2, swap_sides=swap) as em:
                if not forced:
                    assert em.get_option("mfma") == 1.0               # the library's own choice
                em.set_option("mfma", on)
                assert em.get_option("mfma") == float(on)
                em.select(1).set_params(theta2, eta2, pr2)
                em.select(0).set_params(theta, eta, pr)
                if on:
                    for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                        assert rel_err(got, w) < TOL_STEP, (swap, nm)
                        assert_elementwise(got, w, f"{nm} (swap {swap})", rtol=1e-12)
                em.iterate(3)
                outs[on] = [em.select(s).get_params() for s in range(2)]
                if on:
                    assert em.select(0).likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)
                    # (remainders of 4 or 8 groups of the A launch's output side run as 4 x 4 blocks on
                    # v_mfma_f64_4x4x4_4b_f64; the T + S launch keeps padded 16-tiles -- one form each, no switch)
        for got, w, nm in zip(outs[1][0], (t, e, p), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, (swap, nm)
        for s_ in range(2):
            for a, b, nm in zip(outs[1][s_], outs[0][s_], ("theta", "eta", "pr")):
                assert rel_err(a, b) < 1e-12, (swap, s_, nm)


@pytest.mark.parametrize("k,l,mode", [(80, 80, 1), (100, 100, 1), (130, 70, 1), (65, 16, 1), (200, 200, 1), (300, 24, 1),
                                      (600, 5, 1), (8, 520, 1), (1024, 3, 1), (3, 1024, 1), (300, 8, 1), (200, 12, 1), (5, 260, 1),
                                      (20, 20, 2), (50, 50, 2), (7, 64, 2), (64, 64, 2)])
def test_pair_stage_on_the_matrix_cores_blocked(hip, k, l, mode):
    """K or L beyond 64: mfma_rows_kernel + mfma_slab_kernel (64 x 64 blocks), the library's own choice
    for every K x L > 1,024 the one-block kernel does not take -- skinny tiles included (one side of 3 .. 12
    groups: most of a 16-wide tile is padding there, round 3); forced (option mfma = 2) on shapes the one-block
    kernel or the vector kernels would take.  Ragged chunks, restart slots, either orientation; against the oracle
    and against the vector-ALU kernels (lane-per-pair stage or wide rows) of the same context."""
    rng = np.random.default_rng(1000 * k + l)
    n_u, n_i, n_r = 200, 150, 3
    n = 4000 if k * l < 20000 else 2500
    r_col = np.minimum(rng.geometric(0.5, n) - 1, n_r - 1)
    data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), r_col], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(21, n_u, n_i, n_r, k, l, d_u, d_i)
    theta2, eta2, pr2 = orc.init_params(22, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(2):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for swap in (0, 1):
        outs = {}
        with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=2, swap_sides=swap) as em:
            if mode == 1:
                assert em.get_option("mfma") == 2.0                   # the library's own choice
            for on in (mode, 0):
                em.set_option("mfma", on)
                assert em.get_option("mfma") == (2.0 if on else 0.0)
                if on:                                                # (a small problem must not stay with the
                    assert em.get_option("launches") == 4.0           # two-launch kernels once a pair-stage family is forced)
                em.select(1).set_params(theta2, eta2, pr2)
                em.select(0).set_params(theta, eta, pr)
                if on:
                    for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                        assert rel_err(got, w) < TOL_STEP, (swap, nm)
                        assert_elementwise(got, w, f"{nm} (swap {swap})", rtol=1e-12)
                em.iterate(2)
                outs[on] = [em.select(s_).get_params() for s_ in range(2)]
                if on:
                    assert em.select(0).likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)
                    assert np.allclose(em.select(0).prod_dist(data[:400]), orc.prod_dist(data[:400], t, e, p), rtol=1e-11, atol=1e-300)
        for got, w, nm in zip(outs[mode][0], (t, e, p), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, (swap, nm)
        for s_ in range(2):
            for a, b, nm in zip(outs[mode][s_], outs[0][s_], ("theta", "eta", "pr")):
                assert rel_err(a, b) < 1e-12, (swap, s_, nm)


@pytest.mark.parametrize("k,l,family", [(32, 32, 0.0), (31, 32, 0.0), (31, 33, 1.0), (32, 36, 1.0), (36, 29, 1.0), (64, 64, 1.0), (61, 64, 1.0),
                                        (64, 65, 2.0), (68, 16, 2.0), (13, 100, 2.0), (12, 100, 2.0), (9, 130, 2.0), (8, 128, 0.0), (3, 250, 0.0), (4, 260, 2.0)])
def test_pair_stage_families_at_their_borders(hip, k, l, family):
    """Which kernels take the pair stage is decided from (K, L): vector ALUs while the padded tile is at most
    1,024 entries, the one-block matrix-core kernel up to 64 per side, the blocked kernels beyond -- skinny
    tiles included (round 3).  Shapes on both sides of every border, against the oracle."""
    rng = np.random.default_rng(k * 131 + l)
    n_u, n_i, n_r, n = 90, 70, 3, 1500
    data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(k + l, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(2):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    with make_ctx(hip, data, theta, eta, pr, swap_sides=0) as em:
        assert em.get_option("mfma") == family, (k, l, em.get_option("mfma"))
        for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, w) < TOL_STEP, nm
        em.iterate(2)
        for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, nm
        assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)


@pytest.mark.parametrize("swap", [0, 1])
def test_result_is_likelihood_plus_parameters(hip, swap):
    """mmsbm_hip_result (the parameter download overlapped with the likelihood kernels) returns bit for bit
    what likelihood() and get_params() return, for every slot, either orientation, and leaves the context
    ready to go on iterating."""
    rng = np.random.default_rng(5)
    n_u, n_i, n_r, k, l = 700, 300, 5, 12, 9
    data = np.stack([rng.integers(0, n_u, 20000), rng.integers(0, n_i, 20000), rng.integers(0, n_r, 20000)], axis=1).astype(np.int64)
    with hip.HipEM(data, k, l, n_u, n_i, n_r, slots=3, swap_sides=swap) as em:
        for s_ in range(3):
            em.select(s_).init_params(10 + s_)
        em.iterate(7)
        for s_ in range(3):
            lik, t, e, p = em.select(s_).result()
            assert lik == em.likelihood()
            for a, b in zip((t, e, p), em.get_params()):
                assert np.array_equal(a, b)
        em.iterate(2)
        lik2, t2, e2, p2 = em.select(1).result()
        assert lik2 == em.likelihood() and not np.array_equal(t2, t)


def test_big_tiles_on_small_and_degenerate_data(hip):
    """The matrix-core pair stage on inputs far from its design point: a handful of triples, one rating
    value, ratings without rows, one item, one user, absent ids, more groups than rows -- random shapes
    with K, L in 33..140 (one-block and blocked kernels), numerators after one step and parameters after
    two iterations against the oracle."""
    rng = np.random.default_rng(77)
    cases = [(1, 1, 1, 1, 50, 50), (3, 2, 1, 4, 64, 33), (40, 1, 30, 2, 40, 60), (40, 30, 1, 2, 100, 70),
             (200, 50, 20, 1, 52, 52), (300, 40, 25, 9, 36, 90)]
    cases += [(int(rng.integers(5, 400)), int(rng.integers(2, 60)), int(rng.integers(2, 40)), int(rng.integers(1, 8)),
               int(rng.integers(33, 141)), int(rng.integers(33, 141))) for _ in range(8)]
    for n, n_u, n_i, n_r, k, l in cases:
        data = np.stack([rng.integers(0, max(1, n_u - 1), n), rng.integers(0, max(1, n_i - 1), n),   # the last ids never occur
                         rng.integers(0, n_r, n)], axis=1).astype(np.int64)
        if n_r > 2:
            data[data[:, 2] == 1, 2] = 0                                                             # a rating value without rows
        d_u, d_i = orc.degrees(data, n_u, n_i)
        theta, eta, pr = orc.init_params(n + k, n_u, n_i, n_r, k, l, d_u, d_i)
        want = orc.update_coefficients(data, theta, eta, pr)
        t, e, p = theta, eta, pr
        for _ in range(2):
            t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
        with make_ctx(hip, data, theta, eta, pr) as em:
            assert em.get_option("mfma") in (1.0, 2.0), (k, l)
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, (n, n_u, n_i, n_r, k, l, nm)
            em.iterate(2)
            for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-11, (n, n_u, n_i, n_r, k, l, nm)
            assert np.allclose(em.prod_dist(data[:20]), orc.prod_dist(data[:20], t, e, p), rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("n_r,k,l", [(1, 3, 4), (33, 5, 6), (100, 4, 3), (7, 20, 20)])
def test_many_or_single_rating_values(hip, n_r, k, l):
    """R = 1 (p stays 1 everywhere) up to R = 100 (many tiny rating-homogeneous units, several
    passes of six ratings in p_update), some rating values unused."""
    rng = np.random.default_rng(n_r)
    n_u, n_i = 120, 60
    r_col = rng.integers(0, n_r, 2500)
    if n_r > 10:
        r_col[r_col % 7 == 3] = 0            # leave some rating values without any row
    data = np.stack([rng.integers(0, n_u, 2500), rng.integers(0, n_i, 2500), r_col], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(2, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(4):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for swap in (0, 1):
        with make_ctx(hip, data, theta, eta, pr, swap_sides=swap) as em:
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, nm
            em.iterate(4)
            for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-11, nm
            assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)
            test = data[:200]
            em.predict_begin(test, np.arange(n_r, dtype=np.float64))
            st = hip.HipEM.final_stats(em.predict_add())
            ref = orc.score_stats(orc.prod_dist(test, t, e, p), test[:, 2], list(range(n_r)))
            assert st["s2"] == ref["s2"] or rel_err(em.prod_dist(test), orc.prod_dist(test, t, e, p)) < 1e-11
            em.predict_finish()


def test_dense_data_many_split_segments(hip):
    """Dense data (few users/items, hundreds of ratings each, the MovieLens regime): every segment is
    cut into 16-triple work items and summed by the one-group-per-segment combine; plus one heavy
    user whose pieces go through the workgroup combine.  Parity with the oracle, reproducible."""
    rng = np.random.default_rng(31)
    n, n_u, n_i, n_r, k, l = 60_000, 300, 200, 5, 10, 12
    u = np.where(rng.random(n) < 0.15, 11, rng.integers(0, n_u, n))       # user 11: ~9,000 rows -> > 32 pieces
    data = np.stack([u, rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    assert d_u.max() > 32 * 16 and np.median(d_u) > 100
    lay = hip.core.build_layout(data, n_u, n_i, n_r)
    pieces = lay["user_splits"][:, 2]
    assert (pieces <= 32).sum() > 250 and (pieces > 32).sum() >= 1        # both combine kernels run
    theta, eta, pr = orc.init_params(8, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(3):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    outs = []
    for swap in (0, 1, 0):
        with make_ctx(hip, data, theta, eta, pr, swap_sides=swap) as em:
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, nm
            em.iterate(3)
            outs.append(em.get_params())
            for got, w, nm in zip(outs[-1], (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-11, nm
    for a, b in zip(outs[0], outs[2]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("ranges", ["8,8", "16,24", "1,8"])
def test_xcd_local_work_lists(hip, ranges, monkeypatch):
    """Dense data with the segments cut at fixed borders of the gathered index (one range of the
    table per XCD; forced here through MMSBM_HIP_RANGES since the tables of a test-sized problem
    fit any L2): same results as the oracle, reproducible, with restart slots too."""
    monkeypatch.setenv("MMSBM_HIP_RANGES", ranges)
    rng = np.random.default_rng(41)
    n, n_u, n_i, n_r, k, l = 50_000, 260, 180, 4, 12, 9
    u = np.where(rng.random(n) < 0.1, 7, rng.integers(0, n_u, n))
    data = np.stack([u, rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    data = data[data[:, 0] != 100]                                     # an id that never occurs
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(6, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(3):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    outs = []
    for swap in (0, 1, 0):
        with make_ctx(hip, data, theta, eta, pr, swap_sides=swap) as em:
            for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
                assert rel_err(got, w) < TOL_STEP, nm
            em.iterate(3)
            outs.append(em.get_params())
            for got, w, nm in zip(outs[-1], (t, e, p), ("theta", "eta", "pr")):
                assert rel_err(got, w) < 1e-11, nm
    for a, b in zip(outs[0], outs[2]):
        assert np.array_equal(a, b)
    with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0, slots=2) as em:
        em.select(0).set_params(theta, eta, pr)
        em.select(1).set_params(theta * 1.01, eta, pr)
        em.iterate(3)
        for a, b in zip(em.select(0).get_params(), outs[0]):
            assert np.array_equal(a, b)


def test_xcd_local_work_lists_chosen_by_the_data(hip):
    """A shape for which the policy itself turns the XCD-local lists on (no environment override):
    40 (item, rating) pairs with 15,000 ratings each gather a 9.6 MB theta table."""
    rng = np.random.default_rng(12)
    n, n_u, n_i, n_r, k, l = 600_000, 60_000, 10, 4, 20, 3
    data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(4, n_u, n_i, n_r, k, l, d_u, d_i)
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = orc.em_step(data, theta, eta, pr, d_u, d_i)
    t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    with make_ctx(hip, data, theta, eta, pr, swap_sides=0) as em:
        assert em.get_option("ranges_pairs") == 8 and em.get_option("ranges_users") == 1
        assert em.get_option("items_pairs") > 8 * 40 and em.get_option("items_users") == 0
        for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
            assert rel_err(got, w) < TOL_STEP, nm
        em.iterate(2)
        for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
            assert rel_err(got, w) < 1e-11, nm
        assert em.likelihood() == pytest.approx(float(orc.compute_likelihood(data, t, e, p)), rel=1e-11)


def test_level1_cache_never_serves_another_training_set(hip):
    """two training sets that differ only in rows the old fingerprint never sampled (two rows
    swapped: equal shapes, equal column sums) must not share a cached device context."""
    from mmsbm_amd import kernels_hip
    kernels_hip.clear_cache()
    rng = np.random.default_rng(4)
    n, n_u, n_i, n_r, k, l = 6000, 300, 120, 4, 6, 5
    a = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    b = a.copy()
    b[[1, 2]] = b[[2, 1]]                                  # (rows 1 and 2 differ; stride of the old sample: 2)
    assert not np.array_equal(a[1], a[2])
    d_u, d_i = orc.degrees(a, n_u, n_i)
    theta, eta, pr = orc.init_params(1, n_u, n_i, n_r, k, l, d_u, d_i)
    om_a = kernels_hip.compute_omegas(a, theta, eta, pr)
    om_b = kernels_hip.compute_omegas(b, theta, eta, pr)   # omegas come back in the ROW ORDER of `data`
    assert np.array_equal(om_a, orc.compute_omegas(a, theta, eta, pr))
    assert np.array_equal(om_b, orc.compute_omegas(b, theta, eta, pr))
    assert np.array_equal(om_b[1], om_a[2]) and not np.array_equal(om_b[1], om_a[1])
    assert len(kernels_hip._cache) == 2                    # two contexts: a miss, not a stale hit
    kernels_hip.update_coefficients(a, theta, eta, pr)
    assert len(kernels_hip._cache) == 2                    # ... and the same data again is a hit
    # Round 5: a call on the array object of the most recent context goes ahead on that context while the digest is
    # computed beside it -- so an array REWRITTEN IN PLACE (same object, same address, another training set) is the
    # case the digest has to catch: the speculative result is dropped, the call repeated on the right context
    want_a = orc.update_coefficients(a, theta, eta, pr)
    for got, w in zip(kernels_hip.update_coefficients(a, theta, eta, pr), want_a):   # (the fast path itself)
        assert rel_err(got, w) < TOL_STEP
    a[[1, 2]] = a[[2, 1]]                                  # now `a` holds what `b` holds
    a[7] = a[9]
    want_new = orc.update_coefficients(a, theta, eta, pr)
    assert rel_err(want_new[0], want_a[0]) > 1e-6          # (a training set with other numerators)
    for got, w in zip(kernels_hip.update_coefficients(a, theta, eta, pr), want_new):
        assert rel_err(got, w) < TOL_STEP
    assert np.array_equal(kernels_hip.compute_omegas(a, theta, eta, pr), orc.compute_omegas(a, theta, eta, pr))
    kernels_hip.clear_cache()


def test_compute_likelihood_on_held_out_data(hip):
    """src/mmsbm.py:541-553 evaluates the `data` it is given: the training set through the resident
    context, any other set through a context of its own."""
    g = load_golden("g4_2k_k10")
    train = g["train"]
    mm = hip.MMSBM(10, 10, iterations=5, seed=3)
    mm.fit_encoded(train)
    res = mm.results[0]
    held = train[300:900]
    for data in (train, train.copy(), held):
        want = float(orc.compute_likelihood(data, res["theta"], res["eta"], res["pr"]))
        assert mm.compute_likelihood(data, res["theta"], res["eta"], res["pr"]) == pytest.approx(want, rel=1e-11)
    mm._release()


def test_device_memory_query_and_slot_sizing(hip):
    import ctypes as C
    from mmsbm_amd import _lib
    free, total = C.c_int64(0), C.c_int64(0)
    _lib.call("mmsbm_hip_device_mem", 0, C.byref(free), C.byref(total))
    assert 0 < free.value <= total.value and total.value > 100 * 2**30     # an MI355X: 288 GB
    data = orc.synthetic_triples(20_000, 2_000, 500, 5, seed=1)
    with hip.HipEM(data, 10, 10) as em:
        one = em.max_slots(0.5)
        assert one >= 8 and em.max_slots(0.5, sharers=4) <= one // 2 + 1
        assert em.max_slots(1e-12) == 1                                     # never below one slot
        em.set_slots(3)
        assert em.slots == 3 and em.max_slots(0.5) >= one - 3
    with pytest.raises(_lib.HipLibraryError):
        _lib.call("mmsbm_hip_device_mem", 99, C.byref(free), C.byref(total))


@pytest.mark.parametrize("shape", ["uniform", "skewed", "dense", "gaps"])
def test_device_built_layout_equals_the_host_layout(hip, shape, monkeypatch):
    """mmsbm_hip_create sorts large inputs on the GPU (layout_gpu.hpp: radix sorts, scan, lower bounds)
    instead of the host's counting sorts.  Both must give the same context: degrees, pair count and --
    since every sum runs in the order the layout fixes -- bit-identical numerators and parameters."""
    rng = np.random.default_rng(11)
    if shape == "uniform":
        n, n_u, n_i, n_r, k, l = 60_000, 7_000, 1_500, 5, 20, 20
        data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1)
    elif shape == "skewed":      # heavy users / items: split segments, ordered work lists
        n, n_u, n_i, n_r, k, l = 80_000, 3_000, 900, 4, 7, 13
        data = np.stack([np.minimum((rng.pareto(1.1, n) * 3).astype(np.int64), n_u - 1),
                         np.minimum((rng.pareto(1.3, n) * 2).astype(np.int64), n_i - 1), rng.integers(0, n_r, n)], axis=1)
    elif shape == "dense":       # few users and items, many duplicates of the same (u, i, r)
        n, n_u, n_i, n_r, k, l = 50_000, 60, 40, 3, 10, 10
        data = np.stack([rng.integers(0, n_u, n), rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1)
    else:                        # ids and a whole rating that never occur; one triple only for some pairs
        n, n_u, n_i, n_r, k, l = 5_000, 9_000, 4_000, 6, 5, 4
        data = np.stack([rng.integers(0, n_u, n) // 3 * 3, rng.integers(0, n_i, n) // 2 * 2,
                         rng.choice([0, 1, 3, 5], n)], axis=1)
    data = data.astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(2, n_u, n_i, n_r, k, l, d_u, d_i)
    outs = []
    for gpu in ("0", "1"):
        monkeypatch.setenv("MMSBM_HIP_GPU_LAYOUT", gpu)
        for swap in (0, 1):
            with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=swap) as em:
                em.set_params(theta, eta, pr)
                num = em.update_coefficients()
                em.iterate(3)
                outs.append((gpu, swap, em.n_pairs, em.degrees(), num, em.get_params(), em.likelihood()))
    monkeypatch.delenv("MMSBM_HIP_GPU_LAYOUT")
    for swap in (0, 1):
        host, dev = (o for o in outs if o[1] == swap)
        assert host[0] == "0" and dev[0] == "1" and host[2] == dev[2]
        assert all(np.array_equal(a, b) for a, b in zip(host[3], dev[3]))
        assert all(np.array_equal(a, b) for a, b in zip(host[4], dev[4]))
        assert all(np.array_equal(a, b) for a, b in zip(host[5], dev[5]))
        assert host[6] == dev[6]
    want = orc.update_coefficients(data, theta, eta, pr)
    for got, w in zip(outs[2][4], want):
        assert rel_err(got, w) < TOL_STEP
    # bad ids are refused before anything is uploaded, whichever builder would run
    bad = data.copy()
    bad[7, 1] = n_i
    for gpu in ("0", "1"):
        monkeypatch.setenv("MMSBM_HIP_GPU_LAYOUT", gpu)
        with pytest.raises(Exception, match="out of range"):
            hip.HipEM(bad, k, l, n_u, n_i, n_r)
    monkeypatch.delenv("MMSBM_HIP_GPU_LAYOUT")


def test_device_built_range_cuts_equal_the_host_ones(hip, monkeypatch):
    """XCD-local work lists (segments cut at borders of the gathered index) built from cut positions
    found on the device give the very same lists as the host builder: bit-identical results."""
    rng = np.random.default_rng(12)
    n, n_u, n_i, n_r, k, l = 120_000, 700, 260, 5, 10, 10
    users = np.minimum((rng.lognormal(0.0, 1.0, n) * 60).astype(np.int64), n_u - 1)
    data = np.stack([users, rng.integers(0, n_i, n), rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(3, n_u, n_i, n_r, k, l, d_u, d_i)
    monkeypatch.setenv("MMSBM_HIP_RANGES", "8,16")
    outs = []
    for gpu in ("0", "1"):
        monkeypatch.setenv("MMSBM_HIP_GPU_LAYOUT", gpu)
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as em:
            assert em.get_option("ranges_pairs") == 8 and em.get_option("ranges_users") == 16
            items = (em.get_option("items_pairs"), em.get_option("items_users"))
            em.set_params(theta, eta, pr)
            em.iterate(3)
            outs.append((items, em.get_params()))
    assert outs[0][0] == outs[1][0] and outs[0][0][0] > 0
    assert all(np.array_equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    t, e, p = theta, eta, pr
    for _ in range(3):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    for got, w in zip(outs[1][1], (t, e, p)):
        assert rel_err(got, w) < 1e-11

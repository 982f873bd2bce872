"""BASELINE.json's configurations at FULL size against the oracle, through the C ABI (``-m gpu``).

What VERDICT r1 asked for: every numerator (n_theta, n_eta, n_p) after one step and theta / eta / p,
the likelihood and the argmax predictions after several iterations at all 1M ratings of C3
(src/kernels_numpy.py:43-79 -- the dense N x K x L oracle takes ~3 s per step on the GPU box's host),
a C5-shaped dense problem (K = L = 50, R = 10) of 120k ratings the same way, and the two
``sampling = 8`` configurations (C4 = C3 x 8 restarts, C5 x 8 restarts) run as 8 restart slots of one
GPU (src/mmsbm.py:182-185 runs them as 8 processes; with 8 GPUs it is one restart per rank --
mmsbm_amd/restarts.py -- and restart i does not depend on where it runs).
"""
import numpy as np
import pytest

from conftest import assert_elementwise, rel_err
from oracle import mmsbm_factorised as fac
from oracle import mmsbm_oracle as orc

pytestmark = pytest.mark.gpu

TOL_STEP = 1e-12   # one update_coefficients call (max |diff| / max |want|)
TOL_FEW = 1e-11    # after a handful of EM iterations (drift envelope of SURVEY B.5: 1e-14 .. 1e-13)


@pytest.fixture(scope="module")
def hip():
    from mmsbm_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("-m gpu tests need a GPU: no HIP device visible (no CPU fallback exists)")
    import mmsbm_amd
    return mmsbm_amd


@pytest.fixture(scope="module")
def c3_train():
    train = orc.synthetic_triples(1_000_000, 100_000, 20_000, 5, seed=0)
    assert int(train[:, 0].max()) + 1 == 99_997
    return train


@pytest.fixture(scope="module")
def c5_train():
    return orc.synthetic_triples(10_000_000, 1_000_000, 100_000, 10, seed=0)


def _argmax_agreement(got, want):
    """Share of rows with identical argmax among those whose top-2 gap in the oracle exceeds 1e-9
    (exact structural ties make argmax ill-posed: SURVEY 7.3 item 6), and the share of such rows."""
    srt = np.sort(want, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-9
    return float(np.mean(np.argmax(got, 1)[clear] == np.argmax(want, 1)[clear])), float(clear.mean())


def _oracle_case(train, k, l, iters, seed):
    """What the dense oracle gives for restart 0 of MMSBM(k, l, seed=seed) on `train`: the start, the
    numerators of one step, the parameters after `iters` iterations, their likelihood and prod_dist."""
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    d_u, d_i = orc.degrees(train, n_u, n_i)
    start = orc.init_params(orc.child_seeds(seed, 1)[0], n_u, n_i, n_r, k, l, d_u, d_i)
    numer = orc.update_coefficients(train, *start)
    theta, eta, pr = start
    for _ in range(iters):
        theta, eta, pr = orc.em_step(train, theta, eta, pr, d_u, d_i)
    return {"d_u": d_u, "d_i": d_i, "start": start, "numer": numer, "params": (theta, eta, pr),
            "lik": float(orc.compute_likelihood(train, theta, eta, pr)),
            "pd": orc.prod_dist(train, theta, eta, pr), "iters": iters, "k": k, "l": l, "seed": seed}


def _check_against(hip, train, case, options=()):
    """One context on `train` against an _oracle_case: EVERY entry of the three numerators after one step
    and of theta / eta / p after the iterations -- in max-norm (rel_err) and element by element
    (|diff| <= 1e-9 |want| for every entry above 1e-290; north_star's bar is 1e-5) -- the likelihood,
    prod_dist and the argmax of every row.  Returns what the context says it ran."""
    mm = hip.MMSBM(case["k"], case["l"], iterations=case["iters"], seed=case["seed"])
    mm._prepare_objects(train)
    ctx = mm._ctx(0)
    for name, value in options:
        ctx.set_option(name, value)
        assert ctx.get_option(name) == float(value), (name, value, ctx.get_option(name))
    d_u, d_i = ctx.degrees()
    assert np.array_equal(d_u, case["d_u"]) and np.array_equal(d_i, case["d_i"])
    theta, eta, pr = mm.init_params(mm.child_states[0], d_u, d_i)
    for a, b in zip((theta, eta, pr), case["start"]):
        assert np.array_equal(a, b)
    ctx.set_params(theta, eta, pr)
    got = ctx.update_coefficients()
    for g, w, nm in zip(got, case["numer"], ("n_theta", "n_eta", "n_pr")):
        assert g.shape == w.shape
        assert rel_err(g, w) < TOL_STEP, (nm, rel_err(g, w))
        assert_elementwise(g, w, nm)
    ctx.iterate(case["iters"])
    for g, w, nm in zip(ctx.get_params(), case["params"], ("theta", "eta", "pr")):
        assert rel_err(g, w) < TOL_FEW, (nm, rel_err(g, w))
        assert_elementwise(g, w, nm)
    lik = ctx.likelihood()
    assert abs(lik - case["lik"]) <= 1e-11 * abs(case["lik"]), (lik, case["lik"])
    pd_h = ctx.prod_dist(train)
    assert rel_err(pd_h, case["pd"]) < TOL_FEW
    agree, clear = _argmax_agreement(pd_h, case["pd"])
    assert agree == 1.0 and clear > 0.99, (agree, clear)
    info = {nm: ctx.get_option(nm) for nm in ("mfma", "quad", "chunk_pairs", "n_chunks")}
    mm._release()
    return info


def _full_parity(hip, train, k, l, iters, seed):
    return _check_against(hip, train, _oracle_case(train, k, l, iters, seed))


def test_c3_full_size_all_numerators_parameters_likelihood_argmax(hip, c3_train):
    """C3 = BASELINE configs[2], the headline workload: 1M ratings, 99,997 x 20,000, R=5, K=L=20."""
    _full_parity(hip, c3_train, 20, 20, iters=3, seed=0)


@pytest.fixture(scope="module")
def c5_shape():
    """C5's shape (K = L = 50, R = 10, 10 ratings per user, 100 per item) at 120k ratings, where the dense
    oracle (2.4 GB per tensor) fits; 1,200 pairs per rating: every chunk size leaves a ragged last chunk."""
    train = orc.synthetic_triples(120_000, 12_000, 1_200, 10, seed=5)
    return train, _oracle_case(train, 50, 50, iters=3, seed=11)


def test_c5_shape_dense_subproblem_all_numerators_parameters_likelihood_argmax(hip, c5_shape):
    """The library's own choice for this shape: the matrix-core pair stage, 256-pair chunks."""
    info = _check_against(hip, *c5_shape)
    assert info["mfma"] == 1.0 and info["chunk_pairs"] == 256


@pytest.mark.parametrize("chunk", [512, 1024])
def test_c5_shape_with_the_chunk_sizes_full_size_c5_runs(hip, c5_shape, chunk, monkeypatch):
    """From 524,288 pairs on (only full-size C5 gets there) `mmsbm_hip_create` gives every matrix-core
    workgroup 512 pairs = 8 units; the kernel takes up to 1,024.  Forced here (MMSBM_HIP_MFMA_CHUNK) on the
    120k-rating problem the dense oracle can hold: 1,200 pairs per rating = chunks of 512 + 512 + 176
    (resp. 1,024 + 176) and empty padding chunks up to 8 per rating.  EVERY entry of n_theta / n_eta /
    n_p and of theta / eta / p is compared, element-wise too: a permuted or mis-accumulated (k, l) tile of
    a slab cannot pass (src/kernels_numpy.py:43-79)."""
    monkeypatch.setenv("MMSBM_HIP_MFMA_CHUNK", str(chunk))
    info = _check_against(hip, *c5_shape)
    assert info["mfma"] == 1.0 and info["chunk_pairs"] == chunk
    assert info["n_chunks"] == 10 * 8          # ceil(1200 / chunk) = 3 or 2 real chunks per rating, padded to 8


def test_the_a_launch_walks_runs_of_its_own_and_gives_the_same_rows(hip, c5_shape, monkeypatch):
    """The matrix-core A launch writes rows only, so it walks the units in runs of its own length, chosen so that its last
    round of workgroups is (nearly) full (mmsbm_hip.hip: balanced_run_units; C5: 11 units where the T + S launch takes 8).
    A rows do not depend on which workgroup computes them: theta / eta / p after three iterations are BITWISE the same
    for runs of 1, 3 and 16 units (ragged last runs, empty padding runs) as for the T + S launch's own 4-unit chunks --
    and every entry agrees with the dense oracle as in the tests above."""
    train, case = c5_shape
    outs = {}
    for units in (4, 1, 3, 16):
        mm = hip.MMSBM(case["k"], case["l"], iterations=3, seed=case["seed"])
        mm._prepare_objects(train)
        ctx = mm._ctx(0)
        assert ctx.get_option("mfma") == 1.0 and ctx.get_option("chunk_pairs") == 256
        ctx.set_option("a_units", units)
        assert ctx.get_option("a_units") == units
        # 1,200 pairs per rating = 19 units: ceil(19 / units) runs, padded to a multiple of 8 per rating (0: the T + S launch's list)
        runs = -(-19 // units)
        assert ctx.get_option("a_chunks") == (0 if units == 4 else 10 * (-(-runs // 8) * 8))
        ctx.set_params(*case["start"])
        ctx.iterate(3)
        outs[units] = ctx.get_params()
        mm._release()
    for units in (1, 3, 16):
        for a, b, nm in zip(outs[units], outs[4], ("theta", "eta", "pr")):
            assert np.array_equal(a, b), (units, nm)
    for got, want, nm in zip(outs[4], case["params"], ("theta", "eta", "pr")):
        assert rel_err(got, want) < TOL_FEW, nm


def test_c5_shape_on_the_vector_alus_with_the_persistent_a_pipeline(hip, c5_shape):
    """The same problem with the matrix cores off: pair_block_kernel (tile in LDS) for T + S and the
    persistent four-unit pipeline pair_quad_a_kernel for A -- what C5 ran before round 2 and what
    MMSBM_HIP_NO_MFMA / option mfma = 0 still select."""
    info = _check_against(hip, *c5_shape, options=(("mfma", 0), ("quad", 1)))
    assert info["mfma"] == 0.0 and info["quad"] == 1.0


def _check_invariants(train, res, d_u, d_i):
    t, e, p = res["theta"], res["eta"], res["pr"]
    assert np.allclose(t.sum(1), 1, atol=1e-13) and np.allclose(e.sum(1), 1, atol=1e-13)
    assert np.allclose(p.sum(2), 1, atol=1e-13)
    assert np.all(t >= 0) and np.all(e >= 0) and np.all(p >= 0)
    assert np.isfinite(res["likelihood"]) and res["likelihood"] < 0


def test_c4_sampling8_on_the_c3_workload(hip, c3_train):
    """BASELINE configs[3]: the C3 workload with sampling = 8.  On one GPU the 8 restarts are 8 slots
    of one context; restart i is what a sampling=1 run seeded with child seed i gives, bit for bit."""
    train = c3_train
    mm = hip.MMSBM(20, 20, iterations=3, sampling=8, seed=0)
    mm.fit_encoded(train)
    assert len(mm.results) == 8 and mm._ctx(0).slots == 8
    d_u, d_i = mm._ctx(0).degrees()
    liks = np.array([r["likelihood"] for r in mm.results])
    for r in mm.results:
        _check_invariants(train, r, d_u, d_i)
    assert len(set(liks.tolist())) == 8                      # eight different restarts
    assert mm.best_by_likelihood == int(np.argmax(liks))
    # restart 0 does not depend on `sampling`: bitwise the sampling=1 run
    one = hip.MMSBM(20, 20, iterations=3, sampling=1, seed=0)
    one.fit_encoded(train)
    for nm in ("theta", "eta", "pr"):
        assert np.array_equal(one.results[0][nm], mm.results[0][nm]), nm
    assert one.results[0]["likelihood"] == mm.results[0]["likelihood"]
    one._release()
    # one restart that is NOT slot 0, all entries, against the oracle's run of the same child seed
    s = 5
    want = orc.run_one_sampling(train, mm.child_states[s], 20, 20, 3)
    for nm in ("theta", "eta", "pr"):
        assert rel_err(mm.results[s][nm], want[nm]) < TOL_FEW, (nm, rel_err(mm.results[s][nm], want[nm]))
    assert abs(mm.results[s]["likelihood"] - want["likelihood"]) <= 1e-11 * abs(want["likelihood"])
    mm._release()


def test_c5_sampling8_full_size(hip, c5_train):
    """BASELINE configs[4] on one GPU: 10M ratings, 1M x 100k, R=10, K=L=50, sampling = 8 as 8 slots
    (HBM-capacity stress: ~2.5 GB per slot).  The dense oracle cannot hold this (omega = 200 GB); the
    factorised float64 checker (oracle/mmsbm_factorised.py, pinned to the dense oracle at 1e-13 in
    tests/test_oracle_golden.py) can: for a restart that is NOT slot 0, EVERY entry of n_theta / n_eta /
    n_p after one step and of theta / eta / p after two iterations, in max-norm and element-wise, and the
    likelihood (src/kernels_numpy.py:43-79, src/expectation_maximization.py:152-167); beside it a user
    block and an item block against the dense oracle itself, invariants for every restart, and restart 0
    bitwise the one-slot run.  This is the only test that reaches the 512-pair (8-unit) matrix-core
    workgroups by the library's own choice."""
    train = c5_train
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    seeds = orc.child_seeds(0, 8)
    pairs = fac.Pairs(train, n_u, n_i, n_r)
    with hip.HipEM(train, 50, 50, n_u, n_i, n_r, device=0) as em:
        assert em.n_pairs == pairs.n_pairs
        assert em.get_option("mfma") == 1.0 and em.get_option("chunk_pairs") == 512
        assert em.max_slots(0.5) >= 8, "8 restarts of C5 must fit one MI355X"
        em.set_slots(8)
        assert em.slots == 8
        d_u, d_i = em.degrees()
        assert np.array_equal(d_u, np.bincount(train[:, 0])) and np.array_equal(d_i, np.bincount(train[:, 1]))
        for s in range(8):
            em.select(s).init_params(seeds[s])
        # slot 6: un-normalised numerators, every entry against the factorised checker
        s = 6
        theta, eta, pr = orc.init_params(seeds[s], n_u, n_i, n_r, 50, 50, d_u, d_i)
        n_t, n_e, n_p = em.select(s).update_coefficients()
        for g, w, nm in zip((n_t, n_e, n_p), fac.update_coefficients(train, theta, eta, pr, pairs),
                            ("n_theta", "n_eta", "n_pr")):
            assert rel_err(g, w) < TOL_STEP, (nm, rel_err(g, w))
            assert_elementwise(g, w, nm)
        # ... and two blocks against the dense oracle itself (rows of 40 users / of 3 items are complete)
        sub = train[train[:, 0] < 40]
        assert rel_err(n_t[:40], orc.update_coefficients(sub, theta, eta, pr)[0][:40]) < TOL_STEP
        sub = train[train[:, 1] < 3]
        assert rel_err(n_e[:3], orc.update_coefficients(sub, theta, eta, pr)[1][:3]) < TOL_STEP
        assert np.allclose(n_t.sum(1), d_u, rtol=1e-12) and np.allclose(n_e.sum(1), d_i, rtol=1e-12)
        assert np.allclose(n_p.sum(axis=(0, 1)), np.bincount(train[:, 2]), rtol=1e-11)
        del n_t, n_e
        em.iterate(2)
        for _ in range(2):
            theta, eta, pr = fac.em_step(train, theta, eta, pr, d_u, d_i, pairs)
        liks = []
        first = None
        for s in range(8):
            t, e, p = em.select(s).get_params()
            lik = float(em.likelihood())
            _check_invariants(train, {"theta": t, "eta": e, "pr": p, "likelihood": lik}, d_u, d_i)
            liks.append(lik)
            if s == 0:
                first = (t, e, p, lik)
            if s == 6:
                for g, w, nm in zip((t, e, p), (theta, eta, pr), ("theta", "eta", "pr")):
                    assert rel_err(g, w) < TOL_FEW, (nm, rel_err(g, w))
                    assert_elementwise(g, w, nm)
                lik_f = float(fac.compute_likelihood(train, theta, eta, pr, pairs))
                assert abs(lik - lik_f) <= 1e-11 * abs(lik_f), (lik, lik_f)
        del theta, eta
        assert len(set(liks)) == 8
        # the same through a one-slot context: slot 0 of the batch is bitwise that
        em.set_slots(1)
        em.init_params(seeds[0])
        em.iterate(2)
        for a, b in zip(em.get_params(), first[:3]):
            assert np.array_equal(a, b)
        assert float(em.likelihood()) == first[3]


def test_host_class_sampling8_c5_shape_batches_by_free_memory(hip):
    """`MMSBM(sampling=8)` on a C5-shaped problem through the host class: the batch size comes from
    the memory that is free on the device (`max_slots`), results are those of one-at-a-time runs."""
    train = orc.synthetic_triples(200_000, 20_000, 2_000, 10, seed=7)
    mm = hip.MMSBM(50, 50, iterations=2, sampling=8, seed=4)
    mm.fit_encoded(train)
    assert mm._ctx(0).slots == 8
    solo = hip.MMSBM(50, 50, iterations=2, sampling=8, seed=4, restarts_per_launch=1)
    solo.fit_encoded(train, restarts=[0, 7])
    for j, i in enumerate((0, 7)):
        for nm in ("theta", "eta", "pr"):
            assert np.array_equal(solo.results[j][nm], mm.results[i][nm]), (i, nm)
    assert mm.best_by_likelihood == int(np.argmax([r["likelihood"] for r in mm.results]))
    mm._release(); solo._release()


# ---- the reference's DEFAULT run length (iterations=400, src/mmsbm.py:63-72) on north_star's own config and on the
# ---- matrix-core kernel family: fixtures made by RUNNING THE REFERENCE (tests/golden/make_golden.py: g8, g9) ----------
def _long_run_against_the_reference(hip, name, expect, min_clear=0.99):
    """Restart 0 of MMSBM(k, l, seed=0) from the reference's own start through the library's own kernel choice, against
    the reference's snapshots after 100, 200 and 400 iterations: sampled theta / eta entries and ALL of p element-wise
    (1e-6; north_star's bar is 1e-5) and in max-norm (1e-9), the column sums, the likelihood (1e-9) and the argmax
    prediction of every training row whose top-2 gap in the reference exceeds 1e-9.  Returns the worst element-wise
    error per snapshot (printed with -s; quoted in DESIGN.md)."""
    from conftest import elem_rel_err, load_golden
    g = load_golden(name)
    n, u, i, r, k, l = (int(g[x]) for x in ("n", "u", "i", "r", "k", "l"))
    train = orc.synthetic_triples(n, u, i, r, int(g["gen_seed"]))
    assert np.array_equal(train.sum(0), g["train_sum"]) and np.array_equal(train[:64], g["train_head"])
    mm = hip.MMSBM(k, l, iterations=int(max(g["snapshots"])), seed=int(g["model_seed"]))
    mm._prepare_objects(train)
    ctx = mm._ctx(0)
    for option, value in expect.items():            # the library's OWN choice of kernels, not a forced one
        assert ctx.get_option(option) == value, (option, ctx.get_option(option))
    d_u, d_i = ctx.degrees()
    theta0, eta0, pr0 = mm.init_params(mm.child_states[0], d_u, d_i)
    assert np.array_equal(theta0[g["ut"], g["kt"]], g["theta_s_0"]) and np.array_equal(eta0[g["ie"], g["le"]], g["eta_s_0"])
    assert np.array_equal(pr0, g["pr_0"])             # the reference's own start, bit for bit
    ctx.set_params(theta0, eta0, pr0)
    snaps = [int(x) for x in g["snapshots"]]
    assert snaps == sorted(snaps) and len(snaps) == 3
    done, worst = 0, {}
    for j, it in enumerate(snaps):
        ctx.iterate(it - done)
        done = it
        t, e, p = ctx.get_params()
        errs = []
        for got, want, nm in ((t[g["ut"], g["kt"]], g[f"theta_s_{it}"], "theta entries"),
                              (e[g["ie"], g["le"]], g[f"eta_s_{it}"], "eta entries"), (p, g[f"pr_{it}"], "p")):
            assert rel_err(got, want) < 1e-9, (nm, it, rel_err(got, want))
            assert_elementwise(got, want, f"{name}: {nm} after {it} iterations", rtol=1e-6)
            errs.append(elem_rel_err(got, want))
        assert rel_err(t.sum(0), g[f"theta_colsum_{it}"]) < 1e-9 and rel_err(e.sum(0), g[f"eta_colsum_{it}"]) < 1e-9
        assert ctx.likelihood() == pytest.approx(float(g["likelihood_at"][j]), rel=1e-9)
        clear = np.unpackbits(g[f"clear_{it}"])[:len(train)].astype(bool)
        assert clear.mean() > min_clear, clear.mean()
        assert np.array_equal(np.argmax(ctx.prod_dist(train), 1)[clear], g[f"argmax_{it}"][clear]), it
        worst[it] = max(errs)
    print(f"{name}: worst element-wise relative error vs the reference after {' / '.join(map(str, snaps))} iterations: "
          + " / ".join(f"{worst[it]:.1e}" for it in snaps))
    return worst


def test_c3_400_iterations_against_the_reference_on_the_headline_kernels(hip):
    """north_star's parity clause on north_star's own config at the reference's own default: C3 (1M ratings, 100k x 20k,
    R = 5, K = L = 20, seed 0), 400 iterations, the four-launch kernels the bench line measures (seg_pass_kernel<8,4,4>,
    pair_block_kernel, eta_p_kernel<8,4>) -- against the REAL reference's run (fixture g8_c3_400: 95 minutes of it)."""
    _long_run_against_the_reference(hip, "g8_c3_400", {"launches": 4.0, "mfma": 0.0})


def test_k50_400_iterations_against_the_reference_on_the_matrix_cores(hip):
    """The same for the kernel family BASELINE's config 5 runs (K = L = 50, R = 10: pair_mfma_kernel on both pair-stage
    launches, seg_pass_kernel<16,4,4>, lik_wave_kernel) on a C5-shaped problem the dense reference can hold (100k ratings
    of 10k users x 1k items; fixture g9_k50_400)."""
    # (50 x 50 groups on 100k ratings: 3 % of the rows end in an exact tie of their two best ratings in the reference itself)
    _long_run_against_the_reference(hip, "g9_k50_400", {"launches": 4.0, "mfma": 1.0}, min_clear=0.95)


def test_k80_200_iterations_against_the_reference_on_the_blocked_matrix_core_kernels(hip):
    """The third pair-stage family: a side beyond 64 groups runs the two products in 64 x 64 blocks (mfma_rows_kernel +
    mfma_slab_kernel; option `mfma` reads 2).  40k ratings of 4k users x 400 items, R = 8, K = L = 80, against the real
    reference's run (fixture g10_k80_200: snapshots after 50 / 100 / 200 iterations)."""
    _long_run_against_the_reference(hip, "g10_k80_200", {"launches": 4.0, "mfma": 2.0}, min_clear=0.9)

"""One parity case per kernel instantiation the shape-driven suite does not reach by itself.

The library chooses its kernels from the shape and the data (mmsbm_amd/csrc/tu_*.hip), so "does every compiled
kernel WORK, or just compile?" is a question about the test shapes.  The launch log (MMSBM_HIP_LAUNCH_LOG, set by
conftest.py) answers it: scripts/kernel_coverage.py lists the compiled instantiations no test launched.  Every case here
was written from that list (profiles/r6_kernel_coverage.csv): it builds the shape that selects the kernel, checks the
result against the oracle like every other parity test -- and asserts, from the log, that the kernel it is about really
ran.  The reference has no size limit (src/kernels_numpy.py:21-96), so every instantiation is a shape a user can ask for.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_elementwise, rel_err
from oracle import mmsbm_factorised as fact
from oracle import mmsbm_oracle as orc

sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_coverage  # noqa: E402  (demangling + canonical kernel names, shared with the coverage report)

pytestmark = pytest.mark.gpu

TOL_STEP = 1e-12


@pytest.fixture(scope="module")
def hip():
    from mmsbm_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("-m gpu tests need a GPU: no HIP device visible (no CPU fallback exists)")
    import mmsbm_amd
    return mmsbm_amd


class LaunchWindow:
    """Kernels this process launched between __enter__ and names(): the library appends its launch counts to the log
    whenever a context is destroyed, so read it after the contexts of the test are closed."""

    def __init__(self):
        self.path = os.environ.get("MMSBM_HIP_LAUNCH_LOG", "")

    def __enter__(self):
        self.pos = os.path.getsize(self.path) if self.path and os.path.exists(self.path) else 0
        return self

    def __exit__(self, *exc):
        return False

    def names(self):
        if not self.path:
            pytest.skip("launch log switched off (MMSBM_HIP_LAUNCH_LOG is empty)")
        with open(self.path) as fh:
            fh.seek(self.pos)
            rows = [ln.rstrip("\n").split("\t") for ln in fh]
        mine = [r[2] for r in rows if len(r) >= 4 and r[0] == str(os.getpid())]
        return {kernel_coverage.canon(n) for n in kernel_coverage.demangle(mine)} if mine else set()


def uniform(n, n_u, n_i, n_r, seed):
    data = orc.synthetic_triples(n, n_u, n_i, n_r, seed=seed)
    return data, tuple(int(data[:, j].max()) + 1 for j in range(3))


def check_step_and_loop(em, data, theta, eta, pr, d_u, d_i, iters=2, checker=orc, what=""):
    want = checker.update_coefficients(data, theta, eta, pr)
    for got, w, nm in zip(em.update_coefficients(), want, ("n_theta", "n_eta", "n_pr")):
        assert rel_err(got, w) < TOL_STEP, (what, nm)
        assert_elementwise(got, w, f"{what} {nm}", rtol=1e-11)
    em.iterate(iters)
    t, e, p = theta, eta, pr
    for _ in range(iters):
        t, e, p = checker.em_step(data, t, e, p, d_u, d_i)
    for got, w, nm in zip(em.get_params(), (t, e, p), ("theta", "eta", "pr")):
        assert rel_err(got, w) < 1e-11, (what, nm)
    return t, e, p


# ---- the triple passes with four rows in flight (more than 300,000 ratings) at the row widths the big configs skip ----
@pytest.mark.parametrize("n,k,l,kernel", [(600_000, 16, 16, "seg_pass_kernel<4,4,4>"), (301_000, 100, 4, "seg_pass_kernel<32,4,4>"),
                                          (301_000, 200, 4, "seg_pass_kernel<64,4,4>")])
def test_triple_passes_four_rows_in_flight(hip, n, k, l, kernel):
    # (rows of up to 16 groups: the four-launch form takes over from the two-launch one at ratings x (K + L) > 14M -- 18M for data
    # with cut segments, which 37 ratings per pair are)
    data, (n_u, n_i, n_r) = uniform(n, 30_000, 3_000, 4, seed=k)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(5, n_u, n_i, n_r, k, l, d_u, d_i)
    with LaunchWindow() as lw:
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as em:
            em.set_params(theta, eta, pr)
            assert em.get_option("launches") == 4.0
            t, e, p = check_step_and_loop(em, data, theta, eta, pr, d_u, d_i, checker=fact, what=kernel)
            assert em.likelihood() == pytest.approx(float(fact.compute_likelihood(data, t, e, p)), rel=1e-11)
        assert kernel in lw.names()


# ---- restart slots sharing the index stream: super-groups of 4, 8 and 16 slots (narrow rows), 4 slots at K = 20 ----
@pytest.mark.parametrize("k,l,slots,kernel", [(10, 10, 3, "seg_pass_slots_kernel<4,4,4,4>"), (10, 10, 6, "seg_pass_slots_kernel<4,4,4,8>"),
                                              (10, 10, 11, "seg_pass_slots_kernel<4,4,4,16>"), (20, 20, 4, "seg_pass_slots_kernel<8,4,4,4>")])
def test_restart_slots_in_super_groups(hip, k, l, slots, kernel):
    data, (n_u, n_i, n_r) = uniform(4000, 300, 120, 5, seed=slots)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    starts = [orc.init_params(100 + s, n_u, n_i, n_r, k, l, d_u, d_i) for s in range(slots)]
    with LaunchWindow() as lw:
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0, slots=slots) as em:
            em.set_option("fused", 0)                     # the four-launch form: its triple passes share the index stream
            for s in range(slots):
                em.select(s).set_params(*starts[s])
            em.iterate(3)
            got = [em.select(s).get_params() for s in range(slots)]
        assert kernel in lw.names()
    for s in (0, slots // 2, slots - 1):                  # against the oracle ...
        t, e, p = starts[s]
        for _ in range(3):
            t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
        for g, w, nm in zip(got[s], (t, e, p), ("theta", "eta", "pr")):
            assert rel_err(g, w) < 1e-11, (s, nm)
    with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as one:   # ... and bitwise a one-slot context
        one.set_option("fused", 0)
        one.set_params(*starts[slots - 1])
        one.iterate(3)
        for a, b in zip(one.get_params(), got[slots - 1]):
            assert np.array_equal(a, b)


# ---- the two-launch iteration with unequal row widths on the two sides, more workgroups than CUs, cut user segments ----
@pytest.mark.parametrize("k,l,skew,kernel", [(10, 20, False, "tail_fused_kernel<4,4,8,4,16,false>"),
                                             (20, 10, False, "tail_fused_kernel<8,4,4,4,16,false>"),
                                             (20, 10, True, "tail_fused_kernel<8,4,4,4,16,true>")])
def test_two_launch_tail_with_unequal_sides(hip, k, l, skew, kernel):
    if skew:
        rng = np.random.default_rng(9)
        n = 30_000
        u = np.where(rng.random(n) < 0.3, 11, rng.integers(0, 2000, n))
        data = np.stack([u, rng.integers(0, 300, n), rng.integers(0, 5, n)], axis=1).astype(np.int64)
        for j in range(3):
            data[:, j] = np.unique(data[:, j], return_inverse=True)[1]
        n_u, n_i, n_r = (int(data[:, j].max()) + 1 for j in range(3))
    else:
        data, (n_u, n_i, n_r) = uniform(60_000, 20_000, 2_000, 5, seed=k)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(8, n_u, n_i, n_r, k, l, d_u, d_i)
    with LaunchWindow() as lw:
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as em:
            em.set_params(theta, eta, pr)
            assert em.get_option("launches") == 2.0
            if skew:
                assert em.get_option("splits_users") > 0 and int(em.get_option("fused_split")) & 2
            check_step_and_loop(em, data, theta, eta, pr, d_u, d_i, what=kernel)
            fused = em.get_params()
            em.set_option("fused", 0)                     # bitwise the separate launches
            em.set_params(theta, eta, pr)
            em.iterate(2)
            for a, b in zip(em.get_params(), fused):
                assert np.array_equal(a, b)
        assert kernel in lw.names()


# ---- segments cut into many pieces at every row width (the combine kernels) ----
@pytest.mark.parametrize("k,l,kernel", [(100, 4, "seg_combine_both_kernel<32,4>"), (200, 4, "seg_combine_both_kernel<64,4>"),
                                        (400, 2, "seg_combine_both_kernel<64,8>"), (600, 2, "seg_combine_both_kernel<64,16>")])
def test_split_segments_with_wide_rows(hip, k, l, kernel):
    rng = np.random.default_rng(k)
    n = 7000
    # one user with ~45 % of the rows (more than 32 pieces of 64: combined by a whole workgroup), a few with 100-300
    # (a few pieces: combined by one group of lanes), the rest short -- and one item the same way on the pair side
    u = np.where(rng.random(n) < 0.45, 3, np.where(rng.random(n) < 0.3, rng.integers(4, 9, n), rng.integers(9, 400, n)))
    i = np.where(rng.random(n) < 0.5, 1, rng.integers(0, 60, n))
    data = np.stack([u, i, rng.integers(0, 2, n)], axis=1).astype(np.int64)
    for j in range(3):
        data[:, j] = np.unique(data[:, j], return_inverse=True)[1]
    n_u, n_i, n_r = (int(data[:, j].max()) + 1 for j in range(3))
    d_u, d_i = orc.degrees(data, n_u, n_i)
    assert d_u.max() > 64 * 32 and np.sum((d_u > 64) & (d_u <= 64 * 32)) >= 2
    theta, eta, pr = orc.init_params(4, n_u, n_i, n_r, k, l, d_u, d_i)
    with LaunchWindow() as lw:
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as em:
            em.set_params(theta, eta, pr)
            assert em.get_option("splits_users") >= 3 and em.get_option("splits_pairs") >= 1
            check_step_and_loop(em, data, theta, eta, pr, d_u, d_i, what=kernel)
        assert kernel in lw.names()


# ---- the pair stage of big tiles on the vector ALUs (option mfma = 0: north_star's "no MFMA" form) ----
@pytest.mark.parametrize("k,l,kernels", [
    (256, 24, ["pair_block_kernel<false,true,2,false,256,4,true>"]),     # two slots per thread, tile through scalar loads
    (200, 24, ["pair_block_kernel<false,true,2,true,256,4,true>"]),      # ... tile in LDS
    (92, 92, ["pair_block_kernel<false,true,2,true,512,4,true>"]),       # ... 512 threads
    (128, 132, ["pair_block_kernel<false,true,4,false,512,4,true>"]),    # four slots per thread
    (32, 48, ["pair_quad_a_kernel<12>"]), (32, 56, ["pair_quad_a_kernel<14>"])])   # the persistent A pipeline, 12 / 14 double2 per thread
def test_big_tiles_on_the_vector_alus(hip, k, l, kernels):
    data, (n_u, n_i, n_r) = uniform(3000, 200, 90, 3, seed=k + l)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(21, n_u, n_i, n_r, k, l, d_u, d_i)
    outs = {}
    with LaunchWindow() as lw:
        for on in (0, 1):
            with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as em:
                assert em.get_option("mfma") > 0          # the library's own choice for these tiles: the matrix cores
                if not on:
                    em.set_option("mfma", 0)
                em.set_params(theta, eta, pr)
                if not on:
                    check_step_and_loop(em, data, theta, eta, pr, d_u, d_i, what=kernels[0])
                else:
                    em.iterate(2)
                outs[on] = em.get_params()
        launched = lw.names()
    for kn in kernels:
        assert kn in launched, (kn, sorted(n for n in launched if n.startswith(kn.split("<")[0])))
    for a, b, nm in zip(outs[0], outs[1], ("theta", "eta", "pr")):      # the two forms differ in association order only
        assert rel_err(a, b) < 1e-12, nm


# ---- the likelihood through the logarithm tables: eight lanes per triple, twelve columns per lane (rows of 81-96 groups) ----
def test_likelihood_tables_eight_lanes_twelve_columns(hip):
    k, l = 6, 88
    data, (n_u, n_i, n_r) = uniform(2500, 150, 70, 4, seed=88)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(3, n_u, n_i, n_r, k, l, d_u, d_i)
    for _ in range(3):
        theta, eta, pr = orc.em_step(data, theta, eta, pr, d_u, d_i)
    want = float(orc.compute_likelihood(data, theta, eta, pr))
    with LaunchWindow() as lw:
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as em:
            em.set_params(theta, eta, pr)
            em.set_option("lik_fast", 1)                  # the table form (the wave-per-pair form would take these rows)
            assert em.likelihood() == pytest.approx(want, rel=1e-11)
            em.set_option("lik_fast", 0)                  # ... and a logarithm per element
            assert em.likelihood() == pytest.approx(want, rel=1e-11)
        assert "likelihood_fast_kernel<12,8,true>" in lw.names()


# ---- scoring on the device through the (item, rating) table with rows of more than 256 groups ----
@pytest.mark.parametrize("k,l,kernel", [(300, 3, "predict_rows_kernel<64,8,1>"), (600, 2, "predict_rows_kernel<64,16,1>")])
def test_device_scoring_with_very_wide_rows(hip, k, l, kernel):
    data, (n_u, n_i, n_r) = uniform(2500, 150, 40, 4, seed=k)
    test = orc.synthetic_triples(1200, 150, 40, 4, seed=k + 1)
    test = test[(test[:, 0] < n_u) & (test[:, 1] < n_i) & (test[:, 2] < n_r)]
    d_u, d_i = orc.degrees(data, n_u, n_i)
    runs = []
    for s in range(2):
        t, e, p = orc.init_params(60 + s, n_u, n_i, n_r, k, l, d_u, d_i)
        for _ in range(2):
            t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
        runs.append((t, e, p))
    weights = np.arange(n_r, dtype=np.float64)
    with LaunchWindow() as lw:
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0, slots=2) as em:
            for s in range(2):
                em.select(s).set_params(*runs[s])
            em.predict_begin(test, weights)
            for s in range(2):
                stats = hip.HipEM.final_stats(em.select(s).predict_add())
                ref = orc.score_stats(orc.prod_dist(test, *runs[s]), test[:, 2], list(range(n_r)))
                for key in ("accuracy", "one_off_accuracy", "mae", "s2"):
                    assert stats[key] == ref[key], (s, key, stats, ref)
            mean, _ = em.predict_finish()
        assert kernel in lw.names()
    want = (orc.prod_dist(test, *runs[0]) + orc.prod_dist(test, *runs[1])) / 2
    assert np.allclose(mean, want, rtol=1e-11, atol=1e-15)


# ---- eta_p in 256-thread workgroups (launches of more than two rounds) against the 1,024-thread form ----
@pytest.mark.parametrize("n_r", [5, 10])
def test_eta_p_in_256_thread_workgroups_is_bitwise_the_1024_thread_form(hip, n_r):
    """The form is chosen from the launch size alone -- (p_update + item_sum workgroups) x slots > 2 x CUs -- so the same
    data runs eta_p_kernel with one slot and eta_p_w4_kernel with eight: slot s of the eight must be bit for bit the
    one-slot run of the same start.  p_update there takes two ratings per pass (R = 5: three passes, R = 10: five)."""
    k = l = 10
    data, (n_u, n_i, _) = uniform(100_000, 8_000, 20_000, n_r, seed=n_r)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    starts = [orc.init_params(300 + s, n_u, n_i, n_r, k, l, d_u, d_i) for s in range(8)]
    with LaunchWindow() as lw:
        with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0, slots=8) as em:
            em.set_option("fused", 0)
            for s in range(8):
                em.select(s).set_params(*starts[s])
            em.iterate(3)
            eight = [em.select(s).get_params() for s in (0, 5, 7)]
        assert "eta_p_w4_kernel<4,4>" in lw.names()
    with LaunchWindow() as lw:
        for j, s in enumerate((0, 5, 7)):
            with hip.HipEM(data, k, l, n_u, n_i, n_r, swap_sides=0) as one:
                one.set_option("fused", 0)
                one.set_params(*starts[s])
                one.iterate(3)
                for a, b, nm in zip(one.get_params(), eight[j], ("theta", "eta", "pr")):
                    assert np.array_equal(a, b), (s, nm)
        assert "eta_p_kernel<4,4>" in lw.names() and "eta_p_w4_kernel<4,4>" not in lw.names()
    t, e, p = starts[5]
    for _ in range(3):
        t, e, p = fact.em_step(data, t, e, p, d_u, d_i)
    for g, w, nm in zip(eight[1], (t, e, p), ("theta", "eta", "pr")):
        assert rel_err(g, w) < 1e-11, nm


def test_the_product_refuses_phase_ablation(hip):
    """mmsbm_hip_time_stage's stage bits 8+ (skip phases of the pair stage / of eta_p) exist in the diagnostic build only
    (-DMMSBM_ABLATE, csrc/unity.hip): the shipping kernels carry no such switch."""
    from mmsbm_amd import _lib
    data, (n_u, n_i, n_r) = uniform(2000, 200, 100, 5, seed=1)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    with hip.HipEM(data, 10, 10, n_u, n_i, n_r) as em:
        em.set_params(*orc.init_params(1, n_u, n_i, n_r, 10, 10, d_u, d_i))
        assert em.time_stage(0, 2) > 0
        with pytest.raises(_lib.HipLibraryError) as err:
            em.time_stage(2 | (64 << 8), 2)
        assert err.value.code == _lib.E_UNSUPPORTED and "MMSBM_ABLATE" in str(err.value)

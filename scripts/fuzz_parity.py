"""Randomised parity fuzz (not part of the test suite): random shapes, degrees, slot counts and kernel-family
options against the dense oracle -- numerators after one step, parameters after a few iterations, likelihood in all
three device forms, prod_dist.  usage: python scripts/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import HipEM
from oracle import mmsbm_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rel(a, b):
    m = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (m if m > 0 else 1.0))


t0, cases, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    kind = rng.integers(0, 10)
    if kind < 6:
        k, l = int(rng.integers(1, 26)), int(rng.integers(1, 26))
    elif kind < 9:
        k, l = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    else:
        k, l = (int(rng.integers(200, 1300)), int(rng.integers(1, 6)))[:: 1 if rng.random() < 0.5 else -1]
    n_u, n_i, n_r = int(rng.integers(1, 400)), int(rng.integers(1, 200)), int(rng.integers(1, 9))
    n = int(rng.integers(1, 6000)) if k * l < 4000 else int(rng.integers(1, 600))
    shape = rng.random()
    if shape < 0.25 and k * l <= 700:   # few busy users, popular items (the MovieLens-100k shape in small): segments cut into
        n = int(rng.integers(3000, 30000))   # pieces on both sides -- whole-segment lists of the two-launch form (round 4)
        n_u, n_i = int(rng.integers(20, 400)), int(rng.integers(20, 600))
        pu, pi = rng.lognormal(0, rng.uniform(0.3, 1.5), n_u), rng.lognormal(0, rng.uniform(0.3, 2.0), n_i)
        u_col, i_col = rng.choice(n_u, n, p=pu / pu.sum()), rng.choice(n_i, n, p=pi / pi.sum())
    elif shape < 0.5:   # skewed degrees
        u_col = (rng.zipf(1.3, n) - 1) % n_u
        i_col = (rng.zipf(1.3, n) - 1) % n_i
    else:
        u_col, i_col = rng.integers(0, n_u, n), rng.integers(0, n_i, n)
    data = np.stack([u_col, i_col, rng.integers(0, n_r, n)], axis=1).astype(np.int64)
    d_u, d_i = orc.degrees(data, n_u, n_i)
    theta, eta, pr = orc.init_params(int(rng.integers(0, 1 << 30)), n_u, n_i, n_r, k, l, d_u, d_i)
    if rng.random() < 0.3:   # concentrated memberships: clamps in the likelihood
        theta = theta ** rng.integers(1, 30, theta.shape); eta = eta ** rng.integers(1, 30, eta.shape)
    slots, swap, iters = int(rng.integers(1, 4)), int(rng.integers(-1, 2)), int(rng.integers(1, 4))
    want = orc.update_coefficients(data, theta, eta, pr)
    t, e, p = theta, eta, pr
    for _ in range(iters):
        t, e, p = orc.em_step(data, t, e, p, d_u, d_i)
    lik = float(orc.compute_likelihood(data, t, e, p))
    tag = f"K={k} L={l} N={n} U={n_u} I={n_i} R={n_r} slots={slots} swap={swap} iters={iters}"
    with HipEM(data, k, l, n_u, n_i, n_r, slots=slots, swap_sides=swap) as em:
        fused = em.get_option("fused") + 10 * em.get_option("fused_split")
        if em.get_option("fused") and rng.random() < 0.3:
            em.set_option("fused", 0)
        if em.get_option("mfma") and rng.random() < 0.2:
            em.set_option("mfma", 0)
        sel = int(rng.integers(0, slots))
        for s in range(slots):
            em.select(s).set_params(theta * (1.0 if s == sel else 0.5 + 0.1 * s), eta, pr)
        em.select(sel)
        errs = [rel(g, w) for g, w in zip(em.update_coefficients(), want)]
        em.iterate(iters)
        errs += [rel(g, w) for g, w in zip(em.get_params(), (t, e, p))]
        for mode in (2, 1, 0):
            em.set_option("lik_fast", mode)
            got = em.likelihood()
            # (K = L = 1: the likelihood is exactly 0; every form subtracts per element, as the reference does)
            errs.append(abs(got - lik) / max(abs(lik), 1e-6 * n))
        rows = data[: min(n, 200)]
        errs.append(rel(em.prod_dist(rows), orc.prod_dist(rows, t, e, p)))
    bad = max(errs)
    worst = max(worst, bad)
    cases += 1
    if bad > 1e-10 or not np.isfinite(bad):
        print(f"MISMATCH {tag} fused={fused}: {errs}", flush=True)
        sys.exit(1)
    if cases % 50 == 0:
        print(f"{cases} cases, worst relative error {worst:.2e}  [{time.time() - t0:.0f}s]  last: {tag}", flush=True)
print(f"done: {cases} cases, worst relative error {worst:.2e}")

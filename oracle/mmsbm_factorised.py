"""Factorised float64 CPU checker for the MMSBM EM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` (and scripts that check the product) may import this module; nothing under
``mmsbm_amd/`` does.

Why it exists.  ``oracle/mmsbm_oracle.py`` restates the reference's numpy backend with its dense
(N, K, L) responsibility tensor; at BASELINE's config 5 (10M ratings, K = L = 50) that tensor is 200 GB,
so the dense oracle can only check slices there.  This module evaluates the SAME sums without the
tensor -- SURVEY Appendix A.3, extended by grouping the triples of one (item, rating) "pair" -- so that
every entry of n_theta / n_eta / n_p and of theta / eta / p can be checked at full size on the GPU box's
host in tens of seconds.  For triple n = (u, i, r), P_r = p[:, :, r]  (src/kernels_numpy.py:32-36, :49-52,
:63-77 re-associated):

    A[q, k]      = sum_l P_r[k, l] eta[i, l]                        q = the pair (i, r)
    s_n          = sum_k theta[u, k] A[q_n, k]        w_n = 1 / max(s_n, eps)
    n_theta[u,k] = theta[u, k] sum_{n: u_n = u} w_n A[q_n, k]
    C[q, k]      = sum_{n: q_n = q} w_n theta[u_n, k]
    n_eta[i, l]  = eta[i, l] sum_{q: i_q = i} sum_k P_r[k, l] C[q, k]
    n_p[k, l, r] = P_r[k, l] sum_{q: r_q = r} C[q, k] eta[i_q, l]

It is independent of the HIP library (numpy + scipy.sparse, BLAS products per rating; no shared code, a
different association order) and is PINNED to the dense oracle -- itself pinned bit-exact to the
reference's golden vectors -- by ``tests/test_oracle_golden.py`` (<= 1e-13 on G4 and on C2, element-wise
and in max-norm).

The likelihood follows src/expectation_maximization.py:157-167 exactly:  sum_n sum_kl [w log w - w log s~],
w = max(omega, eps), s~ = max(s, eps).  Triples none of whose K x L elements can be clamped
(fl(fl(min theta_u * min eta_i) * min P_r) >= eps, a monotone lower bound of every element) are summed in
factorised form,

    sum_kl omega log omega = sum_k theta_k log theta_k A[q, k] + sum_k theta_k D[q, k],
    D[q, k] = sum_l P_r[k, l] (eta_l log eta_l) + sum_l (P_r[k, l] log P_r[k, l]) eta_l,

all other triples element by element like the reference.
"""

from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import scipy.sparse as sp

EPS = float(np.finfo(np.float64).eps)  # src/kernels_numpy.py:51, src/expectation_maximization.py:162
ROWS = 1 << 19                         # triples per block of the per-triple gathers
THREADS = max(1, min(16, os.cpu_count() or 1))


class Pairs:
    """The distinct (item, rating) combinations of a training set, rating-major, and each triple's pair."""

    def __init__(self, data, n_users=None, n_items=None, n_ratings=None):
        d = np.asarray(data)
        self.user = d[:, 0].astype(np.int64)
        self.item = d[:, 1].astype(np.int64)
        self.rating = d[:, 2].astype(np.int64)
        self.n_obs = len(d)
        self.n_users = int(self.user.max()) + 1 if n_users is None else int(n_users)
        self.n_items = int(self.item.max()) + 1 if n_items is None else int(n_items)
        self.n_ratings = int(self.rating.max()) + 1 if n_ratings is None else int(n_ratings)
        key = self.rating * self.n_items + self.item
        uniq, self.pair = np.unique(key, return_inverse=True)
        self.pair = self.pair.astype(np.int64).reshape(-1)
        self.pair_item = uniq % self.n_items
        self.pair_rating = uniq // self.n_items
        self.n_pairs = len(uniq)
        # pairs of rating r are the rows rating_off[r] : rating_off[r + 1]
        self.rating_off = np.searchsorted(self.pair_rating, np.arange(self.n_ratings + 1))
        ones = np.ones(self.n_pairs)
        self.item_of_pair = sp.csr_matrix((ones, (self.pair_item, np.arange(self.n_pairs))),
                                          shape=(self.n_items, self.n_pairs))

    def per_rating(self):
        for r in range(self.n_ratings):
            lo, hi = int(self.rating_off[r]), int(self.rating_off[r + 1])
            if hi > lo:
                yield r, slice(lo, hi), self.pair_item[lo:hi]


def _pair_matvec(pairs, eta_like, tile_of):
    """out[q, :] = tile_of(r_q) @ eta_like[i_q, :]   (K-vector per pair)."""
    k = tile_of(0).shape[0]
    out = np.zeros((pairs.n_pairs, k))
    for r, rows, items in pairs.per_rating():
        out[rows] = eta_like[items] @ tile_of(r).T
    return out


def _triple_dots(pairs, tables_u, tables_q):
    """[sum_k tu[u_n, k] * tq[q_n, k] for (tu, tq) in zip(tables_u, tables_q)], block by block (the blocks
    on a few host threads: numpy releases the GIL inside take / einsum)."""
    outs = [np.empty(pairs.n_obs) for _ in tables_u]

    def block(lo):
        sl = slice(lo, min(lo + ROWS, pairs.n_obs))
        u, q = pairs.user[sl], pairs.pair[sl]
        gathered_u, gathered_q = {}, {}
        for out, tu, tq in zip(outs, tables_u, tables_q):
            if id(tu) not in gathered_u:
                gathered_u[id(tu)] = np.take(tu, u, axis=0)
            if id(tq) not in gathered_q:
                gathered_q[id(tq)] = np.take(tq, q, axis=0)
            out[sl] = np.einsum("nk,nk->n", gathered_u[id(tu)], gathered_q[id(tq)])

    starts = list(range(0, pairs.n_obs, ROWS))
    workers = min(len(starts), THREADS)
    if workers <= 1:
        for lo in starts:
            block(lo)
    else:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(block, starts))
    return outs


def update_coefficients(data, theta, eta, pr, pairs=None):
    """(n_theta, n_eta, n_pr), the un-normalised numerators of src/kernels_numpy.py:43-79, factorised."""
    pairs = Pairs(data, theta.shape[0], eta.shape[0], pr.shape[2]) if pairs is None else pairs
    a_tab = _pair_matvec(pairs, eta, lambda r: pr[:, :, r])                   # A[q, k]
    (s,) = _triple_dots(pairs, [theta], [a_tab])
    w = 1.0 / np.maximum(s, EPS)                                              # :49-52 (max, NOT + eps)
    # W[u, q] = sum of w_n over the triples (u, q)  (duplicate rows are separate triples: :63-70 count each)
    w_uq = sp.csr_matrix((w, (pairs.user, pairs.pair)), shape=(pairs.n_users, pairs.n_pairs))
    n_theta = theta * (w_uq @ a_tab)
    c_tab = w_uq.T.tocsr() @ theta                                            # C[q, k]
    t_tab = np.zeros((pairs.n_pairs, eta.shape[1]))
    n_pr = np.zeros_like(pr)
    for r, rows, items in pairs.per_rating():
        t_tab[rows] = c_tab[rows] @ pr[:, :, r]                               # T[q, l]
        n_pr[:, :, r] = pr[:, :, r] * (c_tab[rows].T @ eta[items])            # :73-77
    n_eta = eta * (pairs.item_of_pair @ t_tab)
    return n_theta, n_eta, n_pr


def normalize_with_self(p):
    """src/expectation_maximization.py:152-155 (zero rows divide by 1)."""
    flat = p.reshape(-1, p.shape[2])
    tot = flat.sum(axis=1)
    return (flat / np.where(tot == 0, 1, tot)[:, None]).reshape(p.shape)


def em_step(data, theta, eta, pr, d_u, d_i, pairs=None):
    """One trip of src/mmsbm.py:244-250."""
    n_theta, n_eta, n_pr = update_coefficients(data, theta, eta, pr, pairs)
    return n_theta / np.asarray(d_u)[:, None], n_eta / np.asarray(d_i)[:, None], normalize_with_self(n_pr)


def _xlogx(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(x > 0, x * np.log(x), 0.0)


def compute_likelihood(data, theta, eta, pr, pairs=None):
    """src/expectation_maximization.py:157-167, see the module docstring."""
    pairs = Pairs(data, theta.shape[0], eta.shape[0], pr.shape[2]) if pairs is None else pairs
    lo_bound = (theta.min(axis=1)[pairs.user] * eta.min(axis=1)[pairs.item]) * \
        pr.reshape(-1, pr.shape[2]).min(axis=0)[pairs.rating]
    clear = lo_bound >= EPS                       # no element of these triples is clamped
    total = 0.0
    if clear.any():
        a_tab = _pair_matvec(pairs, eta, lambda r: pr[:, :, r])
        d_tab = _pair_matvec(pairs, _xlogx(eta), lambda r: pr[:, :, r]) + \
            _pair_matvec(pairs, eta, lambda r: _xlogx(pr[:, :, r]))
        s, x1, x2 = _triple_dots(pairs, [theta, _xlogx(theta), theta], [a_tab, a_tab, d_tab])
        s, x = s[clear], (x1 + x2)[clear]
        total += float(np.sum(x - s * np.log(np.maximum(s, EPS))))
    rest = np.flatnonzero(~clear)
    per_r = np.moveaxis(pr, 2, 0)
    step = max(1, (1 << 24) // max(1, theta.shape[1] * eta.shape[1]))
    for lo in range(0, len(rest), step):          # element by element, as the reference does
        idx = rest[lo:lo + step]
        om = (theta[pairs.user[idx], :, None] * eta[pairs.item[idx], None, :]) * per_r[pairs.rating[idx]]
        tot = om.sum(axis=(1, 2))
        wv = np.maximum(om, EPS)
        total += float(np.sum(wv * np.log(wv) - wv * np.log(np.maximum(tot, EPS))[:, None, None]))
    return np.float64(total)


def prod_dist(data, theta, eta, pr):
    """P[n, r] = theta_u . (P_r eta_i)  (src/kernels_numpy.py:86-96), block by block."""
    d = np.asarray(data)
    out = np.empty((len(d), pr.shape[2]))
    for lo in range(0, len(d), ROWS):
        sl = slice(lo, min(lo + ROWS, len(d)))
        tu, ei = theta[d[sl, 0]], eta[d[sl, 1]]
        for r in range(pr.shape[2]):
            out[sl, r] = np.einsum("nk,nk->n", tu, ei @ pr[:, :, r].T)
    return out

"""numpy emulation of the DEVICE algorithm (the factorised form in mmsbm_hip.hip), driven by
the layout arrays the library builds.  Test infrastructure: it lets the CPU suite check the
re-association and the sorted layout against the oracle without a GPU."""
import numpy as np

EPS = float(np.finfo(np.float64).eps)


def pair_ratings(lay):
    q = np.arange(len(lay["pair_item"]))
    return np.searchsorted(lay["rating_off"], q, side="right") - 1


def a_table(lay, eta, pr):
    rq = pair_ratings(lay)
    return np.einsum("qkl,ql->qk", np.moveaxis(pr, 2, 0)[rq], eta[lay["pair_item"]])


def iteration(lay, theta, eta, pr):
    """Un-normalised (n_theta, n_eta, n_pr) exactly as the kernels associate the sums."""
    n_users, n_items = len(lay["user_off"]) - 1, len(lay["item_off"]) - 1
    n_pairs = len(lay["pair_item"])
    rq = pair_ratings(lay)
    A = a_table(lay, eta, pr)
    # seg_pass, user segments
    u_of = np.repeat(np.arange(n_users), np.diff(lay["user_off"]))
    q = lay["user_pair"]
    w = 1.0 / np.maximum((theta[u_of] * A[q]).sum(axis=1), EPS)
    acc = np.zeros_like(theta)
    np.add.at(acc, u_of, A[q] * w[:, None])
    n_theta = theta * acc
    # seg_pass, pair segments
    q_of = np.repeat(np.arange(n_pairs), np.diff(lay["pair_off"]))
    u = lay["pair_user"]
    w = 1.0 / np.maximum((theta[u] * A[q_of]).sum(axis=1), EPS)
    C = np.zeros((n_pairs, theta.shape[1]))
    np.add.at(C, q_of, theta[u] * w[:, None])
    # pair_matvec T, item_sum
    T = np.einsum("qkl,qk->ql", np.moveaxis(pr, 2, 0)[rq], C)
    i_of = np.repeat(np.arange(n_items), np.diff(lay["item_off"]))
    acc = np.zeros_like(eta)
    np.add.at(acc, i_of, T[lay["item_pairs"]])
    n_eta = eta * acc
    # p_partial per chunk, p_finalize
    n_pr = np.zeros_like(pr)
    for r, qb, qe, _ in lay["chunks"]:
        n_pr[:, :, r] += C[qb:qe].T @ eta[lay["pair_item"][qb:qe]]
    n_pr *= pr
    return n_theta, n_eta, n_pr

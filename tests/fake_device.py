"""A CPU stand-in for ``mmsbm_amd.core.HipEM`` built on the oracle: same Python surface
(slots, select, set/get_params, iterate, likelihood, prod_dist, predict_begin/add/finish).

TEST INFRASTRUCTURE ONLY.  It lets the CPU suite exercise the host class's orchestration --
restart batching, slot residency, fold lanes, the convergence monitor, the predict/score flow
-- with the reference's arithmetic underneath, so those paths can be pinned to the reference's
golden outputs without a GPU.  Nothing under ``mmsbm_amd/`` imports it."""
import numpy as np

from oracle import mmsbm_oracle as orc

LOG = []  # (event, detail) tuples, for tests that check WHAT the host class asked the device to do


class FakeHipEM:
    STAT_NAMES = ("rows", "true", "almost", "s2", "true_pond", "s2pond")

    def __init__(self, data, k_groups, l_groups, n_users=None, n_items=None, n_ratings=None,
                 device=0, swap_sides=-1, slots=1):
        self.data = np.asarray(data, dtype=np.int64)
        self.k, self.l = int(k_groups), int(l_groups)
        self.n_users, self.n_items, self.n_ratings = int(n_users), int(n_items), int(n_ratings)
        self.n_obs = len(self.data)
        self.device = int(device)
        self.closed = False
        self.set_slots(slots)
        LOG.append(("create", self.device))

    # -- slots
    def set_slots(self, n):
        if n < 1:
            raise ValueError("n_slots")
        self.slots, self._sel = int(n), 0
        self._params = [None] * self.slots
        LOG.append(("set_slots", int(n)))

    def select(self, s):
        if not 0 <= s < self.slots:
            raise IndexError(s)
        self._sel = int(s)
        return self

    @property
    def selected(self):
        return self._sel

    def result(self):
        return (self.likelihood(),) + tuple(self.get_params())

    def max_slots(self, fraction=0.5, sharers=1):
        return getattr(FakeHipEM, "MAX_SLOTS", 1 << 20)

    def suggested_slots(self, most=8):
        return getattr(FakeHipEM, "SUGGESTED_SLOTS", most)

    # -- parameters
    def degrees(self):
        return orc.degrees(self.data, self.n_users, self.n_items)

    def set_params(self, theta, eta, pr):
        self._params[self._sel] = tuple(np.array(a, dtype=np.float64) for a in (theta, eta, pr))
        LOG.append(("set_params", self._sel))

    def init_params(self, seed):
        d_u, d_i = self.degrees()
        t, e, p = orc.init_params(seed, self.n_users, self.n_items, self.n_ratings, self.k, self.l, d_u, d_i)
        self.set_params(t, e, p)
        return p

    def get_params(self):
        return tuple(a.copy() for a in self._params[self._sel])

    def iterate(self, n, sync=True):
        assert all(p is not None for p in self._params), "a slot has no parameters"
        d_u, d_i = self.degrees()
        for s in range(self.slots):
            t, e, p = self._params[s]
            for _ in range(int(n)):
                t, e, p = orc.em_step(self.data, t, e, p, d_u, d_i)
            self._params[s] = (t, e, p)
        LOG.append(("iterate", int(n)))

    def synchronize(self):
        pass

    def likelihood(self):
        return np.float64(orc.compute_likelihood(self.data, *self._params[self._sel]))

    # -- level 1 (what mmsbm_amd/kernels_hip.py asks of a context)
    def compute_omegas(self):
        return orc.compute_omegas(self.data, *self._params[self._sel])

    def update_coefficients(self):
        LOG.append(("update_coefficients", self._sel))
        return orc.update_coefficients(self.data, *self._params[self._sel])

    def prod_dist(self, pairs):
        return orc.prod_dist(np.asarray(pairs), *self._params[self._sel])

    # -- predict / score session
    @staticmethod
    def final_stats(raw):
        n = raw[0]
        with np.errstate(divide="ignore", invalid="ignore"):
            return {"accuracy": np.float64(raw[1]) / n, "one_off_accuracy": np.float64(raw[2]) / n,
                    "mae": 1 - np.float64(raw[4]) / n, "s2": np.int64(raw[3]), "s2pond": np.float64(raw[5])}

    def _raw(self, rat):
        real = self._test[:, 2]
        ok = rat.sum(axis=1) != 0
        rat, real = rat[ok], real[ok]
        pred = np.argmax(rat, axis=1) if len(rat) else np.zeros(0, dtype=np.int64)
        pond = rat @ self._weights
        d = np.abs(pred - real)
        return np.array([ok.sum(), (d == 0).sum(), (d <= 1).sum(), d.sum(),
                         (real == np.round(pond)).sum(), np.abs(pond - real).sum()], dtype=np.float64)

    def predict_begin(self, test, rating_weights):
        self._test = np.asarray(test, dtype=np.int64)
        self._weights = np.asarray(rating_weights, dtype=np.float64)
        self._rats = []

    def predict_add(self):
        rat = self.prod_dist(self._test)
        self._rats.append(rat)
        LOG.append(("predict_add", self._sel))
        return self._raw(rat)

    def predict_finish(self, want_matrix=True):
        mean = np.array(self._rats).mean(axis=0)
        return (mean if want_matrix else None), self._raw(mean)

    def close(self):
        self.closed = True

"""Host class with the surface of the reference's ``MMSBM`` (src/mmsbm.py:15-553) for
``backend='hip'``: same constructor keywords, ``fit`` / ``predict`` / ``score`` / ``cv_fit``,
``results`` as a list of ``{"likelihood", "pr", "theta", "eta"}`` dicts in restart order.

What differs from the reference is where things run: the whole EM loop of a restart stays on
the GPU (src/mmsbm.py:243-250 becomes one C-ABI call), the restarts that share a GPU advance
together as the slots of one ``HipEM`` context (one set of kernel launches per iteration
for all of them: the batching the reference lists as a TODO, README.md:188), and restarts
are spread over GPUs -- threads over ``devices`` inside one process, or ranks of a
``torch.distributed`` job (mmsbm_amd/restarts.py) -- instead of a ``multiprocessing.Pool``
(src/mmsbm.py:182-185).  Restart ``i`` is seeded exactly like the reference
(``SeedSequence(seed).spawn(sampling)[i]``, draw order theta, eta, p) and slots never
interact, so its result does not depend on ``sampling``, on the batch it ran in or on
which GPU ran it.
"""
from __future__ import annotations

import logging
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime

import numpy as np

from .backend import load_backend
from ._lib import HipLibraryError
from .core import HipEM, data_key, normalize_with_self  # noqa: F401  (re-exported)
from .encode import Encoder


class MMSBM:
    data_handler = None
    results = None
    test = None
    theta = None
    eta = None
    pr = None
    likelihood = None
    prediction_matrix = None
    rng = None

    def __init__(self, user_groups, item_groups, iterations=400, sampling=1, seed=None,
                 debug=False, backend="auto", devices=None, restarts_per_launch=None,
                 contexts_per_device=1, tol=None, check_every=50):
        self.start_time = datetime.now()
        self.user_groups = user_groups
        self.item_groups = item_groups
        self.iterations = iterations
        self.sampling = sampling
        self.debug = debug
        self.backend = backend
        self.devices = devices
        # Restarts that share a GPU run as slots of one context, up to this many per batch
        # (C3: 1.2x the restarts/s of one at a time, C2: 3x, C1-sized problems: ~Sx).
        # restarts advanced together as slots of one context; None: HipEM.suggested_slots() decides from the table sizes
        self.restarts_per_launch = None if restarts_per_launch is None else max(1, int(restarts_per_launch))
        # More than one context (= stream) per GPU is possible too; slots do the same job
        # better, so the default is one.
        self.contexts_per_device = max(1, int(contexts_per_device))
        # Convergence monitor (not in the reference, off by default): every `check_every`
        # iterations the likelihood of each restart of the batch is evaluated -- the reference's
        # debug hook, src/mmsbm.py:252-254 -- and with `tol` set the batch stops early once every
        # restart's relative change is below it.  `iterations_run[i]` records what restart i got.
        self.tol = tol
        self.check_every = max(1, int(check_every))
        self.iterations_run = {}
        # src/mmsbm.py:81-85
        self.rng = np.random.default_rng(seed)
        self.child_states = self.rng.bit_generator._seed_seq.spawn(sampling)
        self.logger = logging.getLogger("MMSBM")
        self._backend = None  # resolved in _prepare_objects, like the reference (EM ctor)
        self.best_by_likelihood = None
        self._ctxs = {}
        self._resident = {}      # (device, context) -> restart ids whose final parameters sit in its slots
        self._scored = None      # (prediction matrix, raw device sums) of the last predict()

    # ------------------------------------------------------------------ preparation
    def _prepare_objects(self, train):
        """Dims only; the degrees come from the device layout (src/mmsbm.py:93-146 minus the
        dead O(U*N) index lists)."""
        # 'auto'/'hip' -> hip; anything else raises ImportError like src/backend.py:27-28
        *_, self._backend = load_backend(self.backend)
        train = np.asarray(train)
        self.train = train
        # = sorted(set(train[:, 2])), src/mmsbm.py:95 (ids are small non-negative ints: one counting pass)
        self.ratings = (np.flatnonzero(np.bincount(train[:, 2])).tolist() if len(train) else [])
        self.r = max(self.ratings)
        self.p = int(train[:, 0].max())
        self.m = int(train[:, 1].max())
        self._dims = {"n_samples": len(train), "n_user_groups": self.user_groups,
                      "n_item_groups": self.item_groups, "n_ratings": len(self.ratings)}
        self._release()

    def _device_list(self):
        if self.devices is not None:
            return list(dict.fromkeys(int(d) for d in self.devices))  # unique, order kept
        return [0]

    def _ctx(self, device, slot=0):
        ctx = self._ctxs.get((device, slot))
        if ctx is None:
            ctx = HipEM(self.train, self.user_groups, self.item_groups, n_users=self.p + 1,
                        n_items=self.m + 1, n_ratings=self._dims["n_ratings"], device=device)
            self._ctxs[(device, slot)] = ctx
        return ctx

    def _sharers(self, device):
        """Workers that may size a batch of restart slots on `device` at the same time."""
        lanes = list(self.devices) if self.devices is not None else [0]
        return max(1, self.contexts_per_device * max(1, sum(1 for d in lanes if int(d) == int(device))))

    def _release(self):
        for ctx in self._ctxs.values():
            ctx.close()
        self._ctxs = {}
        self._resident = {}

    # ------------------------------------------------------------------ training
    def fit(self, data, silent=False):
        if not silent:
            self.logger.info(f"Running {self.sampling} runs of {self.iterations} iterations.")
        self.data_handler = Encoder()
        train = self.data_handler.fit_transform(data)
        self.fit_encoded(train)

    def fit_encoded(self, train, restarts=None):
        """fit() on already encoded (N,3) triples.  ``restarts``: subset of restart indices to
        run here (used by the multi-GPU driver); default all."""
        self._prepare_objects(train)
        todo = list(range(self.sampling)) if restarts is None else list(restarts)
        # workers = (GPU, context slot); restart j of `todo` goes to worker j mod #workers
        workers = [(d, s) for s in range(self.contexts_per_device) for d in self._device_list()]
        workers = workers[:max(1, len(todo))]

        def work(w):  # this worker's restarts, in batches of slots
            dev, slot = workers[w]
            mine, out = todo[w::len(workers)], []
            per = self.restarts_per_launch or self._ctx(dev, slot).suggested_slots()
            for b in range(0, len(mine), per):
                batch = mine[b:b + per]
                out.extend(zip(batch, self.run_samplings(batch, device=dev, slot=slot)))
            return out

        if len(workers) == 1:
            parts = [work(0)]
        else:  # one host thread per worker; ctypes releases the GIL inside the library
            with ThreadPoolExecutor(max_workers=len(workers)) as pool:
                parts = list(pool.map(work, range(len(workers))))
        by_i = dict(x for part in parts for x in part)
        done = [by_i[i] for i in todo]
        self.results = done
        self._restart_ids = todo
        liks = [float(r["likelihood"]) for r in done]
        self.best_by_likelihood = todo[int(np.argmax(liks))] if liks else None
        return self

    def init_params(self, seed, d_u, d_i):
        """theta0, eta0, p0 with the reference's draw order (src/mmsbm.py:224-233)."""
        rng = np.random.default_rng(seed)
        k, l, r = self.user_groups, self.item_groups, self._dims["n_ratings"]
        theta = rng.random((self.p + 1, k)) / d_u[:, None]
        eta = rng.random((self.m + 1, l)) / d_i[:, None]
        pr = normalize_with_self(rng.random((k, l, r)))
        return theta, eta, pr

    def run_samplings(self, ids, device=0, slot=0, seeds=None):
        """Restarts ``ids`` together, device resident, as the slots of one context
        (src/mmsbm.py:187-269 for each of them).  Returns their result dicts in order."""
        ctx = self._ctx(device, slot)
        ids = list(ids)
        seeds = [self.child_states[i] for i in ids] if seeds is None else list(seeds)
        if len(ids) > 1:
            # memory: what is FREE on the device now, shared with the other workers on this GPU
            fit_in = ctx.max_slots(0.5, sharers=self._sharers(device))
            if len(ids) > fit_in:  # run what fits, then the rest
                return (self.run_samplings(ids[:fit_in], device, slot, seeds[:fit_in]) +
                        self.run_samplings(ids[fit_in:], device, slot, seeds[fit_in:]))
        try:
            ctx.set_slots(len(ids))
        except HipLibraryError:
            # the device ran out of memory after all (another worker got there first): the
            # context is back to one slot; halve the batch and try again
            if len(ids) == 1:
                raise
            half = len(ids) // 2
            return (self.run_samplings(ids[:half], device, slot, seeds[:half]) +
                    self.run_samplings(ids[half:], device, slot, seeds[half:]))
        for s, seed in enumerate(seeds):  # theta0, eta0 are drawn on the device (same PCG64 stream)
            ctx.select(s).init_params(seed)
        done = 0
        if self.debug or self.tol is not None:
            # src/mmsbm.py:252-254 evaluates the likelihood inside the loop when j % 50 == 0, i.e.
            # after iterations 1, 51, 101, ...; the convergence monitor (tol) checks every
            # `check_every` iterations instead.
            last = None
            while done < self.iterations:
                if self.tol is None:
                    step = min(1 if done == 0 else 50, self.iterations - done)
                else:
                    step = min(self.check_every, self.iterations - done)
                ctx.iterate(step)
                done += step
                if self.tol is None and (done - 1) % 50 != 0:
                    break  # the tail after the last hook: the reference logs nothing there
                liks = np.array([ctx.select(s).likelihood() for s in range(len(ids))])
                if self.debug:
                    for i, lik in zip(ids, liks):
                        self.logger.debug(f"\nLikelihood at run {i} is {lik:.0f}")
                if self.tol is not None and last is not None and np.all(
                        np.abs(liks - last) <= self.tol * np.abs(last)):
                    break
                last = liks
        else:
            ctx.iterate(self.iterations)
            done = self.iterations
        for i in ids:
            self.iterations_run[i] = done
        out = []
        for s in range(len(ids)):
            likelihood, theta, eta, pr = ctx.select(s).result()
            out.append({"likelihood": likelihood, "pr": pr, "theta": theta, "eta": eta})
        self._resident[(device, slot)] = ids
        return out

    def run_one_sampling(self, data, seed, i, device=0, slot=0):
        """One restart, device resident (src/mmsbm.py:187-269)."""
        return self.run_samplings([i], device, slot, seeds=[seed])[0]

    # ------------------------------------------------------------------ prediction
    def _check_is_fitted(self):
        assert self.results is not None, "You need to fit the model before predicting."

    def _check_has_predictions(self):
        assert self.prediction_matrix is not None, (
            "You need to predict before computing the goodness of fit parameters.")

    def predict(self, data):
        """Mean of prod_dist over restarts; stored objects from the run with the best test
        accuracy (src/mmsbm.py:279-317).  Everything per test row -- the distributions, their
        running sum over restarts, argmax and the indicators of src/mmsbm.py:488-528 -- is
        evaluated on the device; per restart only six sums come back."""
        self._check_is_fitted()
        if len(self._restart_ids) != self.sampling:
            # restarts.fit_distributed(gather=False) left this rank with ITS share only: a mean over that share
            # would silently differ from rank to rank (the reference averages over all restarts, src/mmsbm.py:315)
            raise RuntimeError(
                f"this model holds {len(self._restart_ids)} of its {self.sampling} restarts (restarts.fit_distributed "
                "without gather=True): use mmsbm_amd.restarts.predict_distributed(model, data), or fit with gather=True")
        test = self.data_handler.transform(data, self.logger)
        matrix, raw, per_run = self._predict_runs(test)
        self.run_stats = per_run
        self._keep_best_run(int(np.argmax([st["accuracy"] for st in per_run])))  # first best restart, src/mmsbm.py:474-478
        self.prediction_matrix = matrix
        self._scored = (matrix, raw)
        return self.prediction_matrix

    def _predict_runs(self, test, subset=None):
        """The restarts held by THIS model (all of them, or this rank's share) on encoded test triples:
        (mean distribution over them, its six sums, the five scores of every restart).  ``subset``:
        positions in ``self.results`` to score instead of all of them (restarts.predict_distributed)."""
        self.test = test
        dev = self._device_list()[0]
        ctx = self._ctx(dev)
        picked = list(range(len(self.results))) if subset is None else list(subset)
        # restarts whose final parameters still sit in this context's slots need no upload
        resident = self._resident.get((dev, 0)) == list(self._restart_ids) and ctx.slots == len(self.results)
        if not resident:
            ctx.set_slots(1)
            self._resident.pop((dev, 0), None)
        ctx.predict_begin(test, np.asarray(self.ratings, dtype=np.float64))
        per_run = []
        for j in picked:
            a = self.results[j]
            if resident:
                ctx.select(j)
            else:
                ctx.set_params(a["theta"], a["eta"], a["pr"])
            per_run.append(ctx.predict_add())
        matrix, raw = ctx.predict_finish()
        self._raw_per_run = per_run                  # the six sums of each scored restart (restarts.predict_distributed)
        return matrix, raw, [ctx.final_stats(st) for st in per_run]

    def _keep_best_run(self, best, res=None):
        """theta / eta / pr / likelihood of restart ``best`` become the model's stored objects
        (src/mmsbm.py:303-311); ``res``: its result dict when it is not in self.results."""
        import pandas as pd
        enc = self.data_handler  # (None after fit_encoded: rows and ratings keep their integer ids)
        res = self.results[best] if res is None else res
        self.theta = pd.DataFrame(res["theta"], index=enc.user_labels() if enc else None)
        self.eta = pd.DataFrame(res["eta"], index=enc.item_labels() if enc else None)
        labels = enc.rating_labels() if enc else range(res["pr"].shape[2])
        self.pr = {lab: pd.DataFrame(res["pr"][:, :, j]) for j, lab in enumerate(labels)}
        self.likelihood = np.float64(res["likelihood"])

    def choose_best_run(self, rats):
        """Index of the first prediction matrix with the highest accuracy (src/mmsbm.py:474-478)."""
        return int(np.argmax([self._compute_stats(a)["accuracy"] for a in rats]))

    # ------------------------------------------------------------------ scoring (src/mmsbm.py:319-369,488-539)
    def score(self, silent=False):
        self._check_has_predictions()
        if self._scored is not None and self._scored[0] is self.prediction_matrix:
            stats = HipEM.final_stats(self._scored[1])  # reduced on the device by predict()
        else:  # a matrix the caller supplied
            stats = self._compute_stats(self.prediction_matrix)
        stats["likelihood"] = self.likelihood
        if not silent:
            self.logger.info(
                f"The final accuracy is {stats['accuracy']}, the one off accuracy is "
                f"{stats['one_off_accuracy']} and the MAE is {stats['mae']}.")
        return {"stats": stats, "objects": {"theta": self.theta, "eta": self.eta, "pr": self.pr}}

    def _compute_stats(self, rat):
        """Scores of a prediction matrix the CALLER supplies (the model's own predictions are scored on
        the device, predict_score_kernel): the same six sums the device returns -- rows with any mass,
        exact hits, hits within one class, |error|, hits and |error| of the weighted prediction
        (src/mmsbm.py:488-528) -- turned into the five scores by HipEM.final_stats."""
        rat = np.asarray(rat, dtype=np.float64)
        truth = np.asarray(self.test[:, 2])
        keep = rat.sum(axis=1) != 0                       # rows without a prediction do not count
        err = np.abs(np.argmax(rat, axis=1) - truth)[keep]
        weighted = (rat @ self.ratings)[keep]
        real = truth[keep]
        raw = (int(keep.sum()), int((err == 0).sum()), int((err <= 1).sum()), int(err.sum()),
               int((np.round(weighted) == real).sum()), float(np.abs(weighted - real).sum()))
        return HipEM.final_stats(raw)

    def compute_likelihood(self, data, theta, eta, pr):
        """src/mmsbm.py:541-553: the likelihood of ``data`` (encoded triples -- the training set
        or any other, e.g. a held-out split) under explicit parameters, evaluated on the device.
        The training set re-uses the resident context; other data gets a context of its own for
        the call."""
        theta, eta, pr = (np.asarray(a, dtype=np.float64) for a in (theta, eta, pr))
        dev = self._device_list()[0]
        train = getattr(self, "train", None)
        if train is not None and (data is train or data_key(data) == self._train_key()):
            ctx = self._ctx(dev)
            self._resident.pop((dev, 0), None)  # slot 0 is overwritten
            ctx.select(0).set_params(theta, eta, pr)
            return ctx.likelihood()
        ctx = HipEM(data, theta.shape[1], eta.shape[1], n_users=theta.shape[0], n_items=eta.shape[0],
                    n_ratings=pr.shape[2], device=dev)
        try:
            ctx.set_params(theta, eta, pr)
            return ctx.likelihood()
        finally:
            ctx.close()

    def _train_key(self):
        if getattr(self, "_train_key_cache", None) is None or self._train_key_cache[0] is not self.train:
            self._train_key_cache = (self.train, data_key(self.train))
        return self._train_key_cache[1]

    # ------------------------------------------------------------------ cross-validation (src/mmsbm.py:371-472)
    def cv_fit(self, data, folds=5):
        """src/mmsbm.py:371-472.  The fold splits are drawn first, in the reference's order (the
        model's RNG is used for nothing else), so the folds are independent jobs: with several
        entries in ``devices`` they run concurrently, one fold per entry at a time (an entry may
        repeat a GPU), each fold's restarts batched on its GPU."""
        n_items = len(set(data.iloc[:, 1]))
        assert folds <= n_items, (
            f"Fold number can't be higher than {n_items} since this is the number of different "
            f"items you have.")
        per_fold = int(n_items / folds)
        temp = data
        splits = []
        for f in range(folds):
            picked = []
            for _, grp in temp.groupby(temp.columns[0]):
                for cnt in range(per_fold, 0, -1):  # as many as the user has, at most per_fold
                    if cnt <= len(grp.index):
                        picked.extend(self.rng.choice(grp.index, cnt, replace=False).tolist())
                        break
            picked = [a for a in picked if str(a) != "0"]
            test = temp.loc[picked, :]
            splits.append((data[~data.index.isin(test.index)], test))
            temp = temp[~temp.index.isin(picked)]

        def run_fold(model, f):
            self.logger.info(f"Running fold {f + 1} of {folds}...")
            train, test = splits[f]
            model.fit(train, silent=True)
            model.prediction_matrix = model.predict(test)
            results = model.score(silent=True)
            return {"stats": results["stats"],
                    "objects": {"theta": model.theta, "eta": model.eta, "pr": model.pr,
                                "rat": model.prediction_matrix}}

        lanes = [int(d) for d in self.devices] if self.devices is not None else [0]
        if len(lanes) == 1 or folds == 1:
            all_results = [run_fold(self, f) for f in range(folds)]
        else:
            def lane(j):  # folds j, j + lanes, ... on GPU lanes[j], through a private model
                child = self._fold_model(lanes[j])
                try:
                    return [(f, run_fold(child, f)) for f in range(j, folds, len(lanes))]
                finally:
                    child._release()
            with ThreadPoolExecutor(max_workers=len(lanes)) as pool:
                parts = list(pool.map(lane, range(min(len(lanes), folds))))
            by_f = dict(x for part in parts for x in part)
            all_results = [by_f[f] for f in range(folds)]
        accuracies = [a["stats"]["accuracy"] for a in all_results]
        kept = all_results[int(np.argmax(accuracies))]["objects"]   # the first best fold (src/mmsbm.py:460-465)
        self.theta, self.eta, self.pr, self.prediction_matrix = (kept[k] for k in ("theta", "eta", "pr", "rat"))
        self.cv_results = all_results
        self.logger.info(f"Ran {folds} folds with accuracies {accuracies}.")
        return accuracies

    def _fold_model(self, device):
        """A model with this one's settings and restart seeds, bound to one GPU."""
        child = MMSBM(self.user_groups, self.item_groups, iterations=self.iterations,
                      sampling=self.sampling, debug=self.debug, backend=self.backend,
                      devices=[device], restarts_per_launch=self.restarts_per_launch,
                      contexts_per_device=self.contexts_per_device, tol=self.tol,
                      check_every=self.check_every)
        child.child_states = self.child_states  # every fold restarts from the same seeds (src/mmsbm.py:441)
        child.logger = self.logger
        return child

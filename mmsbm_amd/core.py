"""Thin Python handle over the resident C-ABI context (include/mmsbm_hip.h, "level 2").

One ``HipEM`` = one GPU + one encoded training set.  Parameters stay on the device
between calls; the EM loop of src/mmsbm.py:243-250 runs there without host round trips.
"""
from __future__ import annotations

import ctypes as C
import hashlib

import numpy as np

from . import _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def normalize_with_self(p):
    """p[k,l,:] /= sum_r p[k,l,r]; zero rows stay zero (src/expectation_maximization.py:152-155).
    Host-side, used for the random initialisation only."""
    flat = p.reshape(-1, p.shape[2])
    tot = flat.sum(axis=1)
    return (flat / np.where(tot == 0, 1, tot)[:, None]).reshape(p.shape)


def pcg64_words(bit_generator):
    """{state_hi, state_lo, inc_hi, inc_lo} of a numpy PCG64, as the C ABI takes them."""
    st = bit_generator.state["state"]
    mask = (1 << 64) - 1
    return (C.c_uint64 * 4)(st["state"] >> 64, st["state"] & mask, st["inc"] >> 64, st["inc"] & mask)


def pcg64_doubles(seed, offset, n):
    """n doubles of default_rng(seed)'s stream starting `offset` draws in, from the library's own
    generator (host code; no GPU needed)."""
    out = np.empty(int(n), dtype=np.float64)
    _lib.call("mmsbm_hip_pcg64_doubles", pcg64_words(np.random.PCG64(seed)), int(offset), int(n),
              _p(out, C.c_double))
    return out


def split_triples(data):
    """(N,3) integer array (any int dtype, any strides) -> three contiguous int32 columns."""
    d = np.asarray(data)
    if d.ndim != 2 or d.shape[1] < 3:
        raise ValueError("data must have shape (N, 3): [user_idx, item_idx, rating_idx]")
    if d.shape[0] and not np.issubdtype(d.dtype, np.integer):
        raise TypeError("data must hold integer ids")
    if d.size and (d.min() < 0 or d.max() >= 2**31):
        raise ValueError("ids must be in [0, 2^31)")
    return _i32(d[:, 0]), _i32(d[:, 1]), _i32(d[:, 2])


try:  # 128-bit XXH3: ~10 GB/s (about 2.4 ms per million (N,3) int64 rows); blake2b does ~1 GB/s
    from xxhash import xxh3_128 as _digest128
    DIGEST = "xxh3_128"
except ImportError:  # pragma: no cover - depends on the environment
    DIGEST = "blake2b"
    _warned = []

    def _digest128(buf):
        # `xxhash` is an optional dependency (requirements.txt lists it): without it the digest of a level-1 call is
        # ten times slower and no longer hides behind the GPU work (level 1) -- say so, once
        if not _warned:
            _warned.append(True)
            import warnings
            warnings.warn("mmsbm_amd: the `xxhash` module is not installed; the training-set digest of the level-1 "
                          "backend (kernels_hip) falls back to blake2b, about ten times slower per call",
                          RuntimeWarning, stacklevel=3)
        return hashlib.blake2b(buf, digest_size=16)


def data_key(data):
    """Exact identity of a set of encoded triples: (shape, dtype, 128-bit digest over EVERY byte of
    the first three columns).  Two training sets that differ anywhere -- two rows swapped, two
    ratings exchanged, a re-shuffled fold -- get different keys (the level-1 cache of kernels_hip
    and ``MMSBM.compute_likelihood`` rely on that).  Pure host code."""
    d = np.asarray(data)
    if d.ndim != 2 or d.shape[1] < 3:
        raise ValueError("data must have shape (N, 3): [user_idx, item_idx, rating_idx]")
    d = np.ascontiguousarray(d[:, :3])
    return (d.shape, d.dtype.str, _digest128(memoryview(d).cast("B")).hexdigest())


class HipEM:
    """Device-resident EM state for one (GPU, training set, K, L)."""

    def __init__(self, data, k_groups, l_groups, n_users=None, n_items=None, n_ratings=None,
                 device=0, swap_sides=-1, slots=1):
        u, i, r = split_triples(data)
        n = len(u)
        self.n_obs = n
        self.n_users = int(n_users) if n_users is not None else (int(u.max()) + 1 if n else 1)
        self.n_items = int(n_items) if n_items is not None else (int(i.max()) + 1 if n else 1)
        self.n_ratings = int(n_ratings) if n_ratings is not None else (int(r.max()) + 1 if n else 1)
        self.k, self.l = int(k_groups), int(l_groups)
        self.device = int(device)
        self._h = C.c_void_p()
        _lib.call("mmsbm_hip_create", self.device, n, self.n_users, self.n_items, self.n_ratings,
                  self.k, self.l, _p(u, C.c_int32), _p(i, C.c_int32), _p(r, C.c_int32),
                  int(swap_sides), C.byref(self._h))
        dims = (C.c_int64 * 8)()
        _lib.call("mmsbm_hip_dims", self._h, dims)
        self.n_pairs, self.swapped = int(dims[6]), bool(dims[7])
        self.slots = 1
        if int(slots) != 1:
            self.set_slots(slots)

    def suggested_slots(self, most=8):
        """How many restarts to advance together as slots of one launch.  Slots share the index stream and
        turn a gathered row into whole cache lines, which pays while the slot-interleaved gathered tables
        (rows x K x 8 bytes x slots) stay below roughly 700 MB; beyond that the gathers get slower than the
        sharing saves.  Measured per restart-iteration, 1 / 4 / 8 slots: 1M ratings K=20 98 / 77 / 75 us,
        10M x 1M users K=20 1,046 / 838 / 919, 4M x 400k K=50 905 / 878 / 886, 10M x 1M K=50 (BASELINE
        config 5) 2,213 / 2,469 / 2,394 -- there the restarts run one after the other."""
        rows = max(self.n_items if self.swapped else self.n_users, self.n_pairs)
        groups = self.l if self.swapped else self.k
        table = rows * 8 * (-(-groups // 4) * 4)
        s = max(1, int(most))
        while s > 1 and s * table > 700 * 2 ** 20:
            s //= 2
        return s

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.call("mmsbm_hip_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- restart slots -------------------------------------------------------------------
    def set_slots(self, n_slots):
        """Hold ``n_slots`` independent restarts (parameter sets) over the same triples;
        ``iterate`` advances all of them with one set of launches, everything else acts on
        the selected slot.  Drops all parameters and selects slot 0.  If the device runs out of
        memory the library falls back to ONE slot and this raises; ``slots`` always reports
        what the context really holds."""
        try:
            _lib.call("mmsbm_hip_set_slots", self._h, int(n_slots))
        finally:
            n = C.c_int(1)
            _lib.call("mmsbm_hip_slots", self._h, C.byref(n), None, None)
            self.slots = int(n.value)

    def select(self, slot):
        _lib.call("mmsbm_hip_select_slot", self._h, int(slot))
        return self

    @property
    def selected(self):
        s = C.c_int(0)
        _lib.call("mmsbm_hip_slots", self._h, None, C.byref(s), None)
        return int(s.value)

    @property
    def bytes_per_slot(self):
        """Device memory one restart slot occupies (parameters, A/C/T tables, slabs)."""
        b = C.c_int64(0)
        _lib.call("mmsbm_hip_slots", self._h, None, None, C.byref(b))
        return int(b.value)

    def max_slots(self, fraction=0.5, sharers=1):
        """How many slots fit in `fraction` of the memory that is FREE on the device right now
        (hipMemGetInfo: other contexts and processes already count) plus what this context's own
        slots hold, divided among `sharers` workers that size their batches at the same time
        (contexts_per_device, cv_fit lanes that repeat a GPU).  At least 1."""
        free = C.c_int64(0)
        _lib.call("mmsbm_hip_device_mem", self.device, C.byref(free), None)
        per = max(self.bytes_per_slot, 1)
        mine = per * self.slots  # re-used by the next set_slots
        return max(1, int(fraction * (free.value / max(int(sharers), 1) + mine)) // per)

    # -- parameters ----------------------------------------------------------------------
    def _shapes(self):
        return ((self.n_users, self.k), (self.n_items, self.l), (self.k, self.l, self.n_ratings))

    def set_params(self, theta, eta, pr):
        theta, eta, pr = _f64(theta), _f64(eta), _f64(pr)
        for arr, shp, nm in zip((theta, eta, pr), self._shapes(), ("theta", "eta", "pr")):
            if arr.shape != shp:
                raise ValueError(f"{nm} has shape {arr.shape}, expected {shp}")
        _lib.call("mmsbm_hip_set_params", self._h, _p(theta, C.c_double), _p(eta, C.c_double),
                  _p(pr, C.c_double))

    def init_params(self, seed):
        """The reference's random start (src/mmsbm.py:224-233) generated on the device:
        theta0 and eta0 are drawn there from ``default_rng(seed)``'s PCG64 stream (bit-identical
        to numpy), only the small p0 is drawn on the host.  Returns p0."""
        bg = np.random.PCG64(seed)
        words = pcg64_words(bg)
        bg.advance(self.n_users * self.k + self.n_items * self.l)
        pr = normalize_with_self(np.random.Generator(bg).random((self.k, self.l, self.n_ratings)))
        _lib.call("mmsbm_hip_init_params", self._h, words, _p(pr, C.c_double))
        return pr

    def get_params(self):
        theta, eta, pr = (np.empty(s, dtype=np.float64) for s in self._shapes())
        _lib.call("mmsbm_hip_get_params", self._h, _p(theta, C.c_double), _p(eta, C.c_double),
                  _p(pr, C.c_double))
        return theta, eta, pr

    def result(self):
        """(likelihood, theta, eta, pr) of the selected slot: what likelihood() and get_params() return, the
        download overlapped with the likelihood kernels."""
        theta, eta, pr = (np.empty(s, dtype=np.float64) for s in self._shapes())
        out = C.c_double(0.0)
        _lib.call("mmsbm_hip_result", self._h, _p(theta, C.c_double), _p(eta, C.c_double), _p(pr, C.c_double),
                  C.byref(out))
        return np.float64(out.value), theta, eta, pr

    def degrees(self):
        d_u = np.empty(self.n_users, dtype=np.int64)
        d_i = np.empty(self.n_items, dtype=np.int64)
        _lib.call("mmsbm_hip_degrees", self._h, _p(d_u, C.c_int64), _p(d_i, C.c_int64))
        return d_u, d_i

    # -- the hot path ----------------------------------------------------------------------
    def iterate(self, n_iters, sync=True):
        _lib.call("mmsbm_hip_em_iterate", self._h, int(n_iters))
        if sync:
            self.synchronize()

    def synchronize(self):
        _lib.call("mmsbm_hip_synchronize", self._h)

    def update_coefficients(self):
        n_theta, n_eta, n_pr = (np.empty(s, dtype=np.float64) for s in self._shapes())
        _lib.call("mmsbm_hip_update_coefficients", self._h, _p(n_theta, C.c_double),
                  _p(n_eta, C.c_double), _p(n_pr, C.c_double))
        return n_theta, n_eta, n_pr

    def likelihood(self):
        out = C.c_double(0.0)
        _lib.call("mmsbm_hip_likelihood", self._h, C.byref(out))
        return np.float64(out.value)

    def compute_omegas(self):
        out = np.empty((self.n_obs, self.k, self.l), dtype=np.float64)
        _lib.call("mmsbm_hip_compute_omegas", self._h, _p(out, C.c_double), out.size)
        return out

    def prod_dist(self, pairs):
        d = np.asarray(pairs)
        if d.ndim != 2 or d.shape[1] < 2:
            raise ValueError("pairs must have shape (M, >=2): [user_idx, item_idx, ...]")
        u, i = _i32(d[:, 0]), _i32(d[:, 1])
        out = np.empty((len(u), self.n_ratings), dtype=np.float64)
        _lib.call("mmsbm_hip_prod_dist", self._h, len(u), _p(u, C.c_int32), _p(i, C.c_int32),
                  _p(out, C.c_double))
        return out

    # -- predict / score on the device (src/mmsbm.py:297-315, 488-539) ----------------------
    STAT_NAMES = ("rows", "true", "almost", "s2", "true_pond", "s2pond")

    @staticmethod
    def final_stats(raw):
        """The reference's five scores (src/mmsbm.py:530-539) from the six device sums."""
        n = raw[0]
        with np.errstate(divide="ignore", invalid="ignore"):
            return {"accuracy": np.float64(raw[1]) / n, "one_off_accuracy": np.float64(raw[2]) / n,
                    "mae": 1 - np.float64(raw[4]) / n, "s2": np.int64(raw[3]), "s2pond": np.float64(raw[5])}

    def predict_begin(self, test, rating_weights):
        """Open a scoring session over (M,3) test triples [user, item, true rating index]."""
        u, i, r = split_triples(test)
        w = _f64(rating_weights)
        if w.shape != (self.n_ratings,):
            raise ValueError(f"rating_weights has shape {w.shape}, expected ({self.n_ratings},)")
        self._ps_rows = len(u)
        _lib.call("mmsbm_hip_predict_begin", self._h, len(u), _p(u, C.c_int32), _p(i, C.c_int32),
                  _p(r, C.c_int32), _p(w, C.c_double))

    def predict_add(self):
        """Add the selected slot's rating distribution to the session; its six raw sums."""
        st = np.zeros(6, dtype=np.float64)
        _lib.call("mmsbm_hip_predict_add", self._h, _p(st, C.c_double))
        return st

    def predict_finish(self, want_matrix=True):
        """(mean distribution (M,R) or None, six raw sums of the mean); closes the session."""
        st = np.zeros(6, dtype=np.float64)
        out = np.empty((self._ps_rows, self.n_ratings), dtype=np.float64) if want_matrix else None
        _lib.call("mmsbm_hip_predict_finish", self._h,
                  _p(out, C.c_double) if want_matrix else None, _p(st, C.c_double))
        return out, st

    # -- measurement -------------------------------------------------------------------------
    def time_iterations(self, n_iters):
        """Device milliseconds for n_iters EM iterations (HIP events on the context stream)."""
        ms = C.c_float(0.0)
        _lib.call("mmsbm_hip_time_iterations", self._h, int(n_iters), C.byref(ms))
        return float(ms.value)

    def profile_iterations(self, n_iters):
        """{kernel name: (mean microseconds per launch, launches per iteration, bytes read,
        bytes written)} from a HIP event pair around every launch."""
        lib = _lib.load()
        nk = lib.mmsbm_hip_kernel_count()
        us = (C.c_float * nk)()
        cnt = (C.c_int * nk)()
        _lib.call("mmsbm_hip_profile_iterations", self._h, int(n_iters), us, cnt)
        out = {}
        for j in range(nk):
            rd, wr = C.c_int64(0), C.c_int64(0)
            _lib.call("mmsbm_hip_kernel_bytes", self._h, j, C.byref(rd), C.byref(wr))
            out[lib.mmsbm_hip_kernel_name(j).decode()] = (float(us[j]), int(cnt[j]),
                                                         int(rd.value), int(wr.value))
        return out

    def time_stage(self, stage, reps=50):
        """Mean microseconds of `reps` back-to-back launches of one stage (clobbers state)."""
        us = C.c_float(0.0)
        _lib.call("mmsbm_hip_time_stage", self._h, int(stage), int(reps), C.byref(us))
        return float(us.value)

    def set_option(self, name, value):
        """Named tuning knob of the library (currently "graph")."""
        _lib.call("mmsbm_hip_set_option", self._h, name.encode(), float(value))

    def get_option(self, name):
        v = C.c_double(0.0)
        _lib.call("mmsbm_hip_get_option", self._h, name.encode(), C.byref(v))
        return float(v.value)

    def set_graph_mode(self, mode):
        """0 eager launches (default), 1 replay a captured hipGraph of two iterations."""
        _lib.call("mmsbm_hip_set_graph_mode", self._h, int(mode))


def build_layout(data, n_users, n_items, n_ratings, target_chunks=1024, fused_caps=None):
    """Host-only: the sorted CSR layout the library uploads (dict of int32 arrays).  ``fused_caps = (pair cap, user
    cap)`` adds the whole-segment lists of the two-launch iteration (mmsbm_hip_layout_fused): ``fused_pairs`` /
    ``fused_users`` = {"units", "items", "splits", "chunks", "max_parts", "built"}."""
    u, i, r = split_triples(data)
    h = C.c_void_p()
    _lib.call("mmsbm_hip_layout_build", len(u), int(n_users), int(n_items), int(n_ratings),
              _p(u, C.c_int32), _p(i, C.c_int32), _p(r, C.c_int32), int(target_chunks), C.byref(h))
    names = ["pair_off", "pair_user", "pair_item", "rating_off", "user_off", "user_pair",
             "item_off", "item_pairs", "item_deg", "chunk_off", "chunks", "mv_chunks",
             "pair_items", "user_items", "pair_splits", "user_splits"]
    out = {}
    try:
        for which, nm in enumerate(names):
            cnt = C.c_int64(0)
            _lib.call("mmsbm_hip_layout_array", h, which, None, 0, C.byref(cnt))
            arr = np.empty(cnt.value, dtype=np.int32)
            if cnt.value:
                _lib.call("mmsbm_hip_layout_array", h, which, _p(arr, C.c_int32), arr.size,
                          C.byref(cnt))
            out[nm] = arr.reshape(-1, 4) if which >= 10 else arr
        if fused_caps is not None:
            for side, (key, cap) in enumerate(zip(("fused_pairs", "fused_users"), fused_caps)):
                rec = {}
                for which, nm in enumerate(("units", "items", "splits", "chunks", "info")):
                    cnt = C.c_int64(0)
                    _lib.call("mmsbm_hip_layout_fused", h, side, int(cap), which, None, 0, C.byref(cnt))
                    arr = np.empty(cnt.value, dtype=np.int32)
                    if cnt.value:
                        _lib.call("mmsbm_hip_layout_fused", h, side, int(cap), which, _p(arr, C.c_int32), arr.size, C.byref(cnt))
                    rec[nm] = arr.reshape(-1, 4) if which < 4 else arr
                rec["max_parts"], rec["built"] = int(rec["info"][0]), bool(rec["info"][1])
                out[key] = rec
    finally:
        _lib.call("mmsbm_hip_layout_free", h)
    return out

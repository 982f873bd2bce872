"""Value -> dense id encoding with the reference's ordering (src/data_handler.py).

Every cell is turned into ``str(value)`` (data_handler.py:27-32) and the distinct strings
of a column are numbered in lexicographic order (data_handler.py:39-44; '10' < '2').  Test
rows holding a user, item or rating unseen in training are dropped with a warning
(data_handler.py:109-129).  Vectorised with numpy instead of per-cell dict lookups.
"""
from __future__ import annotations

import logging

import numpy as np


def _columns(data):
    """Three python-object columns from a DataFrame, an (N,3) array or a 3-tuple."""
    if hasattr(data, "iloc"):  # pandas
        return [data.iloc[:, j].tolist() for j in range(3)], list(data.columns[:3])
    if isinstance(data, (tuple, list)) and len(data) == 3 and not np.isscalar(data[0]):
        return [list(np.asarray(c).tolist()) for c in data], ["users", "items", "ratings"]
    arr = np.asarray(data, dtype=object)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError("expected three columns: users, items, ratings")
    return [arr[:, j].tolist() for j in range(3)], ["users", "items", "ratings"]


def _as_str(col):
    for v in col:
        if v is None or (isinstance(v, float) and v != v):
            raise AssertionError("Data contains missing values. Aborting.")  # data_handler.py:24
    return np.array([str(v) for v in col], dtype=object).astype(str) if len(col) else np.array([], dtype=str)


class Encoder:
    """fit on the training frame, then encode test frames against the same dictionaries."""

    def __init__(self):
        self.labels = None  # three sorted arrays of str: id -> original label

    def fit_transform(self, data):
        cols, _ = _columns(data)
        self.labels, out = [], []
        for col in cols:
            s = _as_str(col)
            uniq, inv = np.unique(s, return_inverse=True) if len(s) else (s, np.array([], dtype=np.int64))
            self.labels.append(uniq)
            out.append(inv.astype(np.int64))
        return np.stack(out, axis=1) if len(out[0]) else np.zeros((0, 3), dtype=np.int64)

    def transform(self, data, logger=None):
        if self.labels is None:
            raise AssertionError("encoder has not seen training data")
        cols, names = _columns(data)
        strs = [_as_str(c) for c in cols]
        keep = np.ones(len(strs[0]), dtype=bool)
        ids = []
        log = logger or logging.getLogger("MMSBM")
        for name, s, lab in zip(("users", "items", "ratings"), strs, self.labels):
            pos = np.searchsorted(lab, s)
            pos = np.minimum(pos, max(len(lab) - 1, 0))
            hit = lab[pos] == s if len(lab) else np.zeros(len(s), dtype=bool)
            # the reference filters column after column; a row only counts as "unseen" for a
            # column if it survived the previous ones (data_handler.py:119-127)
            missing = sorted(set(s[keep & ~hit].tolist()))
            if missing:
                log.warning(f"The {name} {', '.join(missing)} are in the test set but weren't in "
                            f"the train set so I'll remove them.")
            keep &= hit
            ids.append(pos.astype(np.int64))
        return np.stack(ids, axis=1)[keep]

    # decoding helpers (data_handler.py:74-99)
    def user_labels(self):
        return self.labels[0].tolist()

    def item_labels(self):
        return self.labels[1].tolist()

    def rating_labels(self):
        return self.labels[2].tolist()

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
    """Three columns (typed arrays where possible) from a DataFrame, an (N,3) array or a 3-tuple."""
    if hasattr(data, "iloc"):  # pandas
        return [data.iloc[:, j].to_numpy() for j in range(3)], list(data.columns[:3])
    if isinstance(data, (tuple, list)) and len(data) == 3 and not np.isscalar(data[0]):
        return [np.asarray(c) for c in data], ["users", "items", "ratings"]
    arr = np.asarray(data)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError("expected three columns: users, items, ratings")
    return [arr[:, j] for j in range(3)], ["users", "items", "ratings"]


def _as_str(col):
    for v in col:
        if v is None or (isinstance(v, float) and v != v):
            raise AssertionError("Data contains missing values. Aborting.")  # data_handler.py:24
    return np.array([str(v) for v in col], dtype=object).astype(str) if len(col) else np.array([], dtype=str)


_POW10 = 10 ** np.arange(1, 20, dtype=np.uint64)   # 10 .. 10^19


def _decimal_string_order(values):
    """Permutation that puts non-negative integers in the lexicographic order of their decimal strings
    ('10' < '2', '1' < '10') without making a string: compare the digits left-aligned, i.e. the value
    scaled to the longest length, the shorter string first among equals."""
    v = values.astype(np.uint64)
    digits = np.searchsorted(_POW10, v, side="right") + 1
    scale = np.concatenate([np.ones(1, dtype=np.uint64), _POW10])[int(digits.max()) - digits] if len(v) else v
    return np.lexsort((digits, v * scale))


def _factorize_ints(col, out=None, keep_map=None):
    """_factorize_as_str for a column of non-negative integers (what ids usually are), all in typed numpy
    passes: a presence table instead of a hash when the ids are reasonably dense, the string order computed
    numerically, strings made once per distinct value by numpy.  None if the column does not qualify."""
    lo, hi = int(col.min()), int(col.max())
    if lo < 0 or hi >= 2 ** 62:  # (negative numbers sort as strings with their sign; huge ones would overflow int64 below)
        return None
    span = hi - lo + 1
    codes = None
    if span <= max(8 * len(col), 1 << 22):
        shifted = col.astype(np.int64) - lo if lo else col.astype(np.int64, copy=False)
        present = np.zeros(span, dtype=bool)
        present[shifted] = True
        uniq = np.flatnonzero(present) + lo
    else:  # sparse ids: hash-factorise
        import pandas as pd
        codes, uniq = pd.factorize(col)
        uniq = np.asarray(uniq)
    order = _decimal_string_order(uniq)
    rank = np.empty(len(uniq), dtype=np.int32)
    rank[order] = np.arange(len(uniq), dtype=np.int32)
    if codes is None:
        lookup = np.zeros(span, dtype=np.int32)
        lookup[uniq - lo] = rank
        ids = np.take(lookup, shifted, out=out)
    else:
        ids = np.take(rank, codes, out=out)
    if keep_map is not None:  # value -> id without strings, for Encoder.transform
        keep_map["values"], keep_map["ids"], keep_map["lo"] = uniq, rank, lo
        if codes is None:  # dense ids: a table answers a test column in one gather
            table = np.full(span, -1, dtype=np.int32)
            table[uniq - lo] = rank
            keep_map["table"] = table
    return (out if out is not None else ids), uniq[order].astype(str)


def _factorize_as_str(col, out=None, keep_map=None):
    """(ids, labels): labels = sorted distinct str(value); ids[n] = rank of str(col[n]).

    Hash-factorises the raw values first (pandas, O(N)) and stringifies only the distinct
    ones, so a million-row column costs milliseconds instead of a Python call per cell.
    ``out``: optional int32 array the ids are written into (no temporaries)."""
    import pandas as pd

    if len(col) == 0:
        return np.zeros(0, dtype=np.int32), np.array([], dtype=str)
    col = np.asarray(col)
    if col.dtype.kind in "iu":
        fast = _factorize_ints(col, out, keep_map)
        if fast is not None:
            return fast
    if col.dtype.kind in "US":  # fixed-width numpy strings hash slowly: go through object
        col = col.astype(object)
    codes, uniq = pd.factorize(col, use_na_sentinel=True)
    if (codes < 0).any():
        raise AssertionError("Data contains missing values. Aborting.")  # data_handler.py:24
    as_str = np.array([str(v) for v in uniq.tolist()], dtype=object).astype(str)
    labels, inv = np.unique(as_str, return_inverse=True)  # distinct raw values may share a string
    inv = inv.astype(np.int32)
    if out is None:
        return inv[codes], labels
    np.take(inv, codes, out=out)
    return out, labels


class Encoder:
    """fit on the training frame, then encode test frames against the same dictionaries."""

    def __init__(self):
        self.labels = None  # three sorted arrays of str: id -> original label
        self._int_maps = [None, None, None]  # integer training columns: ascending values and their ids

    def fit_transform(self, data):
        """(N,3) int32 ids [user, item, rating], column-major so that each column is contiguous
        (what the device library takes)."""
        cols, _ = _columns(data)
        out = np.empty((len(cols[0]), 3), dtype=np.int32, order="F")

        maps = [{}, {}, {}]

        def one(j):
            return _factorize_as_str(cols[j], out=out[:, j] if len(cols[j]) else None, keep_map=maps[j])[1]

        if len(cols[0]) >= 1_000_000:  # the hash passes release the GIL: one thread per column
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(3) as pool:
                self.labels = list(pool.map(one, range(3)))
        else:
            self.labels = [one(j) for j in range(3)]
        self._int_maps = [m if m else None for m in maps]
        return out

    def transform(self, data, logger=None):
        if self.labels is None:
            raise AssertionError("encoder has not seen training data")
        cols, names = _columns(data)
        keep = np.ones(len(cols[0]), dtype=bool)
        ids = []
        log = logger or logging.getLogger("MMSBM")
        for j, (name, col, lab) in enumerate(zip(("users", "items", "ratings"), cols, self.labels)):
            col = np.asarray(col)
            imap = self._int_maps[j]
            if imap is not None and col.dtype.kind in "iu" and len(col) and int(col.min()) >= 0 and int(col.max()) < 2 ** 62:
                # integers against an integer training column: binary search on the values, no strings
                c64 = col.astype(np.int64, copy=False)
                if "table" in imap:
                    rel = c64 - imap["lo"]
                    inside = (rel >= 0) & (rel < len(imap["table"]))
                    found = np.where(inside, imap["table"][np.where(inside, rel, 0)], -1)
                else:  # sparse training ids: hash lookup
                    import pandas as pd
                    at = pd.Index(imap["values"]).get_indexer(c64)
                    found = np.where(at >= 0, imap["ids"][np.maximum(at, 0)], -1)
                hit = found >= 0
                unseen = np.unique(c64[keep & ~hit])
                if len(unseen):
                    log.warning(f"The {name} {', '.join(sorted(str(v) for v in unseen.tolist()))} are in the test set "
                                f"but weren't in the train set so I'll remove them.")
                keep &= hit
                ids.append(found.astype(np.int32, copy=False))
                continue
            codes, test_labels = _factorize_as_str(col)          # this frame's own dictionary
            pos_l = np.searchsorted(lab, test_labels)
            pos_l = np.minimum(pos_l, max(len(lab) - 1, 0))
            hit_l = lab[pos_l] == test_labels if len(lab) else np.zeros(len(test_labels), dtype=bool)
            pos, hit = pos_l[codes], hit_l[codes]
            # the reference filters column after column; a row only counts as "unseen" for a
            # column if it survived the previous ones (data_handler.py:119-127)
            missing = sorted(set(test_labels[np.unique(codes[keep & ~hit])].tolist()))
            if missing:
                log.warning(f"The {name} {', '.join(missing)} are in the test set but weren't in "
                            f"the train set so I'll remove them.")
            keep &= hit
            ids.append(pos.astype(np.int32))
        out = np.empty((int(keep.sum()), 3), dtype=np.int32, order="F")
        for j in range(3):
            out[:, j] = ids[j][keep]
        return out

    # decoding helpers (data_handler.py:74-99)
    def user_labels(self):
        return self.labels[0].tolist()

    def item_labels(self):
        return self.labels[1].tolist()

    def rating_labels(self):
        return self.labels[2].tolist()

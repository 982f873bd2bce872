"""Host-side layout builder (mmsbm_amd/csrc/layout.hpp through the C ABI) -- no GPU needed."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from mmsbm_amd import _lib
from mmsbm_amd.core import build_layout
from oracle import mmsbm_oracle as orc

import emulate


def random_triples(rng, n, u, i, r, dense=True):
    d = np.stack([rng.integers(0, u, n), rng.integers(0, i, n), rng.integers(0, r, n)], axis=1)
    return d.astype(np.int64)


def check_invariants(data, n_users, n_items, n_ratings, lay):
    n = len(data)
    u, i, r = data[:, 0], data[:, 1], data[:, 2]
    # distinct (rating, item) pairs, rating-major
    keys = np.unique(r * n_items + i)
    assert len(lay["pair_item"]) == len(keys)
    assert np.array_equal(lay["pair_item"], keys % n_items)
    assert np.array_equal(np.diff(lay["rating_off"]), np.bincount(keys // n_items, minlength=n_ratings))
    assert lay["pair_off"][0] == 0 and lay["pair_off"][-1] == n
    # triples per pair, users ascending inside a pair
    order = np.lexsort((u, i, r))
    assert np.array_equal(lay["pair_user"], u[order])
    cnt = np.unique(r * n_items + i, return_counts=True)[1]
    assert np.array_equal(np.diff(lay["pair_off"]), cnt)
    # user order
    assert np.array_equal(np.diff(lay["user_off"]), np.bincount(u, minlength=n_users))
    pair_id = np.searchsorted(keys, r * n_items + i)
    for uu in range(min(n_users, 50)):
        got = lay["user_pair"][lay["user_off"][uu]:lay["user_off"][uu + 1]]
        assert np.array_equal(got, np.sort(pair_id[u == uu]))
    # pairs of each item, ascending rating
    assert np.array_equal(lay["item_deg"], np.bincount(i, minlength=n_items))
    for ii in range(min(n_items, 50)):
        got = lay["item_pairs"][lay["item_off"][ii]:lay["item_off"][ii + 1]]
        assert np.array_equal(got, np.nonzero(lay["pair_item"] == ii)[0])
    # chunks tile every rating's pair range exactly once and never straddle ratings
    ch = lay["chunks"]
    covered = np.zeros(len(keys), dtype=int)
    for rr, qb, qe, _ in ch:
        assert lay["rating_off"][rr] <= qb < qe <= lay["rating_off"][rr + 1]
        covered[qb:qe] += 1
    assert np.all(covered == 1)
    assert np.array_equal(np.diff(lay["chunk_off"]), np.bincount(ch[:, 0], minlength=n_ratings)
                          if len(ch) else np.zeros(n_ratings, dtype=int))


@pytest.mark.parametrize("n,u,i,r,chunks", [(1, 1, 1, 1, 4), (100, 5, 10, 5, 4), (5000, 300, 40, 7, 16),
                                            (20000, 50, 2000, 3, 1024), (3000, 3000, 2, 2, 8)])
def test_layout_invariants(n, u, i, r, chunks):
    rng = np.random.default_rng(n + u)
    data = random_triples(rng, n, u, i, r)
    lay = build_layout(data, u, i, r, target_chunks=chunks)
    check_invariants(data, u, i, r, lay)


def test_layout_with_duplicates_and_absent_ids():
    # user 3 and item 1 never occur, rating 1 never occurs, rows duplicated
    data = np.array([[0, 0, 0], [0, 0, 0], [2, 2, 2], [4, 0, 2], [0, 0, 2], [4, 2, 0]], dtype=np.int64)
    lay = build_layout(data, 5, 3, 3, target_chunks=2)
    check_invariants(data, 5, 3, 3, lay)
    assert lay["user_off"][4] - lay["user_off"][3] == 0
    assert lay["rating_off"][2] == lay["rating_off"][1]


def test_layout_empty():
    lay = build_layout(np.zeros((0, 3), dtype=np.int64), 3, 2, 2)
    assert len(lay["pair_item"]) == 0 and lay["user_off"].tolist() == [0, 0, 0, 0]
    assert len(lay["chunks"]) == 0


def test_layout_rejects_out_of_range_ids():
    bad = np.array([[0, 0, 0], [5, 0, 0]], dtype=np.int64)
    with pytest.raises(_lib.HipLibraryError) as exc:
        build_layout(bad, 5, 1, 1)
    assert exc.value.code == _lib.E_INVALID and "out of range" in exc.value.message


@pytest.mark.parametrize("fixture,names", [
    ("g4_2k_k10", ("theta_0", "eta_0", "pr_0", "n_theta_1", "n_eta_1", "n_pr_1")),
    ("g1_c1_mock", ("c1_theta_0", "c1_eta_0", "c1_pr_0", "c1_n_theta_1", "c1_n_eta_1", "c1_n_pr_1")),
])
def test_factorised_form_matches_reference_numerators(fixture, names):
    """The re-associated sums the kernels use == the reference's dense ones (golden vectors)."""
    g = load_golden(fixture)
    train = g["train"]
    theta, eta, pr = g[names[0]], g[names[1]], g[names[2]]
    lay = build_layout(train, theta.shape[0], eta.shape[0], pr.shape[2], target_chunks=7)
    got = emulate.iteration(lay, theta, eta, pr)
    for a, nm in zip(got, names[3:]):
        assert rel_err(a, g[nm]) < 1e-13, nm


def test_factorised_form_edge_cases():
    g = load_golden("edge_cases")
    for tag in ("zero", "dup", "tiny", "mix"):
        data, theta, eta, pr = (g[f"{tag}_{x}"] for x in ("data", "theta", "eta", "pr"))
        lay = build_layout(data, theta.shape[0], eta.shape[0], pr.shape[2], target_chunks=3)
        got = emulate.iteration(lay, theta, eta, pr)
        for a, nm in zip(got, ("n_theta", "n_eta", "n_pr")):
            want = g[f"{tag}_{nm}"]
            # 'tiny': the reference's (theta*eta)*p underflows to exactly 0 where the factorised
            # theta*(p.eta) keeps a denormal ~1e-315 -- an absolute floor far below any tolerance
            assert np.allclose(a, want, rtol=1e-12, atol=1e-300), (tag, nm)


def item_length(n_obs, nseg):
    """The cut policy of layout.hpp (item_length), restated."""
    mean = max(n_obs // nseg, 1)
    if nseg >= 65536:
        return min(max(64, 4 * mean), 1 << 20)
    if mean <= 16:
        return 64
    want, length = max(n_obs // 65536, 16), 16
    while length * 2 <= want:
        length *= 2
    return length


def check_worklist(lay, which, off, cut):
    items, splits = lay[f"{which}_items"], lay[f"{which}_splits"]
    lens = np.diff(off)
    if lens.max() <= cut:
        assert len(items) == 0 and len(splits) == 0
        return 0
    assert np.all(items[:, 2] - items[:, 1] <= cut) and np.all(items[:, 2] >= items[:, 1])
    cover = np.zeros(off[-1], dtype=int)                  # the items tile every segment exactly
    np.add.at(cover, np.concatenate([np.arange(b, e) for _, b, e, _ in items]) if len(items) else [], 1)
    assert np.all(cover == 1)
    for seg, b, e, part in items[:: max(1, len(items) // 500)]:
        assert off[seg] <= b and e <= off[seg + 1] and (part >= 0) == (lens[seg] > cut)
    assert sorted(splits[:, 0].tolist()) == np.nonzero(lens > cut)[0].tolist()
    pieces = splits[:, 2]
    assert np.array_equal(pieces, -(-lens[splits[:, 0]] // cut))
    small = pieces <= 32                                   # few pieces first, then the rest
    n_small = int(small.sum())
    assert small[:n_small].all() and not small[n_small:].any()
    for seg, first, cnt, _ in splits[:: max(1, len(splits) // 200)]:
        mine = items[items[:, 0] == seg]
        assert mine[:, 3].tolist() == list(range(first, first + cnt))
    return len(splits)


def test_long_segments_become_work_items():
    """Heavy users / popular pairs are cut into pieces, combined in piece order."""
    rng = np.random.default_rng(9)
    n = 5000
    u = np.where(rng.random(n) < 0.4, 3, rng.integers(0, 300, n))      # user 3 holds ~40 % of the rows
    i = np.where(rng.random(n) < 0.3, 1, rng.integers(0, 40, n))
    data = np.stack([u, i, rng.integers(0, 3, n)], axis=1).astype(np.int64)
    lay = build_layout(data, 300, 40, 3)
    assert check_worklist(lay, "user", lay["user_off"], item_length(n, 300)) > 0
    assert check_worklist(lay, "pair", lay["pair_off"], item_length(n, len(lay["pair_item"]))) > 0
    # uniform sparse data: no items at all (segments are used as they are)
    flat = build_layout(random_triples(rng, 3000, 400, 50, 4), 400, 50, 4)
    assert len(flat["user_items"]) == 0 and len(flat["pair_splits"]) == 0


def test_cut_policy_for_sparse_dense_and_skewed_data():
    """Sparse data with many segments (BASELINE's configs): nothing is cut.  Dense data (few users
    with hundreds of ratings each): cut towards 65,536 work items, never below 16 triples.  Many
    segments plus a heavy one: only the outlier is cut."""
    assert item_length(1_000_000, 100_000) == 64 and item_length(10_000_000, 1_000_000) == 64
    assert item_length(100_000, 10_000) == 64                       # C2: short segments, leave them
    assert item_length(1_000_000, 6040) == 16 and item_length(20_000_000, 27_000) == 256
    assert item_length(20_000_000, 138_000) == 4 * 144
    rng = np.random.default_rng(4)
    dense = random_triples(rng, 60_000, 400, 300, 5)                # 150 ratings per user
    lay = build_layout(dense, 400, 300, 5)
    cut = item_length(60_000, 400)
    assert cut == 16
    n_split = check_worklist(lay, "user", lay["user_off"], cut)
    assert n_split >= 390 and len(lay["user_items"]) >= 60_000 // 16
    check_worklist(lay, "pair", lay["pair_off"], item_length(60_000, len(lay["pair_item"])))
    many = random_triples(rng, 700_000, 70_000, 5_000, 5)           # >= 65,536 users, 10 ratings each
    many[:40_000, 0] = 7                                            # ... and one with 40,000
    lay = build_layout(many, 70_000, 5_000, 5)
    cut = item_length(700_000, 70_000)
    assert cut == 64
    assert check_worklist(lay, "user", lay["user_off"], cut) == 1
    assert lay["user_splits"][0, 2] == -(-np.diff(lay["user_off"])[7] // 64) > 32


def _check_fused_lists(off, work_items, work_splits, fl, cap, whole_workgroups=None):
    """Every workgroup holds WHOLE segments; its items are the segments' pieces in (segment, piece) order and cover
    every triple once; split segments carry workgroup-local partial rows 0 .. max_parts - 1."""
    units, items, splits = fl["units"], fl["items"], fl["splits"]
    nseg = len(off) - 1
    n_pieces = {int(s): int(n) for s, _, n, _ in work_splits}
    seen_seg, cover = np.zeros(nseg, dtype=int), np.zeros(int(off[-1]), dtype=int)
    assert units[0, 0] == 0 and units[-1, 1] == len(items) and units[0, 2] == 0 and units[-1, 3] == len(splits)
    assert np.array_equal(units[1:, 0], units[:-1, 1]) and np.array_equal(units[1:, 2], units[:-1, 3])
    most = 0
    for ib, ie, sb, se in units:
        its, sps = items[ib:ie], splits[sb:se]
        segs = its[:, 0]
        assert np.all(np.diff(segs) >= 0)                                  # segment order
        for s in np.unique(segs):
            mine = its[segs == s]
            seen_seg[s] += 1                                               # ... and a segment is in ONE workgroup
            assert mine[0, 1] == off[s] and mine[-1, 2] == off[s + 1] and np.array_equal(mine[1:, 1], mine[:-1, 2])
            if len(mine) == 1 and mine[0, 3] < 0:
                assert s not in n_pieces
            else:
                assert n_pieces[s] == len(mine) and np.array_equal(mine[:, 3], mine[0, 3] + np.arange(len(mine)))
        for b, e in its[:, 1:3]:
            cover[b:e] += 1
        parts = 0
        for s, first, n, big in sps:
            assert first == parts and n == n_pieces[s] and big == (n > 32)
            parts += n
        assert parts == int((its[:, 3] >= 0).sum())
        most = max(most, parts)
        if len(np.unique(segs)) > 1:
            assert len(its) <= cap
    assert np.all(seen_seg == 1) and np.all(cover == 1) and most == fl["max_parts"]


def test_whole_segment_lists_of_the_two_launch_iteration():
    """layout.hpp: FusedLists -- data with uneven degrees (the MovieLens-100k shape: few users with many ratings each,
    popular items): every cut segment's pieces in ONE workgroup, pair side (64-pair units rebuilt with at most 64 work
    items) and user side (workgroups of at most 64 / 32 items, a longer segment alone)."""
    rng = np.random.default_rng(0)
    n, u, i = 40_000, 400, 700
    pu, pi = rng.lognormal(0, 0.8, u), rng.lognormal(0, 1.4, i)
    data = np.stack([rng.choice(u, n, p=pu / pu.sum()), rng.choice(i, n, p=pi / pi.sum()), rng.integers(0, 5, n)], axis=1).astype(np.int64)
    data[:200, 0] = 7                                                       # (a user with more pieces than a workgroup has groups)
    for cap_u in (64, 32):
        lay = build_layout(data, u, i, 5, fused_caps=(64, cap_u))
        assert len(lay["pair_splits"]) > 0 and len(lay["user_splits"]) > 0
        fp, fu = lay["fused_pairs"], lay["fused_users"]
        assert fp["built"] and fu["built"]
        _check_fused_lists(lay["pair_off"], lay["pair_items"], lay["pair_splits"], fp, 64)
        _check_fused_lists(lay["user_off"], lay["user_items"], lay["user_splits"], fu, cap_u)
        # the rebuilt units: rating-homogeneous, at most 64 pairs and 64 items, tiling every rating's pairs once,
        # padded to a multiple of 8 per rating like the plain list; unit j of the lists is chunk j
        ch = fp["chunks"]
        assert len(ch) == len(fp["units"])
        covered = np.zeros(len(lay["pair_item"]), dtype=int)
        for (rr, qb, qe, _), (ib, ie, _, _) in zip(ch, fp["units"]):
            assert lay["rating_off"][rr] <= qb <= qe <= lay["rating_off"][rr + 1] and qe - qb <= 64 and ie - ib <= 64
            covered[qb:qe] += 1
            segs = fp["items"][ib:ie, 0]
            assert (len(segs) == 0 and qb == qe) or (segs.min() == qb and segs.max() == qe - 1)
        assert np.all(covered == 1) and np.all(np.bincount(ch[:, 0], minlength=5) % 8 == 0)
        assert len(ch) > len(lay["mv_chunks"]) - 8                           # (a few more units than the plain 64-pair list)
    # a pair with more pieces than a unit may hold: the lists cannot be built (the data keeps the separate launches)
    heavy = data.copy()
    heavy[:6000, 1:] = (3, 2)
    lay = build_layout(heavy, u, i, 5, fused_caps=(64, 64))
    assert np.diff(lay["pair_off"]).max() >= 6000 and not lay["fused_pairs"]["built"]
    # no segment cut: one whole item per segment
    small = random_triples(rng, 3000, 300, 100, 4)
    lay = build_layout(small, 300, 100, 4, fused_caps=(64, 64))
    assert len(lay["pair_splits"]) == 0 and lay["fused_pairs"]["max_parts"] == 0 and len(lay["fused_pairs"]["splits"]) == 0
    assert len(lay["fused_users"]["items"]) == 300 and np.all(lay["fused_users"]["items"][:, 3] == -1)

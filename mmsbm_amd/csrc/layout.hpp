// layout.hpp -- host-side construction of the device data layout for the MMSBM EM core.
//
// The reference re-gathers `data[:,0..2]` on every call (src/kernels_numpy.py:26-28) and
// pre-computes degrees with O(U*N) scans (src/mmsbm.py:100-111).  Here the triples are
// sorted ONCE into two CSR-style orders:
//
//   pair order : triples sorted by (rating, item, user).  A "pair" is a distinct
//                (item, rating) combination; pairs are numbered rating-major so that any
//                contiguous run of pairs inside one rating shares one K x L tile p[:,:,r].
//   user order : triples sorted by (user, pair).
//
// plus a CSR of pairs per item and a table of rating-homogeneous pair chunks.
// Pure host code (no HIP): usable and testable without a GPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mmsbm {

struct Chunk {  // one workgroup's share of the pairs of ONE rating
  int32_t rating, q_begin, q_end, pad;
};

// A segment (the triples of a user, or of an (item, rating) pair) is walked by ONE group of lanes.
// Where that would leave too little parallelism -- few, long segments: dense data such as 6,000
// users with 165 ratings each -- or one segment would dominate (a heavy user), segments are cut
// into work items of at most `item_len` triples; the items' partial sums are combined afterwards
// in item order (deterministic).
struct WorkItem {
  int32_t seg, begin, end, part;  // part = slot of the partial row, or -1: the item IS the segment
};
struct SplitSeg {
  int32_t seg, first_part, n_parts, pad;
};
constexpr int32_t kMaxItemLen = 64;       // default cut
constexpr int64_t kTargetUnits = 65536;   // groups of lanes needed to keep the chip busy
constexpr int32_t kSmallSplitParts = 32;  // split segments with at most this many pieces are summed by one group

struct WorkList {  // empty items == no segment is longer than item_len: use the segments as they are
  std::vector<WorkItem> items;
  std::vector<SplitSeg> splits;  // those with <= kSmallSplitParts pieces first (n_small of them), then the rest
  int32_t n_parts = 0, n_small = 0, item_len = kMaxItemLen;
};

// Triples per work item for `nseg` segments holding `n_obs` triples in all.
inline int32_t item_length(int64_t n_obs, int32_t nseg) {
  if (nseg <= 0) return kMaxItemLen;
  const int64_t mean = std::max<int64_t>(n_obs / nseg, 1);
  // enough segments to fill the machine: cut only the outliers (load balance)
  if (nseg >= kTargetUnits) return int32_t(std::min<int64_t>(std::max<int64_t>(kMaxItemLen, 4 * mean), 1 << 20));
  // few segments.  Short ones: cutting adds nothing but the outliers' balance ...
  if (mean <= 16) return kMaxItemLen;
  // ... long ones: cut towards kTargetUnits work items, not below 16 triples
  int64_t want = std::max<int64_t>(n_obs / kTargetUnits, 16), len = 16;
  while (len * 2 <= want) len *= 2;
  return int32_t(std::min<int64_t>(len, 1 << 20));
}

// Segment lengths that differ a lot (log-normal / Zipf degrees: coefficient of variation > 0.5; a
// Poisson(10) has 0.32) make the groups of a wave finish far apart: such data always gets a work list,
// ordered by length.
inline bool lengths_vary(const std::vector<int32_t> &off) {
  const int64_t nseg = int64_t(off.size()) - 1;
  if (nseg < 2 || off.back() <= 0) return false;
  const double mean = double(off.back()) / double(nseg);
  double var = 0.0;
  for (int64_t s = 0; s < nseg; ++s) {
    const double d = double(off[size_t(s) + 1] - off[size_t(s)]) - mean;
    var += d * d;
  }
  return var / double(nseg) > 0.25 * mean * mean;
}

inline void sort_items_by_length(std::vector<WorkItem> &items) {
  // Items of similar length side by side, longest first: the groups of a wave then finish together
  // instead of idling until its longest segment is done, and the long pieces start early
  // (20M ratings with log-normal degrees: 1.33 -> 0.99 ms per iteration).
  // By length CLASS (powers of two), segment order kept inside a class: pieces of nearly equal length
  // stay in the order of their segments, whose fixed rows and partial rows are neighbours in memory
  // (sorting by exact length scattered them: uniform 50M x 480k, 1.29 -> 1.46 ms).
  {  // only where the lengths really differ (coefficient of variation > 0.5)
    double n = 0.0, sum = 0.0, sq = 0.0;
    for (const WorkItem &w : items) {
      if (w.seg < 0) continue;
      const double len = double(w.end - w.begin);
      n += 1.0; sum += len; sq += len * len;
    }
    if (n < 2.0 || sq / n - (sum / n) * (sum / n) <= 0.25 * (sum / n) * (sum / n)) return;
  }
  auto cls = [](const WorkItem &w) {
    int32_t len = w.end - w.begin, c = 0;
    while (len > 1) { len >>= 1; ++c; }
    return c;
  };
  std::stable_sort(items.begin(), items.end(), [&](const WorkItem &a, const WorkItem &b) { return cls(a) > cls(b); });
}

inline void build_worklist(const std::vector<int32_t> &off, WorkList &w, int32_t item_len = kMaxItemLen,
                           bool force_items = false) {
  w = WorkList();
  w.item_len = item_len;
  const int32_t nseg = int32_t(off.size()) - 1;
  bool any = force_items && nseg > 0;
  for (int32_t s = 0; s < nseg && !any; ++s) any = off[s + 1] - off[s] > item_len;
  if (!any) return;
  std::vector<SplitSeg> big;
  for (int32_t s = 0; s < nseg; ++s) {
    const int32_t len = off[s + 1] - off[s];
    if (len <= item_len) {
      w.items.push_back(WorkItem{s, off[s], off[s + 1], -1});
      continue;
    }
    const int32_t parts = (len + item_len - 1) / item_len;
    (parts <= kSmallSplitParts ? w.splits : big).push_back(SplitSeg{s, w.n_parts, parts, 0});
    for (int32_t j = 0; j < parts; ++j)
      w.items.push_back(WorkItem{s, off[s] + j * item_len,
                                 std::min(off[s] + (j + 1) * item_len, off[s + 1]), w.n_parts + j});
    w.n_parts += parts;
  }
  w.n_small = int32_t(w.splits.size());
  w.splits.insert(w.splits.end(), big.begin(), big.end());
  sort_items_by_length(w.items);
}

// ---- XCD-local work lists (dense data) -------------------------------------------------------
// MI355X has 8 XCDs with a 4 MiB L2 each, and workgroups are dealt to them round-robin (blocks b
// and b + 8 share an XCD).  A gathered table (theta, A) larger than one L2 misses on almost every
// row.  With long segments it pays to cut every segment at fixed borders of the GATHERED index
// (the triples of a segment are sorted by it), give range r of the table to XCD r mod 8, and order
// the work items so that every workgroup holds items of one range only and lands on that range's
// XCD: each L2 then serves a slice of the table that fits it.  The price is one partial row per
// (segment, range) piece, combined in piece order like any other split segment.
constexpr int32_t kXcds = 8;
constexpr size_t kRangeSliceBytes = size_t(3) << 20;  // table slice one XCD's 4 MiB L2 is asked to serve
constexpr int32_t kRangeMinMeanLen = 96;              // pieces of fewer than ~12 triples cost more than they save

// Number of ranges for a gathered table of `table_bytes` and segments of `mean_len` triples (1 = off).
// Measured (MI355X, 160-byte rows): 20M ratings x 138k users (145 each) gathering a 43 MB table: 8
// ranges -18 %, 16 ranges +18 %; the other pass of the same data (74 per segment): no gain at 8; 50M
// ratings x 88k pairs (565 each) gathering a 46 MB table: 8 ranges -21 %, 16 ranges -27 %, 32 -22 %.
inline int32_t range_count(size_t table_bytes, int64_t mean_len) {
  if (table_bytes <= kRangeSliceBytes) return 1;
  // shorter segments: fewer, wider ranges, each shared by 2 or 4 XCDs (pieces of >= ~12-18 triples);
  // 20M ratings, 74 per segment, 22 MB table: 4 ranges -10 %, 2 ranges -7.5 %, 8 ranges -1 %
  // (25 per segment, 2 ranges: +3 %; 50 per segment, 4 ranges: -13 %)
  if (mean_len < kRangeMinMeanLen) return mean_len >= 40 ? 4 : (mean_len >= 32 ? 2 : 1);
  int64_t n = kXcds;
  while (table_bytes / size_t(n) > kRangeSliceBytes && n < 64 && mean_len / (2 * n) >= 16) n *= 2;
  return int32_t(n);
}

// Share of all gathers that go to the `rows_fit` most frequently gathered rows (the rows an L2 would
// keep by itself).  off = offsets of the segments whose length is the row's gather count.  Heavy-
// tailed data has a hot set that already lives in L2: cutting by range then only adds partial rows
// (measured, Zipf(1.2) degrees, 20M ratings: 618 us without ranges, 764 us with).
inline double hot_fraction(const std::vector<int32_t> &off, int64_t rows_fit) {
  const int64_t rows = int64_t(off.size()) - 1;
  if (rows <= 0 || off.back() <= 0) return 1.0;
  if (rows_fit >= rows) return 1.0;
  std::vector<int32_t> deg(static_cast<size_t>(rows));
  for (int64_t r = 0; r < rows; ++r) deg[size_t(r)] = off[size_t(r) + 1] - off[size_t(r)];
  std::nth_element(deg.begin(), deg.begin() + rows_fit, deg.end(), [](int32_t a, int32_t b) { return a > b; });
  int64_t top = 0;
  for (int64_t r = 0; r < rows_fit; ++r) top += deg[size_t(r)];
  return double(top) / double(off.back());
}

// off: segment offsets; idx: gathered row of every triple (ascending inside a segment);
// per_block: work items (groups of lanes) per workgroup of the pass.
// `cuts` (optional, instead of idx): for every segment s the n_ranges + 1 positions
// cuts[s * (n_ranges + 1) + r] = first triple of s whose row lies in range >= r (so piece r of the
// segment is [cuts[r], cuts[r + 1])) -- what layout_gpu.hpp computes on the device, where idx lives.
inline void build_worklist_ranges(const std::vector<int32_t> &off, const int32_t *idx, int32_t table_rows,
                                  int32_t n_ranges, int32_t item_len, int32_t per_block, WorkList &w,
                                  const int32_t *cuts = nullptr) {
  w = WorkList();
  w.item_len = item_len;
  const int32_t nseg = int32_t(off.size()) - 1;
  // range of row r: r * n_ranges / table_rows; its first row:
  auto range_of = [&](int32_t r) { return int32_t(int64_t(r) * n_ranges / std::max(table_rows, 1)); };
  std::vector<std::vector<WorkItem>> bucket;
  bucket.resize(static_cast<size_t>(n_ranges));
  std::vector<SplitSeg> big;
  std::vector<std::pair<int32_t, int32_t>> pieces;  // (begin, range) of the current segment's pieces
  for (int32_t s = 0; s < nseg; ++s) {
    const int32_t b = off[s], e = off[s + 1];
    pieces.clear();
    if (b == e) {  // empty segment: its (zero) output row must still be written
      bucket[0].push_back(WorkItem{s, b, e, -1});
      continue;
    }
    const int32_t *cs = cuts ? cuts + size_t(s) * size_t(n_ranges + 1) : nullptr;
    if (e - b < 2 * n_ranges && e - b <= item_len) {  // a handful of triples: pieces of one or two would cost
      int32_t rm;                                     // more than they save: the middle triple's range takes it
      if (cs) {
        const int32_t mid = b + (e - b) / 2;
        rm = 0;
        while (rm + 1 < n_ranges && cs[rm + 1] <= mid) ++rm;
      } else {
        rm = range_of(idx[b + (e - b) / 2]);
      }
      bucket[size_t(rm)].push_back(WorkItem{s, b, e, -1});
      continue;  // (only these: a whole segment gathers from every range and pollutes its XCD's L2 --
    }            //  leaving all segments under 12 triples per range whole turned -33 % into +6 %)
    if (cs) {
      for (int32_t r = 0; r < n_ranges; ++r)
        for (int32_t c = cs[r]; c < cs[r + 1]; c += item_len) pieces.emplace_back(c, r);  // long pieces are cut again
    } else {
      int32_t t = b;
      while (t < e) {  // next border: first triple whose row leaves range r (binary search: rows ascend)
        const int32_t r = range_of(idx[t]);
        int32_t lo = t, hi = e;
        while (hi - lo > 1) {
          const int32_t mid = lo + (hi - lo) / 2;
          if (range_of(idx[mid]) == r) lo = mid; else hi = mid;
        }
        for (int32_t c = t; c < hi; c += item_len) pieces.emplace_back(c, r);  // long pieces are cut again
        t = hi;
      }
    }
    const int32_t np = int32_t(pieces.size());
    if (np == 1) {
      bucket[size_t(pieces[0].second)].push_back(WorkItem{s, b, e, -1});
      continue;
    }
    (np <= kSmallSplitParts ? w.splits : big).push_back(SplitSeg{s, w.n_parts, np, 0});
    for (int32_t j = 0; j < np; ++j) {
      const int32_t pb = pieces[size_t(j)].first;
      int32_t pe = (j + 1 < np) ? pieces[size_t(j) + 1].first : e;
      bucket[size_t(pieces[size_t(j)].second)].push_back(WorkItem{s, pb, pe, w.n_parts + j});
    }
    w.n_parts += np;
  }
  w.n_small = int32_t(w.splits.size());
  w.splits.insert(w.splits.end(), big.begin(), big.end());
  // Per XCD x: the blocks of ranges x, x + 8, x + 16, ... one after the other; then deal the eight
  // lists out round-robin (block j of XCD x becomes block 8 j + x).  Null items (seg = -1) pad.
  const WorkItem null_item{-1, 0, 0, -1};
  std::vector<std::vector<WorkItem>> per_xcd;
  per_xcd.resize(static_cast<size_t>(kXcds));
  // (fewer ranges than XCDs -- 4 or 2: range r is served by the XCDs r, r + n_ranges, ..., its
  // workgroups dealt out among them)
  const int32_t share = n_ranges < kXcds ? kXcds / n_ranges : 1;
  for (int32_t r = 0; r < n_ranges; ++r) {
    auto &src = bucket[size_t(r)];
    sort_items_by_length(src);
    while (src.size() % size_t(per_block)) src.push_back(null_item);
    const size_t nblk = src.size() / size_t(per_block);
    for (size_t j = 0; j < nblk; ++j) {
      auto &dst = per_xcd[size_t((r + int32_t(j % size_t(share)) * n_ranges) % kXcds)];
      dst.insert(dst.end(), src.begin() + j * size_t(per_block), src.begin() + (j + 1) * size_t(per_block));
    }
  }
  size_t rounds = 0;
  for (auto &v : per_xcd) rounds = std::max(rounds, v.size() / size_t(per_block));
  w.items.reserve(rounds * size_t(kXcds) * size_t(per_block));
  for (size_t j = 0; j < rounds; ++j)
    for (int32_t x = 0; x < kXcds; ++x) {
      const auto &v = per_xcd[size_t(x)];
      for (int32_t k = 0; k < per_block; ++k)
        w.items.push_back(j * size_t(per_block) + size_t(k) < v.size() ? v[j * size_t(per_block) + size_t(k)] : null_item);
    }
}

struct Layout {
  int64_t n_obs = 0;
  int32_t n_users = 0, n_items = 0, n_ratings = 0, n_pairs = 0;
  // pair order
  std::vector<int32_t> pair_off;     // n_pairs+1 : triple range of each pair
  std::vector<int32_t> pair_user;    // n_obs     : user of each triple (pair order)
  std::vector<int32_t> pair_item;    // n_pairs   : item of each pair
  std::vector<int32_t> rating_off;   // n_ratings+1 : pair range of each rating
  // user order
  std::vector<int32_t> user_off;     // n_users+1
  std::vector<int32_t> user_pair;    // n_obs : pair id of each triple (user order)
  // pairs of each item (ascending rating)
  std::vector<int32_t> item_off;     // n_items+1
  std::vector<int32_t> item_pairs;   // n_pairs
  std::vector<int32_t> item_deg;     // n_items : triples per item (NOT floored)
  // rating-homogeneous chunks of pairs
  std::vector<Chunk> chunks;
  std::vector<int32_t> chunk_off;    // n_ratings+1 : chunk range of each rating
  int32_t chunk_pairs = 0;           // max pairs per chunk
  // fixed-size (<= kMvChunkPairs) rating-homogeneous chunks for the lane-per-pair mat-vecs
  std::vector<Chunk> mv_chunks;
  std::vector<int32_t> mv_chunk_off;  // n_ratings+1
  WorkList pair_work, user_work;      // only filled when some segment is long
};

constexpr int32_t kMvChunkPairs = 64;  // default: one 64-pair unit per workgroup

// (Re)build the pair_block work list with `pairs` pairs per workgroup (a multiple of 64).  More
// pairs per workgroup = fewer K x L slabs, which matters once a slab is large (big K*L).
// Chunks (workgroups) a rating with `rating_pairs` pairs gets at `pairs` pairs per chunk, padding included.  Every
// rating's chunk count is padded to a multiple of 8 with EMPTY chunks (q_begin == q_end): chunk j of every rating then
// lands on the same XCD (workgroups are dealt to the 8 XCDs round-robin), and the rows of an item, which its R (item,
// rating) pairs all gather, are served to R - 1 of them by that XCD's L2.  (Speed only: an empty chunk's workgroup
// writes a zero slab and nothing else.)  The ONE statement of the rule: build_mv_chunks builds by it and the cost model
// of the matrix-core A launch (mmsbm_hip.hip: balanced_run_units) counts by it.
inline bool chunk_align_on() { return std::getenv("MMSBM_HIP_NO_CHUNK_ALIGN") == nullptr; }
inline int64_t padded_chunk_count(int64_t rating_pairs, int32_t pairs, int32_t n_ratings, bool align) {
  const int64_t chunks = (rating_pairs + pairs - 1) / pairs;
  return (align && n_ratings > 1 && rating_pairs > 0) ? (chunks + kXcds - 1) / kXcds * kXcds : chunks;
}
inline void build_mv_chunks(Layout &L, int32_t pairs) {
  const bool align = chunk_align_on();
  L.mv_chunks.clear();
  L.mv_chunk_off.assign(size_t(L.n_ratings) + 1, 0);
  for (int r = 0; r < L.n_ratings; ++r) {
    const int32_t beg = L.rating_off[r], end = L.rating_off[r + 1];
    for (int32_t q = beg; q < end; q += pairs)
      L.mv_chunks.push_back(Chunk{r, q, std::min<int32_t>(q + pairs, end), 0});
    const int64_t want = int64_t(L.mv_chunk_off[r]) + padded_chunk_count(end - beg, pairs, L.n_ratings, align);
    while (int64_t(L.mv_chunks.size()) < want) L.mv_chunks.push_back(Chunk{r, end, end, 0});
    L.mv_chunk_off[r + 1] = int32_t(L.mv_chunks.size());
  }
}

// ---- split segments inside ONE workgroup (the two-launch iteration of small problems, fused_small.hpp) --------------
// Data with uneven degrees (real rating tables: a few hundred users with ~100 ratings each, popular items with
// thousands) has its long segments cut into work items above, whose partial sums a separate combine launch adds up --
// and the two-launch iteration keeps C in LDS, so it needs every piece of a pair segment inside the workgroup that
// owns the pair's 64-pair unit.  These lists give every workgroup WHOLE segments: its work items in (segment, piece)
// order and the segments among them that are split; the pieces' partial rows then meet in LDS and are added in the
// order the combine kernels use (seg_pass.hpp), so the results are bit for bit those of the separate launches.
struct FusedUnit {   // one workgroup
  int32_t it_begin, it_end;   // its work items (FusedLists::items)
  int32_t sp_begin, sp_end;   // its split segments (FusedLists::splits)
};
struct FusedSplit {
  int32_t seg, first_part, n_parts, big;  // first_part: the unit-LOCAL partial row of piece 0; big: the strided order
};
struct FusedLists {
  std::vector<WorkItem> items;     // part = unit-local partial row, or -1: the item is the whole segment
  std::vector<FusedSplit> splits;
  std::vector<FusedUnit> units;
  int32_t max_parts = 0;           // partial rows a workgroup holds at most
};
// Every segment's pieces in piece order, whatever kind of work list cut them (none: one whole piece per segment).
struct SegPieces {
  std::vector<int32_t> first;      // [nseg + 1] into `pieces`
  std::vector<WorkItem> pieces;    // (seg, begin, end, part = the GLOBAL partial row or -1)
  std::vector<char> big;           // [nseg] combined by seg_combine_big (more than kSmallSplitParts pieces)
};
inline SegPieces segment_pieces(const std::vector<int32_t> &off, const WorkList &w) {
  const int32_t nseg = int32_t(off.size()) - 1;
  SegPieces sp;
  sp.first.assign(size_t(nseg) + 1, 0);
  sp.big.assign(size_t(std::max(nseg, 0)), 0);
  std::vector<int32_t> count(size_t(std::max(nseg, 0)), 1), first_part(size_t(std::max(nseg, 0)), -1);
  for (size_t j = 0; j < w.splits.size(); ++j) {
    const SplitSeg &s = w.splits[j];
    count[size_t(s.seg)] = s.n_parts;
    first_part[size_t(s.seg)] = s.first_part;
    sp.big[size_t(s.seg)] = int32_t(j) >= w.n_small;
  }
  std::vector<WorkItem> by_part(size_t(w.n_parts));
  for (const WorkItem &it : w.items)
    if (it.seg >= 0 && it.part >= 0) by_part[size_t(it.part)] = it;
  for (int32_t s = 0; s < nseg; ++s) sp.first[size_t(s) + 1] = sp.first[size_t(s)] + count[size_t(s)];
  sp.pieces.resize(size_t(sp.first[size_t(nseg)]));
  for (int32_t s = 0; s < nseg; ++s) {
    WorkItem *dst = sp.pieces.data() + sp.first[size_t(s)];
    if (first_part[size_t(s)] < 0) dst[0] = WorkItem{s, off[size_t(s)], off[size_t(s) + 1], -1};
    else for (int32_t j = 0; j < count[size_t(s)]; ++j) dst[j] = by_part[size_t(first_part[size_t(s)] + j)];
  }
  return sp;
}
// The pair_block / pairs_fused work list with at most `pairs` pairs AND at most `cap_items` work items per
// workgroup (a popular item's pairs bring several pieces each).  False -- and nothing changed -- when a single
// pair has more pieces than that.
inline bool build_mv_chunks_capped(Layout &L, const SegPieces &sp, int32_t pairs, int32_t cap_items) {
  for (int32_t q = 0; q < L.n_pairs; ++q)
    if (sp.first[size_t(q) + 1] - sp.first[size_t(q)] > cap_items) return false;
  const bool align = chunk_align_on();
  L.mv_chunks.clear();
  L.mv_chunk_off.assign(size_t(L.n_ratings) + 1, 0);
  for (int r = 0; r < L.n_ratings; ++r) {
    const int32_t end = L.rating_off[r + 1];
    int32_t q = L.rating_off[r];
    while (q < end) {
      int32_t e = q, items = 0;
      while (e < end && e - q < pairs && items + (sp.first[size_t(e) + 1] - sp.first[size_t(e)]) <= cap_items) {
        items += sp.first[size_t(e) + 1] - sp.first[size_t(e)];
        ++e;
      }
      L.mv_chunks.push_back(Chunk{r, q, e, 0});
      q = e;
    }
    if (align && L.n_ratings > 1 && end > L.rating_off[r])
      while ((L.mv_chunks.size() - size_t(L.mv_chunk_off[r])) % size_t(kXcds)) L.mv_chunks.push_back(Chunk{r, end, end, 0});
    L.mv_chunk_off[r + 1] = int32_t(L.mv_chunks.size());
  }
  return true;
}
inline void fused_append(FusedLists &out, const SegPieces &sp, int32_t s_begin, int32_t s_end) {
  FusedUnit u{int32_t(out.items.size()), 0, int32_t(out.splits.size()), 0};
  int32_t parts = 0;
  for (int32_t s = s_begin; s < s_end; ++s) {
    const int32_t a = sp.first[size_t(s)], n = sp.first[size_t(s) + 1] - a;
    if (n == 1 && sp.pieces[size_t(a)].part < 0) {
      out.items.push_back(sp.pieces[size_t(a)]);
      continue;
    }
    out.splits.push_back(FusedSplit{s, parts, n, sp.big[size_t(s)]});
    for (int32_t j = 0; j < n; ++j) {
      WorkItem it = sp.pieces[size_t(a + j)];
      it.part = parts + j;
      out.items.push_back(it);
    }
    parts += n;
  }
  u.it_end = int32_t(out.items.size());
  u.sp_end = int32_t(out.splits.size());
  out.units.push_back(u);
  out.max_parts = std::max(out.max_parts, parts);
}
// pair side: one unit per mv_chunk (its pairs' pieces)
inline FusedLists build_fused_pairs(const Layout &L, const SegPieces &sp) {
  FusedLists out;
  for (const Chunk &ch : L.mv_chunks) fused_append(out, sp, ch.q_begin, ch.q_end);
  return out;
}
// user side: consecutive segments while the workgroup's items stay within `cap_items` (= its groups of lanes: one
// round); a segment with more pieces than that gets a workgroup of its own, which makes several rounds.
inline FusedLists build_fused_users(const SegPieces &sp, int32_t cap_items) {
  FusedLists out;
  const int32_t nseg = int32_t(sp.first.size()) - 1;
  int32_t s = 0;
  while (s < nseg) {
    int32_t e = s, items = 0;
    while (e < nseg && (e == s || items + (sp.first[size_t(e) + 1] - sp.first[size_t(e)]) <= cap_items)) {
      items += sp.first[size_t(e) + 1] - sp.first[size_t(e)];
      ++e;
    }
    fused_append(out, sp, s, e);
    s = e;
  }
  return out;
}

// Host threads for the sorts below: one for small inputs, up to 8 for millions of triples.
inline int &layout_threads_override() {  // tests: force a thread count (0 = automatic)
  static int v = 0;
  return v;
}
inline int layout_threads(int64_t n_obs) {
  if (layout_threads_override() > 0) return layout_threads_override();
  if (n_obs < (int64_t(1) << 20)) return 1;
  return int(std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency())));
}

// fn(t, begin, end) over `threads` contiguous slices of [0, n)
template <class F>
inline void parallel_slices(int64_t n, int threads, F &&fn) {
  if (threads <= 1) {
    fn(0, int64_t(0), n);
    return;
  }
  std::vector<std::thread> th;
  const int64_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    const int64_t a = std::min<int64_t>(n, t * per), b = std::min<int64_t>(n, a + per);
    th.emplace_back([&fn, t, a, b] { fn(t, a, b); });
  }
  for (auto &x : th) x.join();
}

// Stable counting sort of the elements j = 0..n-1 by key(j) in [0, nkeys): calls emit(j, position)
// with the element's rank; `totals` (nkeys + 1) receives the exclusive prefix sums of the key
// counts.  Slices of the input get private histograms, so threads never share a counter and
// the order of equal keys is the input order.
template <class KeyFn, class EmitFn>
inline void counting_sort(int64_t n, size_t nkeys, int threads, KeyFn &&key, EmitFn &&emit,
                          std::vector<int32_t> &totals) {
  while (threads > 1 && uint64_t(threads) * nkeys > (uint64_t(1) << 26)) threads /= 2;  // <= 256 MB of counters
  std::vector<std::vector<uint32_t>> hist(size_t(threads), std::vector<uint32_t>(nkeys, 0));
  parallel_slices(n, threads, [&](int t, int64_t a, int64_t b) {
    uint32_t *h = hist[size_t(t)].data();
    for (int64_t j = a; j < b; ++j) h[key(j)]++;
  });
  totals.assign(nkeys + 1, 0);
  uint32_t run = 0;
  for (size_t k = 0; k < nkeys; ++k) {  // histogram -> first position of (key k, slice t)
    totals[k] = int32_t(run);
    for (int t = 0; t < threads; ++t) {
      const uint32_t c = hist[size_t(t)][k];
      hist[size_t(t)][k] = run;
      run += c;
    }
  }
  totals[nkeys] = int32_t(run);
  parallel_slices(n, threads, [&](int t, int64_t a, int64_t b) {
    uint32_t *h = hist[size_t(t)].data();
    for (int64_t j = a; j < b; ++j) emit(j, int64_t(h[key(j)]++));
  });
}

// Argument and id-range checks shared by the host and the device layout builders.
inline void validate_triples(int64_t n_obs, int32_t n_users, int32_t n_items, int32_t n_ratings,
                             const int32_t *user, const int32_t *item, const int32_t *rating) {
  if (n_obs < 0 || n_obs >= (int64_t(1) << 31) - 64)
    throw std::invalid_argument("n_obs must be in [0, 2^31)");
  if (n_users <= 0 || n_items <= 0 || n_ratings <= 0)
    throw std::invalid_argument("n_users, n_items and n_ratings must be positive");
  if (n_obs > 0 && (!user || !item || !rating))
    throw std::invalid_argument("null triple array");
  const int threads = layout_threads(n_obs);
  std::vector<int64_t> bad(size_t(threads), -1);  // first offending triple of each slice
  parallel_slices(n_obs, threads, [&](int t, int64_t a, int64_t b) {
    for (int64_t n = a; n < b; ++n)
      if (user[n] < 0 || user[n] >= n_users || item[n] < 0 || item[n] >= n_items ||
          rating[n] < 0 || rating[n] >= n_ratings) {
        bad[size_t(t)] = n;
        return;
      }
  });
  for (int64_t n : bad)
    if (n >= 0) throw std::invalid_argument("triple " + std::to_string(n) + " has an id out of range");
}

// What follows the sorts (needs pair_off, rating_off, user_off only): the rating-homogeneous chunks
// and the work lists.  Shared by the host builder below and the device builder (layout_gpu.hpp).
inline void finish_layout(Layout &L, int32_t target_chunks) {
  const int32_t n_ratings = L.n_ratings;
  if (target_chunks < 1) target_chunks = 1;
  int32_t cp = int32_t((int64_t(L.n_pairs) + target_chunks - 1) / target_chunks);
  cp = std::max<int32_t>(cp, 64);
  cp = (cp + 15) / 16 * 16;
  L.chunk_pairs = cp;
  L.chunks.clear();
  L.chunk_off.assign(size_t(n_ratings) + 1, 0);
  for (int r = 0; r < n_ratings; ++r) {
    for (int32_t q = L.rating_off[r]; q < L.rating_off[r + 1]; q += cp)
      L.chunks.push_back(Chunk{r, q, std::min<int32_t>(q + cp, L.rating_off[r + 1]), 0});
    L.chunk_off[r + 1] = int32_t(L.chunks.size());
  }
  build_mv_chunks(L, kMvChunkPairs);
  build_worklist(L.pair_off, L.pair_work, item_length(L.n_obs, L.n_pairs), lengths_vary(L.pair_off));
  build_worklist(L.user_off, L.user_work, item_length(L.n_obs, L.n_users), lengths_vary(L.user_off));
}

inline void build_layout(int64_t n_obs, int32_t n_users, int32_t n_items, int32_t n_ratings,
                         const int32_t *user, const int32_t *item, const int32_t *rating,
                         int32_t target_chunks, Layout &out) {
  validate_triples(n_obs, n_users, n_items, n_ratings, user, item, rating);
  const int threads = layout_threads(n_obs);
  Layout &L = out;
  L = Layout();
  L.n_obs = n_obs; L.n_users = n_users; L.n_items = n_items; L.n_ratings = n_ratings;

  // ---- sort by (rating, item, user) -------------------------------------------------
  // Two stable counting sorts (user, then pair key) when the key space is small enough --
  // O(N + U + R*I) -- else a comparison sort.
  struct Key { uint64_t pk; uint32_t u; };
  std::vector<Key> keys(static_cast<size_t>(n_obs));
  const uint64_t key_space = uint64_t(n_ratings) * uint64_t(n_items);
  if (key_space <= (uint64_t(1) << 26)) {
    std::vector<uint32_t> by_user(static_cast<size_t>(n_obs));
    std::vector<int32_t> unused;
    counting_sort(n_obs, size_t(n_users), threads, [&](int64_t n) { return size_t(user[n]); },
                  [&](int64_t n, int64_t pos) { by_user[size_t(pos)] = uint32_t(n); }, unused);
    auto pk_of = [&](uint32_t n) { return uint64_t(rating[n]) * uint64_t(n_items) + uint64_t(item[n]); };
    counting_sort(n_obs, size_t(key_space), threads,
                  [&](int64_t j) { return size_t(pk_of(by_user[size_t(j)])); },
                  [&](int64_t j, int64_t pos) {
                    const uint32_t n = by_user[size_t(j)];
                    keys[size_t(pos)] = Key{pk_of(n), uint32_t(user[n])};
                  },
                  unused);
  } else {
    for (int64_t n = 0; n < n_obs; ++n)
      keys[n] = Key{uint64_t(rating[n]) * uint64_t(n_items) + uint64_t(item[n]), uint32_t(user[n])};
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
      return a.pk != b.pk ? a.pk < b.pk : a.u < b.u;
    });
  }

  L.pair_user.resize(n_obs);
  L.pair_off.clear(); L.pair_item.clear();
  L.rating_off.assign(size_t(n_ratings) + 1, 0);
  std::vector<int32_t> triple_pair(static_cast<size_t>(n_obs));  // pair id per sorted triple
  std::vector<int32_t> pairs_per_rating(n_ratings, 0);
  uint64_t prev = ~uint64_t(0);
  for (int64_t n = 0; n < n_obs; ++n) {
    if (keys[n].pk != prev) {
      prev = keys[n].pk;
      L.pair_off.push_back(int32_t(n));
      L.pair_item.push_back(int32_t(prev % uint64_t(n_items)));
      pairs_per_rating[size_t(prev / uint64_t(n_items))]++;
    }
    triple_pair[n] = int32_t(L.pair_off.size()) - 1;
    L.pair_user[n] = int32_t(keys[n].u);
  }
  L.n_pairs = int32_t(L.pair_off.size());
  L.pair_off.push_back(int32_t(n_obs));
  for (int r = 0; r < n_ratings; ++r) L.rating_off[r + 1] = L.rating_off[r] + pairs_per_rating[r];

  // ---- user order: stable counting sort of the pair-ordered triples by user ---------
  L.user_pair.resize(n_obs);
  counting_sort(n_obs, size_t(n_users), threads, [&](int64_t n) { return size_t(L.pair_user[size_t(n)]); },
                [&](int64_t n, int64_t pos) { L.user_pair[size_t(pos)] = triple_pair[size_t(n)]; }, L.user_off);

  // ---- pairs of each item + item degrees ----------------------------------------------
  L.item_off.assign(size_t(n_items) + 1, 0);
  L.item_deg.assign(n_items, 0);
  for (int q = 0; q < L.n_pairs; ++q) {
    L.item_off[size_t(L.pair_item[q]) + 1]++;
    L.item_deg[L.pair_item[q]] += L.pair_off[q + 1] - L.pair_off[q];
  }
  for (int i = 0; i < n_items; ++i) L.item_off[i + 1] += L.item_off[i];
  L.item_pairs.resize(L.n_pairs);
  {
    std::vector<int32_t> cur(L.item_off.begin(), L.item_off.end() - 1);
    for (int q = 0; q < L.n_pairs; ++q) L.item_pairs[cur[L.pair_item[q]]++] = q;
  }

  finish_layout(L, target_chunks);
}

}  // namespace mmsbm

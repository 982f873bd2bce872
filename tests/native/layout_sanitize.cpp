// Host-only driver for mmsbm_amd/csrc/layout.hpp, built with -fsanitize=address,undefined by
// tests/test_layout_sanitizers.py (sanitizers run on the CPU build only).  Exercises random,
// skewed, degenerate and invalid inputs and re-checks the layout's invariants in C++.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../mmsbm_amd/csrc/layout.hpp"

static int fails = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED %s line %d\n", #x, __LINE__); ++fails; } } while (0)

static void check(const std::vector<int32_t> &u, const std::vector<int32_t> &i, const std::vector<int32_t> &r,
                  int U, int I, int R, int target) {
  mmsbm::Layout L;
  mmsbm::build_layout(static_cast<int64_t>(u.size()), U, I, R, u.data(), i.data(), r.data(), target, L);
  const int64_t n = static_cast<int64_t>(u.size());
  CHECK(L.pair_off.front() == 0 && L.pair_off.back() == n);
  CHECK(L.user_off.front() == 0 && L.user_off.back() == n);
  CHECK(static_cast<int>(L.pair_item.size()) == L.n_pairs);
  CHECK(L.rating_off.back() == L.n_pairs && L.item_off.back() == L.n_pairs);
  int64_t deg = 0;
  for (int x : L.item_deg) deg += x;
  CHECK(deg == n);
  for (int q = 0; q + 1 < L.n_pairs; ++q) CHECK(L.pair_off[q] < L.pair_off[q + 1]);
  for (int64_t t = 0; t < n; ++t) {
    CHECK(L.pair_user[t] >= 0 && L.pair_user[t] < U);
    CHECK(L.user_pair[t] >= 0 && L.user_pair[t] < L.n_pairs);
  }
  std::vector<int> cover(L.n_pairs, 0);
  for (const auto &c : L.mv_chunks) {
    CHECK(c.q_begin <= c.q_end && c.q_end - c.q_begin <= mmsbm::kMvChunkPairs);
    CHECK(L.rating_off[c.rating] <= c.q_begin && c.q_end <= L.rating_off[c.rating + 1]);
    for (int q = c.q_begin; q < c.q_end; ++q) cover[q]++;
  }
  for (int q = 0; q < L.n_pairs; ++q) CHECK(cover[q] == 1);
  // empty chunks are padding only: at the tail of a rating's list, which they bring to a multiple of the XCD count
  for (int rr = 0; rr < R && L.n_ratings > 1; ++rr) {
    const int a = L.mv_chunk_off[rr], b = L.mv_chunk_off[rr + 1];
    CHECK((b - a) % mmsbm::kXcds == 0);
    CHECK((b == a) == (L.rating_off[rr] == L.rating_off[rr + 1]));
    int pads = 0;
    for (int k = a; k < b; ++k) {
      const auto &c = L.mv_chunks[size_t(k)];
      CHECK(c.rating == rr);
      if (c.q_begin == c.q_end) { ++pads; CHECK(c.q_begin == L.rating_off[rr + 1]); }
      else CHECK(pads == 0);
    }
    CHECK(pads < mmsbm::kXcds);
  }
  for (const mmsbm::WorkList *w : {&L.pair_work, &L.user_work}) {
    const auto &off = (w == &L.pair_work) ? L.pair_off : L.user_off;
    std::vector<int> seen(static_cast<size_t>(n), 0);
    for (const auto &it : w->items) {
      CHECK(it.end - it.begin <= w->item_len && it.begin <= it.end);  // empty segments keep an (empty) item: their output row must still be written
      CHECK(off[it.seg] <= it.begin && it.end <= off[it.seg + 1]);
      CHECK(it.part < w->n_parts);
      for (int t = it.begin; t < it.end; ++t) seen[t]++;
    }
    if (!w->items.empty())
      for (int64_t t = 0; t < n; ++t) CHECK(seen[t] == 1);
  }
  mmsbm::build_mv_chunks(L, 4 * mmsbm::kMvChunkPairs);
  CHECK(L.mv_chunk_off.back() == static_cast<int>(L.mv_chunks.size()));
  // XCD-local work lists: both passes, 8 and 24 ranges, workgroups of 8 and 32 items
  for (int side = 0; side < 2; ++side)
    for (int n_ranges : {4, 8, 24})
      for (int per_block : {8, 32}) {
        const auto &off = side ? L.user_off : L.pair_off;
        const int32_t *idx = side ? L.user_pair.data() : L.pair_user.data();
        const int rows = side ? L.n_pairs : L.n_users;
        mmsbm::WorkList w;
        mmsbm::build_worklist_ranges(off, idx, rows, n_ranges, 50, per_block, w);
        const int nseg = static_cast<int>(off.size()) - 1;
        CHECK(w.items.size() % (static_cast<size_t>(per_block) * mmsbm::kXcds) == 0);
        std::vector<int> seen(static_cast<size_t>(n), 0), out_rows(static_cast<size_t>(std::max(nseg, 1)), 0);
        std::vector<int> part_owner(static_cast<size_t>(w.n_parts), -1);
        for (size_t k = 0; k < w.items.size(); ++k) {
          const auto &it = w.items[k];
          if (it.seg < 0) continue;
          CHECK(it.seg < nseg && off[it.seg] <= it.begin && it.begin <= it.end && it.end <= off[it.seg + 1]);
          CHECK(it.end - it.begin <= 50);
          const int block = static_cast<int>(k / static_cast<size_t>(per_block));
          const bool whole_short = it.part < 0 && it.end - it.begin < 2 * n_ranges;  // a handful of triples: not cut
          for (int t = it.begin; t < it.end; ++t) {
            seen[t]++;
            const int r = static_cast<int>(static_cast<int64_t>(idx[t]) * n_ranges / std::max(rows, 1));
            if (!whole_short) CHECK(r % std::min(n_ranges, mmsbm::kXcds) == (block % mmsbm::kXcds) % std::min(n_ranges, mmsbm::kXcds));  // the workgroup lands on (one of) its range's XCD(s)
          }
          if (it.part < 0) out_rows[it.seg]++;
          else { CHECK(it.part < w.n_parts && part_owner[it.part] == -1); part_owner[it.part] = it.seg; }
        }
        for (int64_t t = 0; t < n; ++t) CHECK(seen[t] == 1);
        for (const auto &sp : w.splits) {
          CHECK(sp.n_parts >= 2 && out_rows[sp.seg] == 0);
          for (int j = 0; j < sp.n_parts; ++j) CHECK(part_owner[sp.first_part + j] == sp.seg);
          out_rows[sp.seg] = 1;
        }
        for (int s2 = 0; s2 < nseg; ++s2) CHECK(out_rows[s2] == 1);  // every segment's row is written exactly once
        for (int k = 0; k < w.n_small; ++k) CHECK(w.splits[k].n_parts <= mmsbm::kSmallSplitParts);
        for (size_t k = w.n_small; k < w.splits.size(); ++k) CHECK(w.splits[k].n_parts > mmsbm::kSmallSplitParts);
      }
}

int main() {
  std::mt19937 rng(5);
  for (int trial = 0; trial < 40; ++trial) {
    const int U = 1 + rng() % 300, I = 1 + rng() % 60, R = 1 + rng() % 7;
    const int n = static_cast<int>(rng() % 4000);
    std::vector<int32_t> u(n), i(n), r(n);
    const bool skew = trial % 3 == 0;
    for (int t = 0; t < n; ++t) {
      u[t] = (skew && rng() % 2) ? 0 : static_cast<int32_t>(rng() % U);
      i[t] = (skew && rng() % 3 == 0) ? 0 : static_cast<int32_t>(rng() % I);
      r[t] = static_cast<int32_t>(rng() % R);
    }
    check(u, i, r, U, I, R, 1 + static_cast<int>(rng() % 50));
  }
  {  // the threaded sorts (forced on a mid-sized input) must give exactly the single-thread layout
    const int n = 300000, U = 5000, I = 700, R = 6;
    std::vector<int32_t> u(n), i(n), r(n);
    for (int t = 0; t < n; ++t) {
      u[t] = (rng() % 5 == 0) ? 3 : static_cast<int32_t>(rng() % U);
      i[t] = static_cast<int32_t>(rng() % I);
      r[t] = static_cast<int32_t>(rng() % R);
    }
    mmsbm::Layout a, b, c;
    mmsbm::layout_threads_override() = 1;
    mmsbm::build_layout(n, U, I, R, u.data(), i.data(), r.data(), 64, a);
    mmsbm::layout_threads_override() = 8;
    mmsbm::build_layout(n, U, I, R, u.data(), i.data(), r.data(), 64, b);
    mmsbm::layout_threads_override() = 3;
    mmsbm::build_layout(n, U, I, R, u.data(), i.data(), r.data(), 64, c);
    mmsbm::layout_threads_override() = 0;
    for (const mmsbm::Layout *x : {&b, &c}) {
      CHECK(a.pair_off == x->pair_off && a.pair_user == x->pair_user && a.pair_item == x->pair_item);
      CHECK(a.user_off == x->user_off && a.user_pair == x->user_pair && a.rating_off == x->rating_off);
      CHECK(a.item_off == x->item_off && a.item_pairs == x->item_pairs && a.item_deg == x->item_deg);
    }
    check(u, i, r, U, I, R, 64);
    mmsbm::layout_threads_override() = 8;
    check(u, i, r, U, I, R, 64);
    mmsbm::layout_threads_override() = 0;
  }
  {  // the policies, at the shapes they were tuned on
    const size_t MB = size_t(1) << 20;
    CHECK(mmsbm::range_count(43 * MB, 145) == 8);    // 20M ratings x 138k users: the user pass
    CHECK(mmsbm::range_count(22 * MB, 74) == 4);     // ... its pair pass: shorter segments, wider ranges
    CHECK(mmsbm::range_count(22 * MB, 35) == 2 && mmsbm::range_count(22 * MB, 25) == 1);
    CHECK(mmsbm::range_count(46 * MB, 565) == 16);   // 50M ratings x 88k pairs
    CHECK(mmsbm::range_count(17 * MB / 2, 104) == 8);
    CHECK(mmsbm::range_count(2 * MB, 5000) == 1);    // the table fits an L2 anyway
    CHECK(mmsbm::range_count(16 * MB, 10) == 1);     // BASELINE's sparse configs
    CHECK(mmsbm::item_length(1000000, 100000) == 64 && mmsbm::item_length(1000000, 6040) == 16);
    std::vector<int32_t> flat{0}, skew{0};
    for (int s = 0; s < 4000; ++s) flat.push_back(flat.back() + 8 + static_cast<int>(rng() % 5));
    for (int s = 0; s < 4000; ++s) skew.push_back(skew.back() + (s % 50 == 0 ? 900 : 3 + static_cast<int>(rng() % 4)));
    CHECK(!mmsbm::lengths_vary(flat) && mmsbm::lengths_vary(skew));
    const double hf = mmsbm::hot_fraction(flat, 400), hs = mmsbm::hot_fraction(skew, 400);
    CHECK(hf > 0.10 && hf < 0.13 && hs > 0.7 && mmsbm::hot_fraction(flat, 5000) == 1.0);
    mmsbm::WorkList w;
    mmsbm::build_worklist(skew, w, 64, true);
    CHECK(!w.items.empty());
    int prev = 1 << 30;                                 // length classes descend through the list
    for (const auto &it : w.items) {
      int len = it.end - it.begin, c = 0;
      while (len > 1) { len >>= 1; ++c; }
      CHECK(c <= prev);
      prev = c;
    }
    mmsbm::build_worklist(flat, w, 64, true);           // nearly equal lengths: segment order is kept
    for (size_t k = 1; k < w.items.size(); ++k) CHECK(w.items[k - 1].seg < w.items[k].seg);
  }
  check({}, {}, {}, 3, 2, 2, 8);            // empty
  check({0}, {0}, {0}, 1, 1, 1, 1);          // single triple
  {                                          // invalid ids must throw, not scribble
    std::vector<int32_t> u{0, 9}, i{0, 0}, r{0, 0};
    bool threw = false;
    try { mmsbm::Layout L; mmsbm::build_layout(2, 5, 1, 1, u.data(), i.data(), r.data(), 4, L); }
    catch (const std::invalid_argument &) { threw = true; }
    CHECK(threw);
  }
  std::printf(fails ? "layout sanitize: %d failures\n" : "layout sanitize: ok\n", fails);
  return fails ? 1 : 0;
}

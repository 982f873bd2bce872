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
    CHECK(c.q_begin < c.q_end && c.q_end - c.q_begin <= mmsbm::kMvChunkPairs);
    CHECK(L.rating_off[c.rating] <= c.q_begin && c.q_end <= L.rating_off[c.rating + 1]);
    for (int q = c.q_begin; q < c.q_end; ++q) cover[q]++;
  }
  for (int q = 0; q < L.n_pairs; ++q) CHECK(cover[q] == 1);
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

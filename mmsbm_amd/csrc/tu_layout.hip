// tu_layout.hip -- translation unit of the layout's sort stage on the device (declared in layout_gpu.hpp)
#include "prelude.hpp"
#include "layout_gpu.hpp"

#include <rocprim/rocprim.hpp>

namespace mmsbm {
namespace gpu_layout {

constexpr int kThreads = 256;

static void check(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
static unsigned blocks_for(int64_t n) { return static_cast<unsigned>((n + kThreads - 1) / kThreads); }

template <class T>
struct Buf {  // plain device allocation, freed on scope exit
  T *p = nullptr;
  explicit Buf(size_t n) { check(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)), "hipMalloc"); }
  Buf(const Buf &) = delete;
  Buf &operator=(const Buf &) = delete;
  ~Buf() { if (p) (void)hipFree(p); }
  T *release() { T *r = p; p = nullptr; return r; }
};

static __global__ __launch_bounds__(kThreads) void make_keys(const int32_t *__restrict__ user,
                                                      const int32_t *__restrict__ item,
                                                      const int32_t *__restrict__ rating, uint64_t n_items,
                                                      int64_t n, uint64_t *__restrict__ key) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (t < n)
    key[t] = ((static_cast<uint64_t>(rating[t]) * n_items + static_cast<uint64_t>(item[t])) << 32) |
             static_cast<uint32_t>(user[t]);
}

// sorted keys -> user of each triple and "a new pair starts here" flags
static __global__ __launch_bounds__(kThreads) void split_keys(const uint64_t *__restrict__ key, int64_t n,
                                                       int32_t *__restrict__ pair_user,
                                                       int32_t *__restrict__ head) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= n) return;
  const uint64_t k = key[t];
  pair_user[t] = static_cast<int32_t>(static_cast<uint32_t>(k));
  head[t] = (t == 0 || (key[t - 1] >> 32) != (k >> 32)) ? 1 : 0;
}

// scan[t] = pair id + 1 of triple t.  Heads record their pair's first triple, item and rating;
// every triple gets its pair id.
static __global__ __launch_bounds__(kThreads) void scatter_heads(const uint64_t *__restrict__ key,
                                                          const int32_t *__restrict__ head,
                                                          const int32_t *__restrict__ scan, int64_t n,
                                                          uint64_t n_items, int32_t *__restrict__ triple_pair,
                                                          int32_t *__restrict__ pair_off,
                                                          int32_t *__restrict__ pair_item,
                                                          int32_t *__restrict__ pair_rating) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= n) return;
  const int32_t q = scan[t] - 1;
  triple_pair[t] = q;
  if (head[t]) {
    const uint64_t pk = key[t] >> 32;
    pair_off[q] = static_cast<int32_t>(t);
    pair_item[q] = static_cast<int32_t>(pk % n_items);
    pair_rating[q] = static_cast<int32_t>(pk / n_items);
  }
}

static __global__ __launch_bounds__(kThreads) void iota(int32_t *__restrict__ out, int64_t n) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (t < n) out[t] = static_cast<int32_t>(t);
}

// off[x] = number of sorted keys < x, for x = 0 .. n_keys (n_keys + 1 outputs)
static __global__ __launch_bounds__(kThreads) void lower_bounds(const int32_t *__restrict__ sorted, int64_t n,
                                                         int32_t n_keys, int32_t *__restrict__ off) {
  const int32_t x = static_cast<int32_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (x > n_keys) return;
  int64_t lo = 0, hi = n;  // first index with sorted[idx] >= x
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < x) lo = mid + 1; else hi = mid;
  }
  off[x] = static_cast<int32_t>(lo);
}

// triples per item = sum of the sizes of its pairs
static __global__ __launch_bounds__(kThreads) void item_degrees(const int32_t *__restrict__ item_off,
                                                         const int32_t *__restrict__ item_pairs,
                                                         const int32_t *__restrict__ pair_off, int32_t n_items,
                                                         int32_t *__restrict__ item_deg) {
  const int32_t i = static_cast<int32_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n_items) return;
  int32_t d = 0;
  for (int32_t j = item_off[i]; j < item_off[i + 1]; ++j) {
    const int32_t q = item_pairs[j];
    d += pair_off[q + 1] - pair_off[q];
  }
  item_deg[i] = d;
}

// XCD-local work lists (layout.hpp: build_worklist_ranges): where every segment crosses the borders of
// the n_ranges row ranges.  Thread = (segment, r): first triple of the segment whose gathered row lies
// in range >= r, i.e. row >= ceil(r * table_rows / n_ranges); r = 0 .. n_ranges.
static __global__ __launch_bounds__(kThreads) void range_cuts_kernel(const int32_t *__restrict__ off,
                                                              const int32_t *__restrict__ idx, int32_t n_seg,
                                                              int32_t table_rows, int32_t n_ranges,
                                                              int32_t *__restrict__ cuts) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int64_t per = n_ranges + 1;
  if (t >= static_cast<int64_t>(n_seg) * per) return;
  const int32_t s = static_cast<int32_t>(t / per), r = static_cast<int32_t>(t - static_cast<int64_t>(s) * per);
  const int64_t first_row = (static_cast<int64_t>(r) * table_rows + n_ranges - 1) / n_ranges;
  int32_t lo = off[s], hi = off[s + 1];
  while (lo < hi) {
    const int32_t mid = lo + (hi - lo) / 2;
    if (idx[mid] < first_row) lo = mid + 1; else hi = mid;
  }
  cuts[t] = lo;
}

template <class T>
inline void to_host(std::vector<T> &dst, const T *src, size_t n, hipStream_t s) {
  dst.resize(n);
  if (n) check(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, s), "hipMemcpyAsync (D2H)");
}

void sort_stage(hipStream_t s, int64_t n_obs, int32_t n_users, int32_t n_items, int32_t n_ratings,
                       const int32_t *d_user, const int32_t *d_item, const int32_t *d_rating,
                       Layout &L, DeviceArrays &dev) {
  L = Layout();
  L.n_obs = n_obs; L.n_users = n_users; L.n_items = n_items; L.n_ratings = n_ratings;
  const int64_t n = n_obs;
  const uint64_t key_space = static_cast<uint64_t>(n_ratings) * static_cast<uint64_t>(n_items);
  if (key_space >= (uint64_t(1) << 31)) throw std::invalid_argument("ratings x items must be below 2^31");
  int pk_bits = 1;
  while ((uint64_t(1) << pk_bits) < key_space) ++pk_bits;
  int user_bits = 1;
  while ((int64_t(1) << user_bits) < n_users) ++user_bits;
  int item_bits = 1;
  while ((int64_t(1) << item_bits) < n_items) ++item_bits;

  Buf<int32_t> pair_user(n), user_pair(n);
  Buf<int32_t> triple_pair(n);
  int32_t n_pairs = 0;
  {
    // ---- pair order -------------------------------------------------------------------------
    Buf<uint64_t> key_in(n), key_out(n);
    Buf<int32_t> head(n), scan(n);
    LAUNCH(make_keys, blocks_for(n), kThreads, 0, s, d_user, d_item, d_rating, static_cast<uint64_t>(n_items), n, key_in.p);
    size_t tmp_bytes = 0;
    check(rocprim::radix_sort_keys(nullptr, tmp_bytes, key_in.p, key_out.p, static_cast<size_t>(n), 0u, static_cast<unsigned>(32 + pk_bits), s), "radix sort (size)");
    size_t scan_bytes = 0;
    check(rocprim::inclusive_scan(nullptr, scan_bytes, head.p, scan.p, static_cast<size_t>(n), rocprim::plus<int32_t>(), s), "scan (size)");
    Buf<char> tmp(std::max(tmp_bytes, scan_bytes));
    size_t tb = std::max(tmp_bytes, scan_bytes);
    check(rocprim::radix_sort_keys(tmp.p, tb, key_in.p, key_out.p, static_cast<size_t>(n), 0u, static_cast<unsigned>(32 + pk_bits), s), "radix sort");
    LAUNCH(split_keys, blocks_for(n), kThreads, 0, s, key_out.p, n, pair_user.p, head.p);
    tb = std::max(tmp_bytes, scan_bytes);
    check(rocprim::inclusive_scan(tmp.p, tb, head.p, scan.p, static_cast<size_t>(n), rocprim::plus<int32_t>(), s), "scan");
    if (n > 0) {
      check(hipMemcpyAsync(&n_pairs, scan.p + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s), "n_pairs");
      check(hipStreamSynchronize(s), "sync");
    }
    L.n_pairs = n_pairs;
    Buf<int32_t> pair_off(static_cast<size_t>(n_pairs) + 1), pair_item(n_pairs), pair_rating(n_pairs);
    LAUNCH(scatter_heads, blocks_for(n), kThreads, 0, s, key_out.p, head.p, scan.p, n, static_cast<uint64_t>(n_items),
                                                     triple_pair.p, pair_off.p, pair_item.p, pair_rating.p);
    const int32_t n32 = static_cast<int32_t>(n);
    check(hipMemcpyAsync(pair_off.p + n_pairs, &n32, sizeof(int32_t), hipMemcpyHostToDevice, s), "pair_off end");
    // rating_off: the pairs are rating-major
    Buf<int32_t> rating_off(static_cast<size_t>(n_ratings) + 1);
    LAUNCH(lower_bounds, blocks_for(n_ratings + 1), kThreads, 0, s, pair_rating.p, n_pairs, n_ratings, rating_off.p);
    // ---- pairs of each item -----------------------------------------------------------------
    Buf<int32_t> q_iota(n_pairs), item_sorted(n_pairs), item_pairs(n_pairs);
    Buf<int32_t> item_off(static_cast<size_t>(n_items) + 1), item_deg(n_items);
    LAUNCH(iota, blocks_for(n_pairs), kThreads, 0, s, q_iota.p, n_pairs);
    size_t ip_bytes = 0;
    check(rocprim::radix_sort_pairs(nullptr, ip_bytes, pair_item.p, item_sorted.p, q_iota.p, item_pairs.p,
                                    static_cast<size_t>(n_pairs), 0u, static_cast<unsigned>(item_bits), s), "item sort (size)");
    {
      Buf<char> tmp2(ip_bytes);
      check(rocprim::radix_sort_pairs(tmp2.p, ip_bytes, pair_item.p, item_sorted.p, q_iota.p, item_pairs.p,
                                      static_cast<size_t>(n_pairs), 0u, static_cast<unsigned>(item_bits), s), "item sort");
      LAUNCH(lower_bounds, blocks_for(n_items + 1), kThreads, 0, s, item_sorted.p, n_pairs, n_items, item_off.p);
      LAUNCH(item_degrees, blocks_for(n_items), kThreads, 0, s, item_off.p, item_pairs.p, pair_off.p, n_items, item_deg.p);
      to_host(L.pair_off, pair_off.p, static_cast<size_t>(n_pairs) + 1, s);
      to_host(L.pair_item, pair_item.p, n_pairs, s);
      to_host(L.rating_off, rating_off.p, static_cast<size_t>(n_ratings) + 1, s);
      to_host(L.item_off, item_off.p, static_cast<size_t>(n_items) + 1, s);
      to_host(L.item_pairs, item_pairs.p, n_pairs, s);
      to_host(L.item_deg, item_deg.p, n_items, s);
      check(hipStreamSynchronize(s), "sync");
    }
  }
  {
    // ---- user order: stable sort of the pair-ordered triples by user ----------------------------
    Buf<int32_t> user_sorted(n), user_off(static_cast<size_t>(n_users) + 1);
    size_t up_bytes = 0;
    check(rocprim::radix_sort_pairs(nullptr, up_bytes, pair_user.p, user_sorted.p, triple_pair.p, user_pair.p,
                                    static_cast<size_t>(n), 0u, static_cast<unsigned>(user_bits), s), "user sort (size)");
    Buf<char> tmp3(up_bytes);
    check(rocprim::radix_sort_pairs(tmp3.p, up_bytes, pair_user.p, user_sorted.p, triple_pair.p, user_pair.p,
                                    static_cast<size_t>(n), 0u, static_cast<unsigned>(user_bits), s), "user sort");
    LAUNCH(lower_bounds, blocks_for(n_users + 1), kThreads, 0, s, user_sorted.p, n, n_users, user_off.p);
    to_host(L.user_off, user_off.p, static_cast<size_t>(n_users) + 1, s);
    check(hipStreamSynchronize(s), "sync");
  }
  check(hipGetLastError(), "layout kernels");
  dev.pair_user = pair_user.release();
  dev.user_pair = user_pair.release();
}

// cuts[s * (n_ranges + 1) + r] for the segments `off` (host copy) over the device index array d_idx
std::vector<int32_t> range_cuts(hipStream_t s, const std::vector<int32_t> &off, const int32_t *d_idx,
                                       int32_t table_rows, int32_t n_ranges) {
  const int32_t n_seg = static_cast<int32_t>(off.size()) - 1;
  const int64_t total = static_cast<int64_t>(n_seg) * (n_ranges + 1);
  std::vector<int32_t> out;
  if (total <= 0) return out;
  Buf<int32_t> d_off(off.size()), d_cuts(static_cast<size_t>(total));
  check(hipMemcpyAsync(d_off.p, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice, s), "offsets");
  LAUNCH(range_cuts_kernel, blocks_for(total), kThreads, 0, s, d_off.p, d_idx, n_seg, std::max(table_rows, 1), n_ranges, d_cuts.p);
  to_host(out, d_cuts.p, static_cast<size_t>(total), s);
  check(hipStreamSynchronize(s), "sync");
  return out;
}

}  // namespace gpu_layout
}  // namespace mmsbm

// sweep_micro.hip -- would a COORDINATED sweep of the gathered tables cut seg_pass's fabric lines?  (round 6)
// Standalone: hipcc -O3 --offload-arch=gfx950 sweep_micro.hip -o sweep_micro && ./sweep_micro
//
// seg_pass at C3 is bound by the fabric's line rate (20.4 us per million 128-byte lines) and insensitive to the rows
// it keeps in flight (B = 1 is within 3 % of B = 4), so what is left is the line COUNT.  Every XCD gathers each row
// 1.25 times per pass on average: with an L2 that held whatever it needs, 43 % of the main lines and 80 % of the 32-byte
// tails would hit (today: ~0-25 % and 60-70 %).  A row's repeats hit only if they come while the line is still there --
// i.e. if all the groups of an XCD work on the same narrow window of the table at the same time.  This emulates it:
//
//   mode 0  today's shape: a group of 8 lanes per segment, 100k + 100k segments of ~10 ascending ids each, one launch
//   mode 1  persistent: 256 x WGS workgroups, every group owns ~4 segments and walks them ONE AFTER THE OTHER
//   mode 2  persistent: every group walks the MERGED ascending list of its segments (all segments of the chip advance
//           through the table together: one sweep per launch), no synchronisation
//   mode 3  mode 2 + a soft barrier per bucket of the table (a counter per bucket; nobody runs more than `slack`
//           buckets ahead of the slowest workgroup)
//
// Two tables (theta-like and A-like: 100k rows of 160 bytes = a 128-byte main line + a 32-byte tail), half of the groups
// gather from each, 1M gathers per table, B rows in flight per group.  Prints us per launch for each mode.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kRows = 100000, kSegs = 100000, kPerSeg = 10, kBlock = 256, kG = 8;

struct Tab { const double *main, *tail; };

template <int B>
__device__ __forceinline__ double walk(const Tab t, const int *__restrict__ idx, int beg, int end, int gl) {
  // lanes 0-3: 4 doubles of the main line each; lane 4: the 4 tail doubles; lanes 5-7 idle (K = 20 in groups of 8)
  const bool main_lane = gl < 4, act = gl < 5;
  double acc = 0.0;
  for (int n = beg; n < end; n += B) {
    int id[B];
    double2 g0[B], g1[B];
#pragma unroll
    for (int b = 0; b < B; ++b) id[b] = idx[min(n + b, end - 1)];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const double *p = main_lane ? t.main + static_cast<size_t>(id[b]) * 16 + gl * 4 : t.tail + static_cast<size_t>(id[b]) * 4;
      if (act) {
        g0[b] = *reinterpret_cast<const double2 *>(p);
        g1[b] = *reinterpret_cast<const double2 *>(p + 2);
      } else {
        g0[b] = double2{0, 0}; g1[b] = double2{0, 0};
      }
    }
#pragma unroll
    for (int b = 0; b < B; ++b)
      if (n + b < end) acc += g0[b].x + g0[b].y + g1[b].x + g1[b].y;
  }
  return acc;
}

// mode 0: one group per segment; blocks [0, nb) table 0, the rest table 1
template <int B>
__global__ __launch_bounds__(kBlock) void per_segment(Tab t0, Tab t1, const int *off0, const int *idx0, const int *off1,
                                                      const int *idx1, double *out, int nseg) {
  const int per = kBlock / kG, nb = (nseg + per - 1) / per;
  const bool first = static_cast<int>(blockIdx.x) < nb;
  const int seg = (first ? blockIdx.x : blockIdx.x - nb) * per + threadIdx.x / kG, gl = threadIdx.x % kG;
  if (seg >= nseg) return;
  const int *off = first ? off0 : off1;
  const double a = walk<B>(first ? t0 : t1, first ? idx0 : idx1, off[seg], off[seg + 1], gl);
  if (gl == 0) out[(first ? 0 : nseg) + seg] = a;
}

// modes 1-3: persistent; group g of the grid owns list g (lists [0, n_lists/2) gather from table 0, the rest from
// table 1; workgroups alternate between the halves so that every XCD sweeps both tables)
template <int B, bool SYNC>
__global__ __launch_bounds__(kBlock) void persistent(Tab t0, Tab t1, const int *off, const int *idx, const int *bucket_cut,
                                                     int n_buckets, int *progress, int slack, double *out, int n_lists) {
  const int per = kBlock / kG;
  const int wg = blockIdx.x, half = gridDim.x / 2;
  const bool first = (wg & 1) == 0;
  const int list = (first ? (wg >> 1) : half + (wg >> 1)) * per + threadIdx.x / kG, gl = threadIdx.x % kG;
  const bool ok = list < n_lists;
  const Tab t = first ? t0 : t1;
  double a = 0.0;
  if (!SYNC) {
    if (ok) a = walk<B>(t, idx, off[list], off[list + 1], gl);
  } else {
    // bucket by bucket: cut[list * (n_buckets + 1) + b] = first position of the list at or beyond bucket b
    const int *cut = bucket_cut + static_cast<size_t>(ok ? list : 0) * (n_buckets + 1);
    for (int b = 0; b < n_buckets; ++b) {
      if (b >= slack) {   // soft barrier: everybody has finished bucket b - slack
        if (threadIdx.x == 0) {
          int spins = 0;
          while (__hip_atomic_load(progress + (b - slack), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < static_cast<int>(gridDim.x) &&
                 ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
      }
      if (ok) a += walk<B>(t, idx, cut[b], cut[b + 1], gl);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(progress + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (ok && gl == 0) out[list] = a;
}

template <class T>
T *upload(const std::vector<T> &v) {
  T *d = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&d), std::max<size_t>(v.size(), 1) * sizeof(T)));
  CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char **argv) {
  const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 6;
  const int n_buckets = argc > 2 ? atoi(argv[2]) : 32;
  const int slack = argc > 3 ? atoi(argv[3]) : 1;
  const bool with_barrier = argc > 4 && atoi(argv[4]) != 0;   // (mode 3 is 20-100x slower than the others: off by default)
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  std::mt19937_64 rng(1);
  // two sets of segments (pair-like gathers theta, user-like gathers A): Poisson(10) lengths, ascending ids
  std::vector<int> off[2], idx[2];
  for (int s = 0; s < 2; ++s) {
    std::poisson_distribution<int> len(kPerSeg);
    std::uniform_int_distribution<int> row(0, kRows - 1);
    off[s].push_back(0);
    for (int g = 0; g < kSegs; ++g) {
      const int n = std::max(1, len(rng));
      std::vector<int> ids(n);
      for (int &v : ids) v = row(rng);
      std::sort(ids.begin(), ids.end());
      idx[s].insert(idx[s].end(), ids.begin(), ids.end());
      off[s].push_back(static_cast<int>(idx[s].size()));
    }
  }
  const long long gathers = static_cast<long long>(idx[0].size() + idx[1].size());
  // persistent lists: `n_lists` groups, half per table; list j of a table = its segments j, j + L, j + 2L, ...
  const int per = kBlock / kG;
  const int grid = cus * wgs_per_cu / 2 * 2, n_lists = grid * per, L = n_lists / 2;
  std::vector<int> p_off_seq{0}, p_idx_seq, p_off_mrg{0}, p_idx_mrg, cuts;
  const int bucket_rows = (kRows + n_buckets - 1) / n_buckets;
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < L; ++j) {
      std::vector<int> merged;
      for (int g = j; g < kSegs; g += L) {
        p_idx_seq.insert(p_idx_seq.end(), idx[s].begin() + off[s][g], idx[s].begin() + off[s][g + 1]);
        merged.insert(merged.end(), idx[s].begin() + off[s][g], idx[s].begin() + off[s][g + 1]);
      }
      p_off_seq.push_back(static_cast<int>(p_idx_seq.size()));
      std::sort(merged.begin(), merged.end());
      const int base = static_cast<int>(p_idx_mrg.size());
      for (int b = 0; b <= n_buckets; ++b)
        cuts.push_back(base + static_cast<int>(std::lower_bound(merged.begin(), merged.end(), b * bucket_rows) - merged.begin()));
      p_idx_mrg.insert(p_idx_mrg.end(), merged.begin(), merged.end());
      p_off_mrg.push_back(static_cast<int>(p_idx_mrg.size()));
    }
  std::vector<double> tab(static_cast<size_t>(kRows) * 16, 1.0), tail(static_cast<size_t>(kRows) * 4, 0.5);
  Tab t0{upload(tab), upload(tail)}, t1{upload(tab), upload(tail)};
  int *d_off0 = upload(off[0]), *d_idx0 = upload(idx[0]), *d_off1 = upload(off[1]), *d_idx1 = upload(idx[1]);
  int *d_poff_seq = upload(p_off_seq), *d_pidx_seq = upload(p_idx_seq), *d_poff_mrg = upload(p_off_mrg), *d_pidx_mrg = upload(p_idx_mrg);
  int *d_cuts = upload(cuts);
  int *d_prog = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&d_prog), sizeof(int) * n_buckets));
  double *d_out = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(double) * std::max(2 * kSegs, n_lists)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%lld gathers of 160-byte rows from two tables of %d rows; persistent grid %d workgroups (%d per CU), %d lists of ~%.1f ids; "
         "%d buckets, slack %d\n", gathers, kRows, grid, wgs_per_cu, n_lists, double(gathers) / n_lists, n_buckets, slack);
  auto time = [&](const char *what, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      for (int j = 0; j < 10; ++j) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms * 100.0f);
    }
    CK(hipGetLastError());
    printf("  %-64s %8.2f us  (%.1f us per million gathers)\n", what, best, best / (gathers * 1e-6));
  };
  const int nb = (kSegs + per - 1) / per;
  time("mode 0: a group per segment, B = 4 (today)", [&] { per_segment<4><<<2 * nb, kBlock>>>(t0, t1, d_off0, d_idx0, d_off1, d_idx1, d_out, kSegs); });
  time("mode 0: a group per segment, B = 1", [&] { per_segment<1><<<2 * nb, kBlock>>>(t0, t1, d_off0, d_idx0, d_off1, d_idx1, d_out, kSegs); });
  time("mode 1: persistent, segments one after the other, B = 4", [&] { persistent<4, false><<<grid, kBlock>>>(t0, t1, d_poff_seq, d_pidx_seq, nullptr, 0, nullptr, 0, d_out, n_lists); });
  time("mode 2: persistent, merged ascending lists, B = 4", [&] { persistent<4, false><<<grid, kBlock>>>(t0, t1, d_poff_mrg, d_pidx_mrg, nullptr, 0, nullptr, 0, d_out, n_lists); });
  time("mode 2: persistent, merged ascending lists, B = 2", [&] { persistent<2, false><<<grid, kBlock>>>(t0, t1, d_poff_mrg, d_pidx_mrg, nullptr, 0, nullptr, 0, d_out, n_lists); });
  time("mode 2: persistent, merged ascending lists, B = 1", [&] { persistent<1, false><<<grid, kBlock>>>(t0, t1, d_poff_mrg, d_pidx_mrg, nullptr, 0, nullptr, 0, d_out, n_lists); });
  if (!with_barrier) return 0;
  auto synced = [&](auto kern) {
    CK(hipMemsetAsync(d_prog, 0, sizeof(int) * n_buckets));
    kern();
  };
  time("mode 3: merged lists + soft barrier per bucket, B = 4", [&] { synced([&] { persistent<4, true><<<grid, kBlock>>>(t0, t1, d_poff_mrg, d_pidx_mrg, d_cuts, n_buckets, d_prog, slack, d_out, n_lists); }); });
  time("mode 3: merged lists + soft barrier per bucket, B = 2", [&] { synced([&] { persistent<2, true><<<grid, kBlock>>>(t0, t1, d_poff_mrg, d_pidx_mrg, d_cuts, n_buckets, d_prog, slack, d_out, n_lists); }); });
  return 0;
}

// once_kernels.hpp -- once-per-run kernels: likelihood, random start, prod_dist / predict / score, compute_omegas
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
#pragma once

namespace {

// ======================================================================================
// once-per-run kernels: likelihood, prod_dist, compute_omegas (element-wise forms that
// follow the reference's association order)
// ======================================================================================
// src/expectation_maximization.py:157-167.  One thread per triple (original order).
__global__ __launch_bounds__(kBlock) void likelihood_kernel(
    const int32_t *__restrict__ tu, const int32_t *__restrict__ ti,
    const int32_t *__restrict__ tr, RowTab theta,
    const double *__restrict__ eta, const double *__restrict__ p, double *__restrict__ block_out,
    int64_t n_obs, int k_groups, int l_groups, int kp, int lp) {
  __shared__ double red[kBlock];
  double total = 0.0;
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; n < n_obs;
       n += static_cast<int64_t>(gridDim.x) * kBlock) {
    const size_t urow = static_cast<size_t>(tu[n]);
    const double *et = eta + static_cast<size_t>(ti[n]) * lp;
    const double *pr = p + static_cast<size_t>(tr[n]) * kp * lp;
    double s = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = *rowtab_ptr(theta, urow, k);
      for (int l = 0; l < l_groups; ++l) s += (tk * et[l]) * pr[k * lp + l];
    }
    const double ls = log(fmax(s, kEps));
    double acc = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = *rowtab_ptr(theta, urow, k);
      for (int l = 0; l < l_groups; ++l) {
        const double w = fmax((tk * et[l]) * pr[k * lp + l], kEps);
        acc += w * log(w) - w * ls;
      }
    }
    total += acc;
  }
  red[threadIdx.x] = total;
  __syncthreads();
  for (int h = kBlock / 2; h > 0; h >>= 1) {
    if (static_cast<int>(threadIdx.x) < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_out[blockIdx.x] = red[0];
}

// Faster form of the same sum for the usual sizes: one thread per triple in PAIR order, a
// workgroup per unit of <= 64 pairs of one rating.  Each thread parks its theta row and its
// pair's eta row in LDS (transposed: conflict-free column reads), the rating tile is
// lane-uniform and comes through scalar loads.  Element-wise formula and association order
// are the reference's; only the order of the outer sum differs.
constexpr int kLikThreads = 128;
static_assert(mmsbm::kMvChunkPairs == kUnitPairs, "likelihood units are built with kMvChunkPairs pairs");

__global__ __launch_bounds__(kLikThreads) void likelihood_units_kernel(
    const mmsbm::Chunk *__restrict__ units, const int32_t *__restrict__ pair_off,
    const int32_t *__restrict__ pair_user, const int32_t *__restrict__ pair_item, RowTab theta,
    const double *__restrict__ eta, const double *__restrict__ p, double *__restrict__ block_out,
    int k_groups, int l_groups, int kp, int lp) {
  extern __shared__ double lds[];
  double *ths = lds;                                          // [kp][kLikThreads]
  double *ets = lds + static_cast<size_t>(kp) * kLikThreads;  // [lp][kLikThreads]
  __shared__ int32_t poff[kUnitPairs + 4];  // (+4: keeps the dynamic LDS base 16-byte aligned)
  __shared__ double red[kLikThreads];
  const mmsbm::Chunk ch = units[blockIdx.x];
  const int tid = threadIdx.x;
  const int npairs = ch.q_end - ch.q_begin;
  if (tid <= npairs) poff[tid] = pair_off[ch.q_begin + tid];
  __syncthreads();
  typedef const double __attribute__((address_space(4))) * const_tile_ptr;
  const const_tile_ptr tile = (const_tile_ptr)(reinterpret_cast<uintptr_t>(
      p + static_cast<size_t>(ch.rating) * kp * lp));
  const int t0 = poff[0], t1 = poff[npairs];
  double total = 0.0;
  for (int base = t0; base < t1; base += kLikThreads) {
    const int n = base + tid;
    const bool have = n < t1;
    int lo = 0, hi = npairs;  // pair of triple n: last q with poff[q] <= n
    const int nn = have ? n : t1 - 1;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (poff[mid] <= nn) lo = mid; else hi = mid;
    }
    const size_t urow = static_cast<size_t>(pair_user[nn]);
    const double *erow = eta + static_cast<size_t>(pair_item[ch.q_begin + lo]) * lp;
    for (int k = 0; k < kp; k += 2) {
      const double2 v = *reinterpret_cast<const double2 *>(rowtab_ptr(theta, urow, k));
      ths[k * kLikThreads + tid] = v.x;
      ths[(k + 1) * kLikThreads + tid] = v.y;
    }
    for (int l = 0; l < lp; l += 2) {
      const double2 v = *reinterpret_cast<const double2 *>(erow + l);
      ets[l * kLikThreads + tid] = v.x;
      ets[(l + 1) * kLikThreads + tid] = v.y;
    }
    // own column only: no workgroup barrier needed between the writes above and the reads below
    double s = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = ths[k * kLikThreads + tid];
      for (int l = 0; l < l_groups; ++l) s += (tk * ets[l * kLikThreads + tid]) * tile[k * lp + l];
    }
    const double ls = log(fmax(s, kEps));
    double acc = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = ths[k * kLikThreads + tid];
      for (int l = 0; l < l_groups; ++l) {
        const double w = fmax((tk * ets[l * kLikThreads + tid]) * tile[k * lp + l], kEps);
        acc += w * log(w) - w * ls;
      }
    }
    if (have) total += acc;
  }
  red[tid] = total;
  __syncthreads();
  for (int h = kLikThreads / 2; h > 0; h >>= 1) {
    if (tid < h) red[tid] += red[tid + h];
    __syncthreads();
  }
  if (tid == 0) block_out[blockIdx.x] = red[0];
}

// The same likelihood without a logarithm per element.  With w = max(omega, eps) and
// ls = log max(s, eps), a triple contributes
//   sum_{omega >= eps} omega (log omega - ls)  +  #{omega < eps} * eps (log eps - ls)
// and log omega = log theta_k + log eta_l + log p_kl comes from tables of logarithms that
// log_table_kernel fills once per evaluation (U*K + I*L + R*K*L logs instead of N*K*L).  One
// pass gathers A = sum w max(log omega, log eps), W = sum w with w = max(omega, eps), and s; ls enters at the
// end: A - ls W.  The element sum is
// re-associated relative to the reference (agreement ~1e-15 relative), the formula is not
// changed.  G lanes share a triple, each holding LW columns of the eta row and of its logarithms
// in registers; the rating's tile (and its logarithms) is lane-uniform for G = 1 (scalar loads)
// or sits in LDS.
__global__ __launch_bounds__(kBlock) void log_table_kernel(const double *__restrict__ in,
                                                           double *__restrict__ out, size_t n) {
  const size_t e = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e < n) out[e] = log(in[e]);  // log(0) = -inf belongs to elements that are clamped, never used
}

// the same for a RowTab (theta): `out` has the one-slot layout whatever `in` has
__global__ __launch_bounds__(kBlock) void log_rows_kernel(RowTab in, RowTab out, size_t rows, int dp) {
  const size_t e = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= rows * dp) return;
  const size_t row = e / dp;
  const int off = static_cast<int>(e - row * dp);
  *rowtab_ptr(out, row, off) = log(*rowtab_ptr(in, row, off));
}

template <int LW, int G, bool TLDS>
__global__ __launch_bounds__(kLikThreads) void likelihood_fast_kernel(
    const mmsbm::Chunk *__restrict__ units, const int32_t *__restrict__ pair_off,
    const int32_t *__restrict__ pair_user, const int32_t *__restrict__ pair_item, RowTab theta,
    RowTab ltheta, RowTab atab, const double *__restrict__ eta, const double *__restrict__ leta,
    const double *__restrict__ p, const double *__restrict__ logp, double *__restrict__ block_out,
    int k_groups, int l_groups, int kp, int lp) {
  extern __shared__ double lds[];  // TLDS: [kp*lp] tile, [kp*lp] its logarithms
  __shared__ int32_t poff[kUnitPairs + 4];
  __shared__ double red[kLikThreads];
  const mmsbm::Chunk ch = units[blockIdx.x];
  const int tid = threadIdx.x;
  const int npairs = ch.q_end - ch.q_begin;
  if (tid <= npairs) poff[tid] = pair_off[ch.q_begin + tid];
  const size_t toff = static_cast<size_t>(ch.rating) * kp * lp;
  if (TLDS) {
    for (int t = tid * 2; t < kp * lp; t += kLikThreads * 2) {
      *reinterpret_cast<double2 *>(lds + t) = *reinterpret_cast<const double2 *>(p + toff + t);
      *reinterpret_cast<double2 *>(lds + kp * lp + t) = *reinterpret_cast<const double2 *>(logp + toff + t);
    }
  }
  __syncthreads();
  typedef const double __attribute__((address_space(4))) * const_tile_ptr;
  const const_tile_ptr gtile = (const_tile_ptr)(reinterpret_cast<uintptr_t>(p + toff));
  const const_tile_ptr gltile = (const_tile_ptr)(reinterpret_cast<uintptr_t>(logp + toff));
  constexpr int TPB = kLikThreads / G;  // triples per round
  const int grp = tid / G, g = tid % G;
  // lane g of a group owns the column PAIRS 2g, 2g + 2G, 2g + 4G, ...: one (16-byte) read instruction of a
  // group then covers 2G consecutive tile entries in LDS.  (Blocks of LW consecutive columns per lane put the
  // lanes 8 LW bytes apart -- a two-way bank conflict on every tile read: SQ_LDS_BANK_CONFLICT was twice
  // SQ_ACTIVE_INST_LDS at C5; single columns g, g + G, ... are conflict-free too but cannot be read as
  // 16-byte pairs: 52 vs 39 ms.)
#define LIK_COL(j) (2 * g + ((j) & 1) + 2 * G * ((j) >> 1))
  const double log_eps = log(kEps);
  const int t0 = poff[0], t1 = poff[npairs];
  double total = 0.0;
  for (int base = t0; base < t1; base += TPB) {
    const int n = base + grp;
    const bool have = n < t1;
    const int nn = have ? n : t1 - 1;
    int lo = 0, hi = npairs;  // pair of triple nn: last q with poff[q] <= nn
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (poff[mid] <= nn) lo = mid; else hi = mid;
    }
    const size_t urow = static_cast<size_t>(pair_user[nn]);
    const size_t irow = static_cast<size_t>(pair_item[ch.q_begin + lo]);
    double e[LW], le[LW];
    int n_fake = 0;  // this lane's columns beyond L: they run on omega = 0, log omega = -inf and are taken off below
#pragma unroll
    for (int j = 0; j < LW; j += 2) {  // (columns past lp: any in-range address)
      const int cc = min(LIK_COL(j), lp - 2);
      const double2 v = *reinterpret_cast<const double2 *>(eta + irow * lp + cc);
      const double2 lv = *reinterpret_cast<const double2 *>(leta + irow * lp + cc);
      const bool r0 = LIK_COL(j) < l_groups, r1 = LIK_COL(j + 1) < l_groups;
      e[j] = r0 ? v.x : 0.0; e[j + 1] = r1 ? v.y : 0.0;
      le[j] = r0 ? lv.x : -INFINITY; le[j + 1] = r1 ? lv.y : -INFINITY;
      n_fake += (r0 ? 0 : 1) + (r1 ? 0 : 1);
    }
    // s_n = theta_n . A[q_n] from the iteration's own table (K multiply-adds), so that log s~ is known BEFORE the
    // elements are visited and every element is subtracted by itself, w (log w - log s~), as the reference does
    // (src/expectation_maximization.py:163-167).  Forming sum w log w and log s~ sum w separately cancels where one
    // element carries the row (one group on a side: 1.6e-10 of the small difference, found by scripts/fuzz_parity.py).
    double s = 0.0;
    {
      const size_t qrow = static_cast<size_t>(ch.q_begin + lo);
      for (int k = 0; k < k_groups; ++k) s = fma(*rowtab_ptr(theta, urow, k), *rowtab_ptr(atab, qrow, k), s);
    }
    const double ls = log(fmax(s, kEps));
    // The clamp as two maxima (round 3): log is monotone, so with w = max(omega, eps) the element is
    // w * (max(log omega, log eps) - ls) -- no compare, no select, no counter of clamped elements.
    double a_sum = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = *rowtab_ptr(theta, urow, k);
      const double ltk = *rowtab_ptr(ltheta, urow, k);
#pragma unroll
      for (int j = 0; j < LW; ++j) {
        const int lc = min(LIK_COL(j), lp - 1);
        double pv, lpv;
        if (TLDS) {
          pv = lds[k * lp + lc];
          lpv = lds[kp * lp + k * lp + lc];
        } else {  // G == 1: l is the same for every lane
          pv = gtile[k * lp + lc];
          lpv = gltile[k * lp + lc];
        }
        const double om = (tk * e[j]) * pv;
        const double w = fmax(om, kEps);
        a_sum = fma(w, fmax((ltk + le[j]) + lpv, log_eps) - ls, a_sum);
      }
    }
    a_sum = group_sum<G>(a_sum);
    const double fake = static_cast<double>(group_sum<G>(static_cast<double>(n_fake * k_groups)));
    if (have && g == 0) total += a_sum - fake * (kEps * (log_eps - ls));
  }
  red[tid] = total;
  __syncthreads();
  for (int h = kLikThreads / 2; h > 0; h >>= 1) {
    if (tid < h) red[tid] += red[tid + h];
    __syncthreads();
  }
  if (tid == 0) block_out[blockIdx.x] = red[0];
}
#undef LIK_COL

// ======================================================================================
// Initial parameters on the device (src/mmsbm.py:224-233): table[row][j] = U / degree(row) with
// U the (offset + row*d + j)-th double of the restart's PCG64 stream -- bit for bit what
// ``default_rng(child_seed).random((rows, d)) / degree`` gives on the host.  Each thread jumps
// the stream to its own 8 consecutive draws.
// ======================================================================================
constexpr int kDrawsPerThread = 8;

__global__ __launch_bounds__(kBlock) void init_rows_kernel(
    RowTab out, const int32_t *__restrict__ off, const int32_t *__restrict__ deg, int rows, int d,
    uint64_t s_hi, uint64_t s_lo, uint64_t i_hi, uint64_t i_lo, uint64_t stream_offset) {
  const uint64_t total = static_cast<uint64_t>(rows) * d;
  const uint64_t f0 = (static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x) * kDrawsPerThread;
  if (f0 >= total) return;
  pcg64::Stream g{pcg64::make128(s_hi, s_lo), pcg64::make128(i_hi, i_lo)};
  pcg64::advance(g, stream_offset + f0);
  int row = static_cast<int>(f0 / d), j = static_cast<int>(f0 % d);
  for (int e = 0; e < kDrawsPerThread && f0 + e < total; ++e) {
    const int cnt = off ? off[row + 1] - off[row] : deg[row];  // rows of this user / item
    *rowtab_ptr(out, static_cast<size_t>(row), j) = pcg64::next_double(g) / static_cast<double>(max(cnt, 1));
    if (++j == d) {
      j = 0;
      ++row;
    }
  }
}

// P[m,r] = sum_kl theta[u,k] eta[i,l] p[k,l,r] for one (row, rating): src/kernels_numpy.py:94-96.
__device__ __forceinline__ double prod_dist_elem(const RowTab &theta, size_t urow,
                                                 const double *__restrict__ et,
                                                 const double *__restrict__ pr, int k_groups,
                                                 int l_groups, int lp) {
  double acc = 0.0;
  for (int k = 0; k < k_groups; ++k) {
    const double tk = *rowtab_ptr(theta, urow, k);
    double inner = 0.0;
    for (int l = 0; l < l_groups; ++l) inner = fma(et[l], pr[k * lp + l], inner);
    acc = fma(tk, inner, acc);
  }
  return acc;
}

// src/kernels_numpy.py:86-96.  One thread per (pair, rating).
__global__ __launch_bounds__(kBlock) void prod_dist_kernel(
    const int32_t *__restrict__ pu, const int32_t *__restrict__ pi,
    RowTab theta, const double *__restrict__ eta,
    const double *__restrict__ p, double *__restrict__ out, int64_t n_pairs, int n_ratings,
    int k_groups, int l_groups, int kp, int lp) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n_pairs * n_ratings) return;
  const int64_t m = e / n_ratings;
  const int r = static_cast<int>(e % n_ratings);
  const size_t urow = static_cast<size_t>(pu[m]);
  const double *et = eta + static_cast<size_t>(pi[m]) * lp;
  const double *pr = p + static_cast<size_t>(r) * kp * lp;
  out[e] = prod_dist_elem(theta, urow, et, pr, k_groups, l_groups, lp);
}

// ======================================================================================
// predict / score on the device (src/mmsbm.py:297-315 and 488-539): the rating distribution
// of every test row for ONE restart is added into a running sum (restart order = call order,
// the order numpy's mean over the restart axis adds in) and reduced on the spot to the
// reference's indicators, so only six numbers per restart travel back:
//   [0] rows kept (distribution not all zero)   [1] argmax == real   [2] |argmax - real| <= 1
//   [3] sum |argmax - real|   [4] real == round(P . w)   [5] sum |P . w - real|
// One thread per test row; fixed-order tree per workgroup, workgroup sums added by the host in
// block order.  FINISH: the distribution is the running sum divided by the number of restarts
// (written back in place), nothing new is computed.
// ======================================================================================
constexpr int kScoreStats = 6;

template <bool FINISH>
__global__ __launch_bounds__(kBlock) void predict_score_kernel(
    const int32_t *__restrict__ pu, const int32_t *__restrict__ pi, const int32_t *__restrict__ preal,
    RowTab theta, const double *__restrict__ eta, const double *__restrict__ p,
    const double *__restrict__ weights, double *__restrict__ sum, double *__restrict__ block_out,
    int64_t n_rows, int n_ratings, int k_groups, int l_groups, int kp, int lp, int first,
    double n_added) {
  __shared__ double red[kScoreStats][kBlock];
  const int64_t m = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  double st[kScoreStats];
#pragma unroll
  for (int j = 0; j < kScoreStats; ++j) st[j] = 0.0;
  if (m < n_rows) {
    const size_t urow = static_cast<size_t>(FINISH ? 0 : pu[m]);
    const double *et = FINISH ? nullptr : eta + static_cast<size_t>(pi[m]) * lp;
    double *srow = sum + m * n_ratings;
    int best = 0;
    double bestv = 0.0, tot = 0.0, pond = 0.0;
    for (int r = 0; r < n_ratings; ++r) {
      double v;
      if (FINISH) {
        v = srow[r] / n_added;
        srow[r] = v;
      } else {
        v = prod_dist_elem(theta, urow, et, p + static_cast<size_t>(r) * kp * lp, k_groups, l_groups, lp);
        srow[r] = first ? v : srow[r] + v;
      }
      if (r == 0 || v > bestv) {  // np.argmax: the first maximum
        bestv = v;
        best = r;
      }
      tot += v;
      pond += v * weights[r];
    }
    if (tot != 0.0) {  // src/mmsbm.py:505-510: rows whose distribution is all zero are dropped
      const int real = preal[m];
      const int dist = abs(best - real);
      st[0] = 1.0;
      st[1] = dist == 0 ? 1.0 : 0.0;
      st[2] = dist <= 1 ? 1.0 : 0.0;
      st[3] = static_cast<double>(dist);
      st[4] = (static_cast<double>(real) == rint(pond)) ? 1.0 : 0.0;  // np.round: half to even
      st[5] = fabs(pond - static_cast<double>(real));
    }
  }
#pragma unroll
  for (int j = 0; j < kScoreStats; ++j) red[j][threadIdx.x] = st[j];
  __syncthreads();
  for (int h = kBlock / 2; h > 0; h >>= 1) {
    if (static_cast<int>(threadIdx.x) < h) {
#pragma unroll
      for (int j = 0; j < kScoreStats; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + h];
    }
    __syncthreads();
  }
  if (threadIdx.x < kScoreStats) block_out[blockIdx.x * kScoreStats + threadIdx.x] = red[threadIdx.x][0];
}

// ======================================================================================
// prod_dist / predict through the factorisation (round 2).  P[m, r] = theta_u . (p_r eta_i): the inner
// vector B[(i, r), :] = p_r eta_i is the A launch's mat-vec over EVERY (item, rating) combination (a
// rating-major pair list q = r I + i built once per context), after which a test row costs R dot
// products of length K instead of R K L multiply-adds behind dependent loads: 1M rows at K = L = 50,
// R = 10 took 357 ms per restart in predict_score_kernel, now the B launch (0.2 ms) plus this kernel.
// A group of G lanes per test row (lane gl owns VEC entries, as in seg_pass); MODE 0 writes the
// distribution (prod_dist), MODE 1 adds it to the session's running sum and reduces the restart's six
// indicator sums exactly as predict_score_kernel does (fixed-order tree over the workgroup's rows).
// ======================================================================================
template <int G, int VEC, int MODE>
__global__ __launch_bounds__(kBlock) void predict_rows_kernel(
    const int32_t *__restrict__ pu, const int32_t *__restrict__ pi, const int32_t *__restrict__ preal,
    RowTab theta, const double *__restrict__ btab, size_t rating_stride, const double *__restrict__ weights,
    double *__restrict__ dist, double *__restrict__ block_out, int64_t n_rows, int n_ratings, int dp,
    int first) {
  constexpr int PER = kBlock / G;
  __shared__ double red[kScoreStats][PER];
  const int gl = threadIdx.x % G, grp = threadIdx.x / G;
  const int64_t m = static_cast<int64_t>(blockIdx.x) * PER + grp;
  double st[kScoreStats];
#pragma unroll
  for (int j = 0; j < kScoreStats; ++j) st[j] = 0.0;
  if (m < n_rows) {  // (whole groups)
    const bool act = gl * VEC < dp;
    const int lane_off = act ? gl * VEC : 0;
    double f[VEC];
    load_vec<VEC>(rowtab_ptr(theta, static_cast<size_t>(pu[m]), lane_off), f);
    if (!act) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) f[v] = 0.0;
    }
    const double *brow = btab + static_cast<size_t>(pi[m]) * dp + lane_off;
    double *srow = dist + m * n_ratings;
    int best = 0;
    double bestv = 0.0, tot = 0.0, pond = 0.0;
    constexpr int RB = 4;  // ratings whose rows are in flight together
    for (int r0 = 0; r0 < n_ratings; r0 += RB) {
      double g[RB][VEC];
#pragma unroll
      for (int b = 0; b < RB; ++b) load_vec<VEC>(brow + static_cast<size_t>(min(r0 + b, n_ratings - 1)) * rating_stride, g[b]);
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const int r = r0 + b;
        if (r < n_ratings) {
          double pt = 0.0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) pt = fma(g[b][v], f[v], pt);
          // (rows of more than G x VEC = 1,024 groups: the remaining columns, 1,024 at a time)
          for (int cb = G * VEC + gl * VEC; cb < dp; cb += G * VEC) {
            double f2[VEC], g2[VEC];
            load_vec<VEC>(rowtab_ptr(theta, static_cast<size_t>(pu[m]), cb), f2);
            load_vec<VEC>(btab + static_cast<size_t>(pi[m]) * dp + static_cast<size_t>(r) * rating_stride + cb, g2);
#pragma unroll
            for (int v = 0; v < VEC; ++v) pt = fma(g2[v], f2[v], pt);
          }
          const double v = group_sum<G>(pt);
          if (MODE == 0) {
            if (gl == 0) srow[r] = v;
          } else {
            if (gl == 0) srow[r] = first ? v : srow[r] + v;
            if (r == 0 || v > bestv) {  // np.argmax: the first maximum
              bestv = v;
              best = r;
            }
            tot += v;
            pond += v * weights[r];
          }
        }
      }
    }
    if (MODE == 1 && gl == 0 && tot != 0.0) {  // src/mmsbm.py:505-510: all-zero rows are dropped
      const int real = preal[m];
      const int dd = abs(best - real);
      st[0] = 1.0;
      st[1] = dd == 0 ? 1.0 : 0.0;
      st[2] = dd <= 1 ? 1.0 : 0.0;
      st[3] = static_cast<double>(dd);
      st[4] = (static_cast<double>(real) == rint(pond)) ? 1.0 : 0.0;  // np.round: half to even
      st[5] = fabs(pond - static_cast<double>(real));
    }
  }
  if (MODE == 0) return;
  if (gl == 0) {
#pragma unroll
    for (int j = 0; j < kScoreStats; ++j) red[j][grp] = st[j];
  }
  __syncthreads();
  for (int h = PER / 2; h > 0; h >>= 1) {
    if (gl == 0 && grp < h) {
#pragma unroll
      for (int j = 0; j < kScoreStats; ++j) red[j][grp] += red[j][grp + h];
    }
    __syncthreads();
  }
  if (threadIdx.x < kScoreStats) block_out[blockIdx.x * kScoreStats + threadIdx.x] = red[threadIdx.x][0];
}

// src/kernels_numpy.py:21-36.  One thread per element; sk/sl = output strides of the
// internal (k, l) indices (they differ from (L, 1) when the sides are swapped).
__global__ __launch_bounds__(kBlock) void omegas_kernel(
    const int32_t *__restrict__ tu, const int32_t *__restrict__ ti,
    const int32_t *__restrict__ tr, RowTab theta,
    const double *__restrict__ eta, const double *__restrict__ p, double *__restrict__ out,
    int64_t n_elems, int k_groups, int l_groups, int kp, int lp, int sk, int sl) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n_elems) return;
  const int kl = k_groups * l_groups;
  const int64_t n = e / kl;
  const int rem = static_cast<int>(e % kl);
  const int k = rem / l_groups, l = rem % l_groups;
  const double v = (*rowtab_ptr(theta, static_cast<size_t>(tu[n]), k) *
                    eta[static_cast<size_t>(ti[n]) * lp + l]) *
                   p[static_cast<size_t>(tr[n]) * kp * lp + k * lp + l];
  out[n * kl + static_cast<int64_t>(k) * sk + static_cast<int64_t>(l) * sl] = v;
}


}  // namespace

// lik_fact.hpp -- the likelihood through the factorisation (round 3): row classification + per-pair tables
// Part of the single translation unit mmsbm_hip.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// ======================================================================================
// src/expectation_maximization.py:157-167 once more:  sum_n sum_kl [ w log w - w log s~ ],  w = max(omega, eps),
// s~ = max(s, eps), omega = theta_k eta_l p_kl.  likelihood_fast_kernel (round 2) visits all K x L elements
// of every triple (2.5e10 element terms at BASELINE's config 5: 32.7 ms).  But the clamp is the only thing
// that does not factorise, and whether it can bite is decided ROW BY ROW of a triple's K x L block from two
// numbers per (item, rating) pair q:  lo_q = min_l eta_l * min_kl p_kl  and  hi_q = max_l eta_l * max_kl p_kl.
//   theta_k * lo_q >= eps : no element of row k is clamped  ("live")   -> the row sums factorise:
//        sum_l omega log omega = theta_k log theta_k A[q,k] + theta_k D[q,k],    sum_l omega = theta_k A[q,k]
//        A[q,k] = sum_l p_kl eta_l          (the EM iteration's own table)
//        D[q,k] = sum_l p_kl (eta_l log eta_l) + sum_l (p_kl log p_kl) eta_l     (two more A-launch mat-vecs)
//   theta_k * hi_q <  eps : every element of row k is clamped ("dead") -> L times eps (log eps - log s~)
//   otherwise ("mixed")   : element by element, in lik_rows_kernel, for that row only.
// s = theta . A is the pair pass's own dot product.  (Where the classification and the reference's
// (theta eta) p differ in the last bit around eps nothing is lost: w and w log w are continuous there.)
// Early in a run every row is live (one gather pass, no element work at all); late in a run memberships are
// concentrated, most rows are dead and the few others cost L elements each instead of K x L per triple.
// Sums are re-associated relative to the reference (agreement ~1e-15 relative); the formula is not changed.
// ======================================================================================

__device__ __forceinline__ double xlogx(double x) { return x > 0.0 ? x * log(x) : 0.0; }

__global__ __launch_bounds__(kBlock) void xlogx_table_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                             size_t n) {
  const size_t e = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e < n) out[e] = xlogx(in[e]);
}

// mm[2 row] = min, mm[2 row + 1] = max over the first `d` entries of each row of a [rows][dp] table
__global__ __launch_bounds__(kBlock) void row_minmax_kernel(const double *__restrict__ tab, double *__restrict__ mm,
                                                            int rows, int d, int dp) {
  const int row = blockIdx.x * kBlock + threadIdx.x;
  if (row >= rows) return;
  const double *p = tab + static_cast<size_t>(row) * dp;
  double lo = p[0], hi = p[0];
  for (int j = 1; j < d; ++j) {
    lo = fmin(lo, p[j]);
    hi = fmax(hi, p[j]);
  }
  mm[2 * row] = lo;
  mm[2 * row + 1] = hi;
}

// one workgroup per rating: min / max over the real K x L entries of its tile [kp][lp]
__global__ __launch_bounds__(kBlock) void tile_minmax_kernel(const double *__restrict__ p, double *__restrict__ mm,
                                                             int k_groups, int l_groups, int kp, int lp) {
  __shared__ double lo_s[kBlock], hi_s[kBlock];
  const double *t = p + static_cast<size_t>(blockIdx.x) * kp * lp;
  double lo = t[0], hi = t[0];
  for (int e = threadIdx.x; e < k_groups * l_groups; e += kBlock) {
    const double v = t[(e / l_groups) * lp + e % l_groups];
    lo = fmin(lo, v);
    hi = fmax(hi, v);
  }
  lo_s[threadIdx.x] = lo;
  hi_s[threadIdx.x] = hi;
  __syncthreads();
  for (int h = kBlock / 2; h > 0; h >>= 1) {
    if (static_cast<int>(threadIdx.x) < h) {
      lo_s[threadIdx.x] = fmin(lo_s[threadIdx.x], lo_s[threadIdx.x + h]);
      hi_s[threadIdx.x] = fmax(hi_s[threadIdx.x], hi_s[threadIdx.x + h]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    mm[2 * blockIdx.x] = lo_s[0];
    mm[2 * blockIdx.x + 1] = hi_s[0];
  }
}

// bounds[2 q] = lo_q, bounds[2 q + 1] = hi_q;  blockIdx.y = rating, pairs rating_off[r] .. rating_off[r + 1]
__global__ __launch_bounds__(kBlock) void pair_bounds_kernel(const int32_t *__restrict__ rating_off,
                                                             const int32_t *__restrict__ pair_item,
                                                             const double *__restrict__ eta_mm,
                                                             const double *__restrict__ p_mm, double *__restrict__ bounds) {
  const int r = blockIdx.y;
  const int q = rating_off[r] + static_cast<int>(blockIdx.x) * kBlock + static_cast<int>(threadIdx.x);
  if (q >= rating_off[r + 1]) return;
  const int it = pair_item[q];
  bounds[2 * static_cast<size_t>(q)] = eta_mm[2 * it] * p_mm[2 * r];
  bounds[2 * static_cast<size_t>(q) + 1] = eta_mm[2 * it + 1] * p_mm[2 * r + 1];
}

struct LikPassArgs {
  RowTab a_tab;                 // A[q, :]  (fixed row of a pair segment)
  const double *d1, *d2;        // [Q][dp] plain tables; D = d1 + d2
  const double *bounds;         // [Q][2]
  RowTab theta;                 // gathered by pair_user
  const int32_t *off, *idx;     // pair_off, pair_user
  const mmsbm::WorkItem *items; // null: unit w is pair w
  int32_t nseg;
  int32_t k_groups, l_groups;
  double *ls_out;               // [N] log max(s_n, eps), in pair order
  unsigned char *flag_out;      // [N] 1: the triple has mixed rows (lik_rows_kernel adds them)
  double *block_out;            // one partial sum per workgroup
};

// The pair pass of seg_pass once more (same segments, same gathers, same group-of-lanes dot product), with the
// likelihood's row sums instead of the C rows.  One partial sum per workgroup (fixed-order tree).
template <int G, int VEC, int B>
__global__ __launch_bounds__(kBlock) void lik_pass_kernel(LikPassArgs a, int dp) {
  __shared__ double red[kBlock / G];
  const int gl = threadIdx.x % G, grp = threadIdx.x / G;
  const int unit = blockIdx.x * (kBlock / G) + grp;
  double total = 0.0;
  if (unit < a.nseg) {  // whole groups
    int seg = unit, beg, end;
    if (a.items) {
      const mmsbm::WorkItem it = a.items[unit];
      seg = it.seg; beg = it.begin; end = it.end;
      if (seg < 0) { beg = 0; end = 0; seg = 0; }  // padding of an XCD-local work list
    } else {
      beg = a.off[unit];
      end = a.off[unit + 1];
    }
    const bool act = gl * VEC < dp;
    const int lane_off = act ? gl * VEC : 0;
    double fa[VEC], fd[VEC];
    load_vec<VEC>(rowtab_ptr(a.a_tab, seg, lane_off), fa);
    {
      double t1[VEC], t2[VEC];
      load_vec<VEC>(a.d1 + static_cast<size_t>(seg) * dp + lane_off, t1);
      load_vec<VEC>(a.d2 + static_cast<size_t>(seg) * dp + lane_off, t2);
#pragma unroll
      for (int v = 0; v < VEC; ++v) fd[v] = t1[v] + t2[v];
    }
    bool real[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) real[v] = act && lane_off + v < a.k_groups;
    const double lo = a.bounds[2 * static_cast<size_t>(seg)], hi = a.bounds[2 * static_cast<size_t>(seg) + 1];
    const double dead_term = static_cast<double>(a.l_groups) * kEps, log_eps = log(kEps);
    const bool g_main = lane_off < a.theta.mw;
    const double *gbase = g_main ? a.theta.main + lane_off : a.theta.tail + (lane_off - a.theta.mw);
    const size_t gstride = g_main ? a.theta.rs_m : a.theta.rs_t;
    constexpr int CH = (G < 16) ? 2 * G : G;
    for (int c0 = beg; c0 < end; c0 += CH) {
      const int cnt = min(CH, end - c0);
      const int mine0 = a.idx[c0 + min(gl, cnt - 1)];
      const int mine1 = (CH > G) ? a.idx[c0 + min(G + gl, cnt - 1)] : 0;
      for (int n = 0; n < cnt; n += B) {
        double g[B][VEC];
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const int jj = min(n + b, cnt - 1);
          const int id = __shfl((CH > G && jj >= G) ? mine1 : mine0, jj, G);
          load_vec<VEC>(gbase + static_cast<size_t>(id) * gstride, g[b]);
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
          if (n + b < cnt) {
            double s = 0.0, x = 0.0, wl = 0.0, nd = 0.0, mx = 0.0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              const double t = g[b][v];
              const double ta = t * fa[v];
              s += real[v] ? ta : 0.0;
              const bool live = real[v] && t * lo >= kEps;
              const bool dead = real[v] && t * hi < kEps;
              x += live ? fma(log(t), ta, t * fd[v]) : 0.0;
              wl += live ? ta : 0.0;
              nd += dead ? 1.0 : 0.0;
              mx += (real[v] && !live && !dead) ? 1.0 : 0.0;
            }
            s = group_sum<G>(s);
            x = group_sum<G>(x);
            wl = group_sum<G>(wl);
            nd = group_sum<G>(nd);
            mx = group_sum<G>(mx);
            const double ls = log(fmax(s, kEps));
            if (gl == 0) {
              total += (x - ls * wl) + nd * (dead_term * (log_eps - ls));
              a.ls_out[c0 + n + b] = ls;
              a.flag_out[c0 + n + b] = mx > 0.0 ? 1 : 0;
            }
          }
        }
      }
    }
  }
  if (gl == 0) red[grp] = total;
  __syncthreads();
  for (int h = (kBlock / G) / 2; h > 0; h >>= 1) {
    if (gl == 0 && grp < h) red[grp] += red[grp + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.block_out[blockIdx.x] = red[0];
}

// The mixed rows, element by element.  Workgroup = a unit of <= 64 pairs of one rating (the units of
// likelihood_fast_kernel); G lanes share a triple, lane g owning the column pairs 2g, 2g + 2G, ... as there.
// Per round of TPB triples: skip if none is flagged; else every flagged triple parks its theta row and the
// logarithms of it in LDS and builds the bit mask of its mixed rows (64 rows at a time), then walks ONLY those
// rows: the tile row and its logarithms from LDS (per-lane rows now, no longer lane-uniform), eta and log eta
// of the lane's columns in registers.
template <int LW, int G>
__global__ __launch_bounds__(kLikThreads) void lik_rows_kernel(
    const mmsbm::Chunk *__restrict__ units, const int32_t *__restrict__ pair_off,
    const int32_t *__restrict__ pair_user, const int32_t *__restrict__ pair_item, RowTab theta,
    RowTab ltheta, const double *__restrict__ eta, const double *__restrict__ leta,
    const double *__restrict__ p, const double *__restrict__ logp, const double *__restrict__ bounds,
    const double *__restrict__ ls_in, const unsigned char *__restrict__ flag_in,
    double *__restrict__ block_out, int k_groups, int l_groups, int kp, int lp) {
  extern __shared__ double lds[];  // [kp*lp] tile, [kp*lp] its logarithms, [TPB][kp] theta rows, [TPB][kp] log theta rows
  __shared__ int32_t poff[kUnitPairs + 4];
  __shared__ double red[kLikThreads];
  __shared__ int any_s;
  constexpr int TPB = kLikThreads / G;
  const mmsbm::Chunk ch = units[blockIdx.x];
  const int tid = threadIdx.x;
  const int npairs = ch.q_end - ch.q_begin;
  if (tid <= npairs) poff[tid] = pair_off[ch.q_begin + tid];
  if (tid == 0) any_s = 0;
  __syncthreads();
  const int t0 = poff[0], t1 = poff[npairs];
  {  // anything to do in this unit at all?
    int any = 0;
    for (int n = t0 + tid; n < t1; n += kLikThreads) any |= flag_in[n];
    if (any) any_s = 1;  // (benign race: every writer stores 1)
  }
  __syncthreads();
  if (!any_s) {  // (uniform)
    if (tid == 0) block_out[blockIdx.x] = 0.0;
    return;
  }
  const size_t toff = static_cast<size_t>(ch.rating) * kp * lp;
  double *tile = lds, *ltile = lds + static_cast<size_t>(kp) * lp;
  double *ths = ltile + static_cast<size_t>(kp) * lp, *lths = ths + static_cast<size_t>(TPB) * kp;
  for (int t = tid * 2; t < kp * lp; t += kLikThreads * 2) {
    *reinterpret_cast<double2 *>(tile + t) = *reinterpret_cast<const double2 *>(p + toff + t);
    *reinterpret_cast<double2 *>(ltile + t) = *reinterpret_cast<const double2 *>(logp + toff + t);
  }
  __syncthreads();
  const int grp = tid / G, g = tid % G;
#define LIK_COL(j) (2 * g + ((j) & 1) + 2 * G * ((j) >> 1))
  const double log_eps = log(kEps);
  double total = 0.0;
  for (int base = t0; base < t1; base += TPB) {
    const int n = base + grp;
    const bool have = n < t1 && flag_in[min(n, t1 - 1)] != 0;
    // (no workgroup barrier inside: a group only reads the LDS rows it wrote itself, and a group never
    // straddles two waves -- G divides 64)
    if (!have) continue;
    int lo_q = 0, hi_q = npairs;  // pair of triple n: last q with poff[q] <= n
    while (hi_q - lo_q > 1) {
      const int mid = (lo_q + hi_q) >> 1;
      if (poff[mid] <= n) lo_q = mid; else hi_q = mid;
    }
    const int q = ch.q_begin + lo_q;
    const size_t urow = static_cast<size_t>(pair_user[n]);
    const size_t irow = static_cast<size_t>(pair_item[q]);
    const double blo = bounds[2 * static_cast<size_t>(q)], bhi = bounds[2 * static_cast<size_t>(q) + 1];
    const double ls = ls_in[n];
    double e[LW], le[LW];
#pragma unroll
    for (int j = 0; j < LW; j += 2) {  // (columns past lp: any in-range address, masked below)
      const int cc = min(LIK_COL(j), lp - 2);
      const double2 v = *reinterpret_cast<const double2 *>(eta + irow * lp + cc);
      const double2 lv = *reinterpret_cast<const double2 *>(leta + irow * lp + cc);
      e[j] = v.x; e[j + 1] = v.y;
      le[j] = lv.x; le[j + 1] = lv.y;
    }
    double a_sum = 0.0, w_sum = 0.0, clamped = 0.0;
    for (int kb = 0; kb < k_groups; kb += 64) {
      // lanes of the group classify the rows kb + g, kb + g + G, ... and park them
      unsigned long long mask = 0ull;
      for (int k = kb + g; k < min(kb + 64, k_groups); k += G) {
        const double tk = *rowtab_ptr(theta, urow, k);
        ths[grp * kp + k] = tk;
        lths[grp * kp + k] = *rowtab_ptr(ltheta, urow, k);
        const bool live = tk * blo >= kEps, dead = tk * bhi < kEps;
        if (!live && !dead) mask |= 1ull << (k - kb);
      }
#pragma unroll
      for (int d = 1; d < G; d <<= 1) {
        const unsigned lo32 = __shfl_xor(static_cast<unsigned>(mask), d, G);
        const unsigned hi32 = __shfl_xor(static_cast<unsigned>(mask >> 32), d, G);
        mask |= (static_cast<unsigned long long>(hi32) << 32) | lo32;
      }
      while (mask) {
        const int k = kb + __builtin_ctzll(mask);
        mask &= mask - 1;
        const double tk = ths[grp * kp + k], ltk = lths[grp * kp + k];
        const double *trow = tile + k * lp, *ltrow = ltile + k * lp;
#pragma unroll
        for (int j = 0; j < LW; j += 2) {
          const int l = LIK_COL(j);
          const int lc = min(l, lp - 2);
          const double2 pv = *reinterpret_cast<const double2 *>(trow + lc);
          const double2 lpv = *reinterpret_cast<const double2 *>(ltrow + lc);
          {
            const bool real = l < l_groups;
            const double w = (tk * e[j]) * pv.x;
            const bool big = real && w >= kEps;
            a_sum += big ? w * ((ltk + le[j]) + lpv.x) : 0.0;
            w_sum += big ? w : 0.0;
            clamped += (real && !big) ? 1.0 : 0.0;
          }
          {
            const bool real = l + 1 < l_groups;
            const double w = (tk * e[j + 1]) * pv.y;
            const bool big = real && w >= kEps;
            a_sum += big ? w * ((ltk + le[j + 1]) + lpv.y) : 0.0;
            w_sum += big ? w : 0.0;
            clamped += (real && !big) ? 1.0 : 0.0;
          }
        }
      }
    }
    a_sum = group_sum<G>(a_sum);
    w_sum = group_sum<G>(w_sum);
    clamped = group_sum<G>(clamped);
    if (g == 0) total += (a_sum - ls * w_sum) + clamped * (kEps * (log_eps - ls));
  }
#undef LIK_COL
  red[tid] = total;
  __syncthreads();
  for (int h = kLikThreads / 2; h > 0; h >>= 1) {
    if (tid < h) red[tid] += red[tid + h];
    __syncthreads();
  }
  if (tid == 0) block_out[blockIdx.x] = red[0];
}

}  // namespace

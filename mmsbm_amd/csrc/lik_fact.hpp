// lik_fact.hpp -- the likelihood pair by pair (round 3): one wave per (item, rating) pair, theta through scalar loads
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
#pragma once

namespace {

// ======================================================================================
// src/expectation_maximization.py:157-167 once more:  sum_n sum_kl [ w log w - w log s~ ],  w = max(omega, eps),
// s~ = max(s, eps), omega = theta_k eta_l p_kl.  likelihood_fast_kernel (round 2) visits the K x L elements of
// every triple on its own: ~14 vector instructions per element, 2.5e10 elements at BASELINE's config 5.
// Three things make an element cheaper here:
//   * the clamp as two maxima.  log is monotone, so log max(omega, eps) = max(log omega, log eps), and an
//     element is  max(omega, eps) * max(log omega - ls, log eps - ls): multiply, max, add, max, fused
//     multiply-add -- no compare, no select, no counter of clamped elements;
//   * pairs.  The triples of one (item, rating) pair share eta_l p_kl and log eta_l + log p_kl; they are
//     neighbours in the pair-sorted order, so a wave takes a PAIR, keeps its eta row (one column per lane)
//     and walks the pair's triples four at a time: per tile row one product and one sum per lane serve four
//     elements;
//   * theta through the scalar unit.  With the whole wave on one pair, theta[u_t, k] and its logarithm are the
//     same for every lane: they arrive as scalar loads (a row's value and logarithm in one s_load_dwordx4 per
//     triple) and cost no vector instruction, no LDS staging and no cross-lane traffic.  s_t = theta_t . A[q] is one coalesced row
//     product + a wave sum.
// log omega = log theta_k + (log eta_l + log p_kl) from the logarithm tables of likelihood_fast_kernel; omega
// is associated as theta_k (eta_l p_kl).  Both differ from the reference's (theta eta) p in the last bit
// only (and w, w log w are continuous at the clamp): agreement ~1e-15 relative, the formula is unchanged.
// Lanes beyond L (and columns beyond L in the last 64-block) run on omega = 0, log omega = -inf: each of their
// elements is exactly eps (log eps - ls), which is taken off again in closed form -- no bounds checks, no divergence.
// ======================================================================================
constexpr int kLikWaveThreads = 512;  // eight waves share a staged tile (43 KB at K = L = 50); 2 workgroups = 16 waves per CU

typedef const double __attribute__((address_space(4))) * const_f64_ptr;
typedef const int32_t __attribute__((address_space(4))) * const_i32_ptr;

// tl[u][k] = (theta[u, k], log theta[u, k]) as one 16-byte pair, plain rows of kp pairs: ONE base address per
// triple for the scalar loads (theta and its logarithm of a row arrive as one s_load_dwordx4), and the vector
// loads of the s_t row product pull a triple's whole row into L2 before the scalar loads ask for it.
__global__ __launch_bounds__(kBlock) void theta_log_pairs_kernel(RowTab in, double2 *__restrict__ tl, size_t rows, int dp) {
  const size_t e = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= rows * dp) return;
  const size_t row = e / dp;
  const double v = *rowtab_ptr(in, row, static_cast<int>(e - row * dp));
  double2 o;
  o.x = v;
  o.y = log(v);  // log(0) = -inf: such an element is clamped, its logarithm is never the larger operand
  tl[e] = o;
}

template <int G>
__device__ __forceinline__ double group_max(double x) {  // like group_sum (common.hpp), with max
  if (G >= 2) x = fmax(x, dpp_move<0xB1>(x));
  if (G >= 4) x = fmax(x, dpp_move<0x4E>(x));
  if (G >= 8) x = fmax(x, dpp_move<0x141>(x));
  if (G >= 16) x = fmax(x, dpp_move<0x140>(x));
  if (G >= 32) x = fmax(x, __shfl_xor(x, 16, 64));
  if (G >= 64) x = fmax(x, __shfl_xor(x, 32, 64));
  return x;
}

// TB triples (nt <= TB of them real) of one pair: s_t, then the elements of each, two tile rows per step.
// Rows that are dead for the whole chunk are not visited: row k is dead when max_t theta[u_t, k] * (max eta_i *
// max_l p[k, l]) < eps -- by monotonicity of rounding every omega of the row is then below eps too, so each of
// its real elements is exactly eps (log eps - ls_t) and goes in as a count.  Early in a run no row is dead;
// late, memberships are concentrated (BASELINE's config 5 after 400 iterations: 83 % of the theta entries are
// below eps) and about half of the rows of a chunk of four triples drop out.  `bk`: this lane's rows' bounds
// max eta_i * max_l p[k, l] (rows lane, lane + 64, ...).
template <int LW, int TB>
__device__ __forceinline__ double lik_wave_chunk(const double2 *__restrict__ tl, const size_t (&u)[4], int nt, int lane,
                                                 const double (&av)[3], const double (&bk)[3], int na, const double *tile,
                                                 const double *ltile, const double (&e)[LW], const double (&le)[LW],
                                                 const int (&cc)[LW], int k_groups, int kp, int lp, int n_inv, double log_eps) {
  double ls[TB], cl[TB], acc[TB];
  double sp[TB], vmax[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) vmax[m] = 0.0;
#pragma unroll
  for (int t = 0; t < TB; ++t) sp[t] = 0.0;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    if (m < na) {
#pragma unroll
      for (int t = 0; t < TB; ++t) {  // s_t = theta_t . A[q]: a coalesced row product + a wave sum
        const double th = tl[u[t] * kp + min(lane + 64 * m, kp - 1)].x;
        sp[t] = fma(th, av[m], sp[t]);
        vmax[m] = fmax(vmax[m], th);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TB; ++t) {
    const double s = group_sum<64>(sp[t]);
    ls[t] = log(fmax(s, kEps));
    cl[t] = log_eps - ls[t];
    acc[t] = 0.0;
  }
  int n_proc = 0;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    if (m >= na) break;
    // rows 64 m .. 64 m + 63 that are not dead (rows >= K hold theta = 0: never live)
    unsigned long long rows = __ballot(lane + 64 * m < kp && vmax[m] * bk[m] >= kEps);
    n_proc += __builtin_popcountll(rows);
    while (rows) {  // (uniform)
      const int ka = 64 * m + __builtin_ctzll(rows);
      rows &= rows - 1;
      const bool two = rows != 0;
      const int kb = two ? 64 * m + __builtin_ctzll(rows) : ka;
      rows &= rows - 1;  // (0 stays 0)
      double tq[TB][4];  // theta, log theta of rows ka, kb: scalar registers
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const const_f64_ptr pa = (const_f64_ptr)(reinterpret_cast<uintptr_t>(tl + u[t] * kp + ka));
        const const_f64_ptr pb = (const_f64_ptr)(reinterpret_cast<uintptr_t>(tl + u[t] * kp + kb));
        tq[t][0] = pa[0]; tq[t][1] = pa[1];
        tq[t][2] = pb[0]; tq[t][3] = pb[1];
      }
      double pv[2][LW], lpv[2][LW];
#pragma unroll
      for (int j = 0; j < LW; ++j) {
        pv[0][j] = tile[ka * lp + cc[j]];
        lpv[0][j] = ltile[ka * lp + cc[j]];
        pv[1][j] = tile[kb * lp + cc[j]];
        lpv[1][j] = ltile[kb * lp + cc[j]];
      }
#pragma unroll
      for (int j = 0; j < LW; ++j) {
        const double ep = e[j] * pv[0][j], lep = le[j] + lpv[0][j];
#pragma unroll
        for (int t = 0; t < TB; ++t)
          acc[t] = fma(fmax(tq[t][0] * ep, kEps), fmax((tq[t][1] - ls[t]) + lep, cl[t]), acc[t]);
      }
      if (two) {
#pragma unroll
        for (int j = 0; j < LW; ++j) {
          const double ep = e[j] * pv[1][j], lep = le[j] + lpv[1][j];
#pragma unroll
          for (int t = 0; t < TB; ++t)
            acc[t] = fma(fmax(tq[t][2] * ep, kEps), fmax((tq[t][3] - ls[t]) + lep, cl[t]), acc[t]);
        }
      }
    }
  }
  // this lane's columns: LW - n_inv real ones; a visited row's other columns each added exactly eps (log eps - ls)
  // -- eta = 0, log eta = -inf -- and come off again, a dead row's real columns each go in with the same amount
  const double count = static_cast<double>((k_groups - n_proc) * (LW - n_inv) - n_proc * n_inv);
  double tot = 0.0;
#pragma unroll
  for (int t = 0; t < TB; ++t)
    if (t < nt) tot += acc[t] + count * (kEps * cl[t]);
  return tot;
}

#ifndef MMSBM_LIK_WPE
#define MMSBM_LIK_WPE 4  // waves per SIMD the kernel is compiled for
#endif
template <int LW>  // columns per lane: lane, lane + 64, ...
// (three columns per lane -- rows of 129 to 192 groups -- hold 12 more doubles per lane than two: compiled for one wave
// per SIMD fewer, it keeps them in registers instead of 28 bytes of scratch)
__global__ __launch_bounds__(kLikWaveThreads) __attribute__((amdgpu_waves_per_eu(LW == 3 ? MMSBM_LIK_WPE - 1 : MMSBM_LIK_WPE, 8))) void lik_wave_kernel(
    const mmsbm::Chunk *__restrict__ units, const int32_t *__restrict__ pair_off,
    const int32_t *__restrict__ pair_user, const int32_t *__restrict__ pair_item, const double2 *__restrict__ tl,
    RowTab a_tab, const double *__restrict__ eta, const double *__restrict__ leta, const double *__restrict__ p,
    const double *__restrict__ logp, double *__restrict__ block_out, int k_groups, int l_groups, int kp, int lp) {
  constexpr int NW = kLikWaveThreads / 64;
  extern __shared__ double lds[];  // [kp*lp] tile, [kp*lp] its logarithms, [kp] the tile's row maxima
  __shared__ int32_t poff[kUnitPairs + 4];
  __shared__ double red[NW];
  const mmsbm::Chunk ch = units[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = ch.q_end - ch.q_begin;
  if (tid <= npairs) poff[tid] = pair_off[ch.q_begin + tid];
  double *tile = lds, *ltile = lds + static_cast<size_t>(kp) * lp;
  {
    const size_t toff = static_cast<size_t>(ch.rating) * kp * lp;
    for (int t = tid * 2; t < kp * lp; t += kLikWaveThreads * 2) {
      *reinterpret_cast<double2 *>(tile + t) = *reinterpret_cast<const double2 *>(p + toff + t);
      *reinterpret_cast<double2 *>(ltile + t) = *reinterpret_cast<const double2 *>(logp + toff + t);
    }
  }
  __syncthreads();
  double *prmax = ltile + static_cast<size_t>(kp) * lp;
  for (int k = tid; k < kp; k += kLikWaveThreads) {
    double mx = 0.0;
    for (int l = 0; l < l_groups; ++l) mx = fmax(mx, tile[k * lp + l]);
    prmax[k] = mx;
  }
  __syncthreads();
  const const_i32_ptr users = (const_i32_ptr)(reinterpret_cast<uintptr_t>(pair_user));
  const double log_eps = log(kEps);
  int cc[LW], n_inv = 0;
#pragma unroll
  for (int j = 0; j < LW; ++j) {
    const int col = lane + 64 * j;
    cc[j] = min(col, lp - 1);
    n_inv += col >= l_groups ? 1 : 0;
  }
  const int na = (kp + 63) / 64;  // (<= 3: the host sends wider rows elsewhere)
  double total = 0.0;
  for (int pq = wave; pq < npairs; pq += NW) {
    const int q = ch.q_begin + pq;
    const int t0 = __builtin_amdgcn_readfirstlane(poff[pq]), t1 = __builtin_amdgcn_readfirstlane(poff[pq + 1]);
    const size_t irow = static_cast<size_t>(__builtin_amdgcn_readfirstlane(pair_item[q]));
    double e[LW], le[LW], av[3], bk[3], emax = 0.0;
#pragma unroll
    for (int j = 0; j < LW; ++j) {
      const bool ok = lane + 64 * j < l_groups;
      const double ev = eta[irow * lp + cc[j]], lv = leta[irow * lp + cc[j]];
      e[j] = ok ? ev : 0.0;
      le[j] = ok ? lv : -INFINITY;
      emax = fmax(emax, e[j]);
    }
    emax = group_max<64>(emax);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int kk = lane + 64 * m;
      const bool on = m < na && kk < kp;
      av[m] = on ? *rowtab_ptr(a_tab, static_cast<size_t>(q), kk) : 0.0;
      bk[m] = on ? emax * prmax[kk] : 0.0;
    }
    for (int c0 = t0; c0 < t1; c0 += 4) {
      const int nt = min(4, t1 - c0);
      size_t u[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) u[t] = static_cast<size_t>(users[c0 + min(t, nt - 1)]);
      // (a pair's last one or two triples run the narrower instantiations: no work for slots that hold nothing)
      if (nt > 2) total += lik_wave_chunk<LW, 4>(tl, u, nt, lane, av, bk, na, tile, ltile, e, le, cc, k_groups, kp, lp, n_inv, log_eps);
      else if (nt == 2) total += lik_wave_chunk<LW, 2>(tl, u, nt, lane, av, bk, na, tile, ltile, e, le, cc, k_groups, kp, lp, n_inv, log_eps);
      else total += lik_wave_chunk<LW, 1>(tl, u, nt, lane, av, bk, na, tile, ltile, e, le, cc, k_groups, kp, lp, n_inv, log_eps);
    }
  }
  total = group_sum<64>(total);
  if (lane == 0) red[wave] = total;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w];
    block_out[blockIdx.x] = tot;
  }
}

// ======================================================================================
// Rows of up to 32 groups (round 4; BASELINE's configs 1-4: K = L = 10, 20).  A wave per pair would leave most of its
// lanes idle there (one column per lane), so here a LANE takes a triple -- every lane busy whatever L is -- and the
// rating's tile comes through the scalar unit instead of theta:
//   * the triple's theta row and its logarithms sit in registers (KP 16-byte loads in ONE burst from the table of
//     (theta, log theta) pairs; likelihood_fast_kernel fetched them inside its row loop, two dependent loads per row:
//     SQ_WAIT_ANY 66 % of the wave cycles at C3);
//   * s_n = theta_n . A[q_n] from the A table of the iteration (K multiply-adds instead of K L additions), so that
//     ls = log max(s, eps) is known BEFORE the elements are visited and the sum of the clamped omegas is not needed:
//     an element is  max(omega, eps) * max(log omega - ls, log eps - ls);
//   * the workgroup's triples share one rating, so (p_kl, log p_kl) is the same for every lane: scalar loads of
//     contiguous (value, logarithm) pairs of the TRANSPOSED tile, operands of the vector instructions as SGPRs;
//   * the item's (eta_l, log eta_l) pair is one 16-byte load per column, asked for a column ahead.
// Seven vector instructions per element (multiply, multiply, max, add, add, max, fused multiply-add) against nine and
// two LDS / scalar round trips.  Padded rows (theta = 0, log theta = -inf) each add exactly eps (log eps - ls) per
// column and are taken off in closed form.
// ======================================================================================
template <int KP, int NT>
__global__ __launch_bounds__(NT) void lik_lane_kernel(
    const mmsbm::Chunk *__restrict__ units, const int32_t *__restrict__ pair_off,
    const int32_t *__restrict__ pair_user, const int32_t *__restrict__ pair_item, const double2 *__restrict__ tl,
    RowTab a_tab, const double2 *__restrict__ el, const double2 *__restrict__ ptl, double *__restrict__ block_out,
    int k_groups, int l_groups, int lp) {
  __shared__ int32_t poff[kUnitPairs + 4];
  __shared__ double red[NT / 64];
  const mmsbm::Chunk ch = units[blockIdx.x];
  const int tid = threadIdx.x;
  const int npairs = ch.q_end - ch.q_begin;
  if (tid <= npairs) poff[tid] = pair_off[ch.q_begin + tid];
  __syncthreads();
  // (value, logarithm) pairs of the transposed tile of this rating: [l][k][2] doubles, uniform addresses
  const const_f64_ptr tile = (const_f64_ptr)(reinterpret_cast<uintptr_t>(ptl + static_cast<size_t>(ch.rating) * lp * KP));
  const double log_eps = log(kEps);
  const int t0 = poff[0], t1 = poff[npairs];
  double total = 0.0;
  for (int base = t0; base < t1; base += NT) {
    const int n = base + tid;
    const bool have = n < t1;
    const int nn = have ? n : t1 - 1;
    int lo = 0, hi = npairs;  // pair of triple nn: last q with poff[q] <= nn
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (poff[mid] <= nn) lo = mid; else hi = mid;
    }
    const size_t urow = static_cast<size_t>(pair_user[nn]);
    const size_t q = static_cast<size_t>(ch.q_begin + lo);
    const double2 *__restrict__ erow = el + static_cast<size_t>(pair_item[q]) * lp;
    double2 th[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) th[k] = tl[urow * KP + k];
    double2 ev = erow[0];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < KP; k += 2) {
      const double2 av = *reinterpret_cast<const double2 *>(rowtab_ptr(a_tab, q, k));
      s = fma(th[k].x, av.x, s);
      s = fma(th[k + 1].x, av.y, s);
    }
    const double ls = log(fmax(s, kEps)), cl = log_eps - ls;
#pragma unroll
    for (int k = 0; k < KP; ++k) th[k].y -= ls;
    double acc0 = 0.0, acc1 = 0.0;
    // The tile row of column l is KP (value, logarithm) pairs = KP / 4 scalar loads of four pairs; rows follow each
    // other in memory, so the chunk after the current one is always 8 doubles further on.  It is asked for BEFORE the
    // current chunk is multiplied (two sets of scalar registers): with one set the compiler put every load right in
    // front of its use and every chunk waited out a scalar-cache round trip (5 per column at K = 20: more stall than work).
    double tc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tc[j] = tile[j];
    for (int l = 0; l < l_groups; ++l) {
      const double2 nxt = erow[min(l + 1, l_groups - 1)];
      const const_f64_ptr tr = tile + static_cast<size_t>(l) * KP * 2;
#pragma unroll
      for (int c = 0; c < KP / 4; ++c) {
        double tn[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tn[j] = tr[8 * (c + 1) + j];   // (the last chunk of the last row reads the 64 bytes behind the tile: allocated)
        __builtin_amdgcn_sched_barrier(0);
        const int k = 4 * c;
        const double ep0 = ev.x * tc[0], ep1 = ev.x * tc[2], ep2 = ev.x * tc[4], ep3 = ev.x * tc[6];
        const double lw0 = (th[k].y + ev.y) + tc[1], lw1 = (th[k + 1].y + ev.y) + tc[3];
        const double lw2 = (th[k + 2].y + ev.y) + tc[5], lw3 = (th[k + 3].y + ev.y) + tc[7];
        acc0 = fma(fmax(th[k].x * ep0, kEps), fmax(lw0, cl), acc0);
        acc1 = fma(fmax(th[k + 1].x * ep1, kEps), fmax(lw1, cl), acc1);
        acc0 = fma(fmax(th[k + 2].x * ep2, kEps), fmax(lw2, cl), acc0);
        acc1 = fma(fmax(th[k + 3].x * ep3, kEps), fmax(lw3, cl), acc1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) tc[j] = tn[j];
      }
      ev = nxt;
    }
    if (have) total += (acc0 + acc1) - static_cast<double>((KP - k_groups) * l_groups) * (kEps * cl);
  }
  total = group_sum<64>(total);
  if ((tid & 63) == 0) red[tid >> 6] = total;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tot += red[w];
    block_out[blockIdx.x] = tot;
  }
}

}  // namespace

// fused_small.hpp -- small problems: the EM iteration in TWO launches instead of four (round 3)
// Part of the single translation unit mmsbm_hip.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// ======================================================================================
// Below a few hundred thousand ratings the four launches of an iteration are four launch floors plus one
// chain of dependent memory round trips each (BASELINE's config 1, 100k ratings: 9.6 + 7.2 + 5.3 + 5.3 us,
// 7 % of its roofline).  What keeps them apart is only WHERE a table is produced and consumed, and two of the
// three hand-overs are local to a 64-pair unit of one rating:
//
//   launch 1  pairs_fused_kernel, workgroup = unit of <= 64 pairs of one rating
//       A[q,:]  = pT_r eta[i_q,:]              (the A launch's mat-vec; kept in LDS, also written out for launch 2)
//       C[q,:]  = sum_{n in pair q} theta[u_n,:] / max(theta[u_n,:] . A[q,:], eps)      (the pair pass; C stays in LDS)
//       T[q,:]  = p_r^T C[q,:]   and the unit's K x L slab  S += C^T eta              (the T + S launch)
//   launch 2  tail_fused_kernel, three roles in one grid
//       user segments: theta' (gathers A of launch 1)  ||  p_update: p', pT' from the slabs  ||  item_sum: eta' from T
//
// A is computed at the START of an iteration from the current parameters instead of at the end of the previous
// one (same values: the context's `a_ok` says whether atab[cur] matches the parameters, ensure_a() refreshes it
// for consumers outside the iteration).  C never goes to memory.  Arithmetic and association order of every
// output are those of the four-launch form: results are bitwise identical (tested).
// ======================================================================================
struct FusedPairArgs {
  const double *pt_tiles;  // pT [R][lp][kp]  (A mat-vec)
  const double *p_tiles;   // p  [R][kp][lp]  (T mat-vec)
  const double *eta;       // [I][lp]
  RowTab theta;            // gathered by pair_user
  RowTab a_out;            // A table
  const int32_t *pair_off, *pair_user, *pair_item;
  const mmsbm::Chunk *chunks;
  double *t_out;           // T [Q][lp]
  double *partial;         // one K x L slab per workgroup
  int kp, lp, spb, nsub;
  size_t bs_tiles, bs_eta, bs_t, bs_partial;  // restart slots (blockIdx.y): strides of the streamed tables
};

size_t pairs_fused_lds(int kp, int lp) {
  return (static_cast<size_t>(lp) * (kUnitPairs + 1) + static_cast<size_t>(kUnitPairs) * lp +
          static_cast<size_t>(kUnitPairs) * kp + static_cast<size_t>(kp) * (kUnitPairs + 1)) * sizeof(double);
}

template <int G, int VEC>
__global__ __launch_bounds__(kBlock) void pairs_fused_kernel(FusedPairArgs fa) {
  constexpr int CS = kUnitPairs + 1, KT = 2, TV = KT * 4, B = 8;
  const size_t slot = blockIdx.y;
  const double *__restrict__ pt_tiles = fa.pt_tiles + slot * fa.bs_tiles;
  const double *__restrict__ p_tiles = fa.p_tiles + slot * fa.bs_tiles;
  const double *__restrict__ eta = fa.eta + slot * fa.bs_eta;
  const RowTab theta = slot_tab(fa.theta, slot), a_out = slot_tab(fa.a_out, slot);
  double *__restrict__ t_out = fa.t_out + slot * fa.bs_t;
  double *__restrict__ partial = fa.partial + slot * fa.bs_partial;
  const int kp = fa.kp, lp = fa.lp;
  extern __shared__ double lds[];
  double *etaT = lds;                                       // [lp][CS]  eta rows of the unit, transposed (A mat-vec input)
  double *es = etaT + static_cast<size_t>(lp) * CS;         // [64][lp]  the same rows, row-major (S); later the T rows
  double *aout = es + static_cast<size_t>(kUnitPairs) * lp; // [64][kp]  A rows of the unit
  double *cT = aout + static_cast<size_t>(kUnitPairs) * kp; // [kp][CS]  C rows, transposed
  const mmsbm::Chunk ch = fa.chunks[blockIdx.x];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q0 = ch.q_begin, np = ch.q_end - ch.q_begin;  // (np <= 64; 0 for the padding of the unit lists)
  typedef const double __attribute__((address_space(4))) * const_tile_ptr;
  const const_tile_ptr gpt = (const_tile_ptr)(reinterpret_cast<uintptr_t>(pt_tiles + static_cast<size_t>(ch.rating) * lp * kp));
  const const_tile_ptr gp = (const_tile_ptr)(reinterpret_cast<uintptr_t>(p_tiles + static_cast<size_t>(ch.rating) * kp * lp));

  // ---- the pair segments' first indices and offsets: asked for now, used after the A mat-vec ----
  constexpr int NGRP = kBlock / G, ROUNDS = (kUnitPairs + NGRP - 1) / NGRP;
  const int grp = tid / G, gl = tid % G;
  constexpr int CH = (G < 16) ? 2 * G : G;
  int beg[ROUNDS], end[ROUNDS], mine0[ROUNDS], mine1[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int pr = grp + r * NGRP;
    const bool on = pr < np;
    beg[r] = on ? fa.pair_off[q0 + pr] : 0;
    end[r] = on ? fa.pair_off[q0 + pr + 1] : 0;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int cnt = min(CH, end[r] - beg[r]);
    mine0[r] = cnt > 0 ? fa.pair_user[beg[r] + min(gl, cnt - 1)] : 0;
    mine1[r] = (CH > G && cnt > 0) ? fa.pair_user[beg[r] + min(G + gl, cnt - 1)] : 0;
  }

  // ---- eta rows of the unit's pairs -> etaT (transposed) and es (row-major) ----
  {
    const int tot = np * lp;
    for (int t0 = tid * 2; t0 < tot; t0 += kBlock * 4) {
      double2 v[2];
      int pr[2], d[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = min(t0 + j * kBlock * 2, max(tot - 2, 0));
        pr[j] = t / lp;
        d[j] = t - pr[j] * lp;
        v[j] = *reinterpret_cast<const double2 *>(eta + static_cast<size_t>(fa.pair_item[q0 + pr[j]]) * lp + d[j]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (t0 + j * kBlock * 2 < tot) {
          etaT[d[j] * CS + pr[j]] = v[j].x;
          etaT[(d[j] + 1) * CS + pr[j]] = v[j].y;
          *reinterpret_cast<double2 *>(es + pr[j] * lp + d[j]) = v[j];
        }
      }
    }
    if (np < kUnitPairs) {  // ragged tail of a rating: the missing pairs are zero columns / zero rows
      for (int t = tid; t < (kUnitPairs - np) * lp; t += kBlock) {
        etaT[(t / (kUnitPairs - np)) * CS + np + t % (kUnitPairs - np)] = 0.0;
        es[np * lp + t] = 0.0;
      }
    }
  }
  __syncthreads();
  // ---- A[q,:] = pT_r eta_q: lane = pair, wave = chunk of 4 outputs (pair_block's A mode) ----
  for (int c = __builtin_amdgcn_readfirstlane(wave); c < (kp >> 2); c += kBlock / 64) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int d = 0; d < lp; d += 4) {
      double x[4];
      double2 m0[4], m1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i] = etaT[(d + i) * CS + lane];
        const const_tile_ptr row = gpt + static_cast<size_t>(d + i) * kp + c * 4;  // uniform: scalar loads
        m0[i].x = row[0]; m0[i].y = row[1];
        m1[i].x = row[2]; m1[i].y = row[3];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0 = fma(x[i], m0[i].x, a0);
        a1 = fma(x[i], m0[i].y, a1);
        a2 = fma(x[i], m1[i].x, a2);
        a3 = fma(x[i], m1[i].y, a3);
      }
    }
    double2 w0, w1;
    w0.x = a0; w0.y = a1; w1.x = a2; w1.y = a3;
    *reinterpret_cast<double2 *>(aout + lane * kp + c * 4) = w0;
    *reinterpret_cast<double2 *>(aout + lane * kp + c * 4 + 2) = w1;
  }
  __syncthreads();
  // the unit's A rows go to memory for launch 2 (the user pass gathers them)
  for (int t = tid * 2; t < np * kp; t += kBlock * 2) {
    const int pr = t / kp, j = t - pr * kp;
    *reinterpret_cast<double2 *>(rowtab_ptr(a_out, static_cast<size_t>(q0 + pr), j)) =
        *reinterpret_cast<const double2 *>(aout + t);
  }
  // ---- the pair segments (seg_body's arithmetic): a group of G lanes per pair, C row into cT ----
  {
    const bool act = gl * VEC < kp;
    const int lane_off = act ? gl * VEC : 0;
    const bool g_main = lane_off < theta.mw;
    const double *gbase = g_main ? theta.main + lane_off : theta.tail + (lane_off - theta.mw);
    const size_t gstride = g_main ? theta.rs_m : theta.rs_t;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int pr = grp + r * NGRP;
      if (pr >= kUnitPairs) continue;  // (whole groups)
      double f[VEC], acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc[v] = 0.0;
        f[v] = act ? aout[pr * kp + lane_off + v] : 0.0;
      }
      for (int c0 = beg[r]; c0 < end[r]; c0 += CH) {
        const int cnt = min(CH, end[r] - c0);
        int m0 = mine0[r], m1 = mine1[r];
        if (c0 != beg[r]) {  // (segments longer than the first batch of indices: rare here)
          m0 = fa.pair_user[c0 + min(gl, cnt - 1)];
          m1 = (CH > G) ? fa.pair_user[c0 + min(G + gl, cnt - 1)] : 0;
        }
        for (int n = 0; n < cnt; n += B) {
          double g[B][VEC];
#pragma unroll
          for (int b = 0; b < B; ++b) {
            const int jj = min(n + b, cnt - 1);
            const int id = __shfl((CH > G && jj >= G) ? m1 : m0, jj, G);
            load_vec<VEC>(gbase + static_cast<size_t>(id) * gstride, g[b]);
          }
#pragma unroll
          for (int b = 0; b < B; ++b) {
            if (n + b < cnt) {
              double pt = 0.0;
#pragma unroll
              for (int v = 0; v < VEC; ++v) pt = fma(g[b][v], f[v], pt);
              const double s = group_sum<G>(pt);
              const double w = 1.0 / fmax(s, kEps);
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[v] = fma(g[b][v], w, acc[v]);
            }
          }
        }
      }
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) cT[(lane_off + v) * CS + pr] = acc[v];  // (pairs >= np: zero columns)
      }
    }
  }
  __syncthreads();
  // ---- S: thread = (k, 4 l) slot of a KT x 4 register tile, copies split the unit's pairs (pair_block) ----
  const int nch = lp >> 2;
  const int nslot = (kp / KT) * nch, spb = fa.spb, nsub = fa.nsub;
  const int sub = tid / spb, slot0 = tid % spb;
  const bool s_active = sub < nsub;
  const int o = min(slot0, nslot - 1);
  const int coff = (o / nch) * KT * CS, eoff = (o % nch) * 4;
  double sacc[TV];
#pragma unroll
  for (int j = 0; j < TV; ++j) sacc[j] = 0.0;
  if (s_active) {
#pragma unroll 2
    for (int j = sub; j < np; j += nsub) {
      double cv[KT];
#pragma unroll
      for (int i = 0; i < KT; ++i) cv[i] = cT[coff + i * CS + j];
      const double2 e0 = *reinterpret_cast<const double2 *>(es + j * lp + eoff);
      const double2 e1 = *reinterpret_cast<const double2 *>(es + j * lp + eoff + 2);
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        sacc[4 * i + 0] = fma(cv[i], e0.x, sacc[4 * i + 0]);
        sacc[4 * i + 1] = fma(cv[i], e0.y, sacc[4 * i + 1]);
        sacc[4 * i + 2] = fma(cv[i], e1.x, sacc[4 * i + 2]);
        sacc[4 * i + 3] = fma(cv[i], e1.y, sacc[4 * i + 3]);
      }
    }
  }
  __syncthreads();  // es is dead: its space takes the T rows
  double *tout = es;
  // ---- T[q,:] = p_r^T C_q: lane = pair, wave = chunk of 4 outputs ----
  for (int c = __builtin_amdgcn_readfirstlane(wave); c < nch; c += kBlock / 64) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int d = 0; d < kp; d += 4) {
      double x[4];
      double2 m0[4], m1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i] = cT[(d + i) * CS + lane];
        const const_tile_ptr row = gp + static_cast<size_t>(d + i) * lp + c * 4;
        m0[i].x = row[0]; m0[i].y = row[1];
        m1[i].x = row[2]; m1[i].y = row[3];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0 = fma(x[i], m0[i].x, a0);
        a1 = fma(x[i], m0[i].y, a1);
        a2 = fma(x[i], m1[i].x, a2);
        a3 = fma(x[i], m1[i].y, a3);
      }
    }
    double2 w0, w1;
    w0.x = a0; w0.y = a1; w1.x = a2; w1.y = a3;
    *reinterpret_cast<double2 *>(tout + lane * lp + c * 4) = w0;
    *reinterpret_cast<double2 *>(tout + lane * lp + c * 4 + 2) = w1;
  }
  __syncthreads();
  {  // the unit's T rows are contiguous in memory
    double *dst = t_out + static_cast<size_t>(q0) * lp;
    for (int t = tid * 2; t < np * lp; t += kBlock * 2)
      *reinterpret_cast<double2 *>(dst + t) = *reinterpret_cast<const double2 *>(tout + t);
  }
  // ---- the slab: the copies hand their sums over through LDS, added in copy order ----
  if (nsub > 1) {
    __syncthreads();
    if (s_active && sub > 0 && slot0 < nslot) {
#pragma unroll
      for (int j = 0; j < TV; ++j) lds[(j * (nsub - 1) + sub - 1) * nslot + slot0] = sacc[j];
    }
    __syncthreads();
    if (sub == 0 && slot0 < nslot) {
      for (int oo = 1; oo < nsub; ++oo)
#pragma unroll
        for (int j = 0; j < TV; ++j) sacc[j] += lds[(j * (nsub - 1) + oo - 1) * nslot + slot0];
    }
  }
  if (sub == 0 && slot0 < nslot) {
    double *cell = partial + static_cast<size_t>(blockIdx.x) * kp * lp + (o / nch) * KT * lp + eoff;
#pragma unroll
    for (int h = 0; h < KT; ++h) {
      double2 x, y;
      x.x = sacc[4 * h]; x.y = sacc[4 * h + 1]; y.x = sacc[4 * h + 2]; y.y = sacc[4 * h + 3];
      *reinterpret_cast<double2 *>(cell + h * lp) = x;
      *reinterpret_cast<double2 *>(cell + h * lp + 2) = y;
    }
  }
}

// launch 2: blocks [0, bu) user segments, [bu, bu + nb_p) p_update, the rest item_sum
template <int G, int VEC, int GL, int VECL>
__global__ __launch_bounds__(kBlock) void tail_fused_kernel(SegArgs su, EtaPArgs a, int bu, int dp) {
  __shared__ double red[kRedGroup][kRedRows][kRedCols];  // (eta_p_kernel's pattern: the same sums in the same order)
  const int bx = static_cast<int>(blockIdx.x);
  const size_t slot = blockIdx.y;
  if (bx < bu) {
    seg_body<G, VEC, 8, 1>(su, bx * (kBlock / G) + threadIdx.x / G, dp, 1);
  } else if (bx < bu + a.nb_p) {
    p_update_block<kRedRows, kBlock / kRedCols>(red, bx - bu, a.partial + slot * a.bs_partial, a.chunk_off,
                                      a.p_old + slot * a.bs_p, a.p_new + slot * a.bs_p, a.pt_new + slot * a.bs_p,
                                      a.npr + slot * a.bs_p, a.n_ratings, a.kp, a.lp, a.normalize);
  } else {
    item_sum_block<GL, VECL>(bx - bu - a.nb_p, a.ttab + slot * a.bs_t, a.item_off, a.item_pairs, a.item_deg,
                             a.eta + slot * a.bs_eta, a.eta_new + slot * a.bs_eta, a.n_items, a.lp, a.normalize,
                             a.item_grid, a.n_ratings);
  }
}

}  // namespace

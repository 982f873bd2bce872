// eta_p.hpp -- p_update and item_sum, the two roles of eta_p_kernel
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
#pragma once

namespace {

// ======================================================================================
// kernel 4: p_update -- n_p[r][k][l] = p[r][k][l] * sum_{chunks c of r} S_c[k][l] (fixed
// order: 64 strided partial sums, then 8 sums of 8, then a sum of 8), followed by
// normalize_with_self over r (src/expectation_maximization.py:152-155; zero rows stay
// zero).  One block owns 16 (k,l) columns for ALL ratings (8 x 128 rows is 0.5 us faster launched
// back to back but 0.5 us slower inside the iteration; 4 x 256 rows is slower either way), so no
// inter-block hand-off is
// needed; every thread's slab loads are independent and issued back to back.  Writes
// p_new as [R][Kp][Lp] and transposed [R][Lp][Kp]; optionally the raw numerators.
// ======================================================================================
constexpr int kRedCols = 16, kRedRows = 64, kRedGroup = 6;  // ratings per LDS pass
constexpr int kRedThreads = kRedCols * kRedRows, kRedBatch = 5;  // 64 x 5 = 320 slabs per rating in one round (C3: 313)

// ROWS = rows of the reduction pattern (which fixes the association order of the sum over slabs); NTY = rows of
// threads that carry it out, ROWS / NTY pattern rows each.  eta_p_kernel: 64 / 64.  tail_fused_kernel
// (fused_small.hpp) has 256 threads: 64 / 16 -- the same sums in the same order, four rounds of loads per thread.
// RG = ratings per pass (every rating's sum is its own: the grouping does not touch the arithmetic).
template <int ROWS, int NTY = ROWS, int BATCH = kRedBatch, int RG = kRedGroup>
__device__ __forceinline__ void p_update_block(
    double (*red)[ROWS][kRedCols], int block, const double *__restrict__ partial,
    const int32_t *__restrict__ chunk_off, const double *__restrict__ p_old,
    double *__restrict__ p_new, double *__restrict__ pt_new, double *__restrict__ npr,
    int n_ratings, int kp, int lp, int normalize) {
  static_assert(ROWS % NTY == 0 && NTY >= 8, "pattern rows are dealt to the thread rows; the tree's second level needs 8");
  const int tx = threadIdx.x % kRedCols, ty = threadIdx.x / kRedCols;
  const int kl = kp * lp;
  const int col = block * kRedCols + tx;
  const bool ok = col < kl;
  double tot_all = 0.0;  // meaningful for ty == 0
  for (int r0 = 0; r0 < n_ratings; r0 += RG) {
    const int nr = min(RG, n_ratings - r0);
    int c0[RG], c1[RG];
    double s[RG], pold[RG];
    int longest = 0;
#pragma unroll
    for (int j = 0; j < RG; ++j)  // (needed at the very end: fetched up front, off the tail of the chain)
      pold[j] = (ty == 0 && ok && j < nr) ? p_old[static_cast<size_t>(r0 + j) * kl + col] : 0.0;
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const int r = min(r0 + j, n_ratings - 1);
      c0[j] = chunk_off[r];
      c1[j] = (j < nr) ? chunk_off[r + 1] : c0[j];
      longest = max(longest, c1[j] - c0[j]);
      s[j] = 0.0;
    }
    if (longest >= 0) STAMP(1);  // (diagnostic builds: the slab ranges have arrived)
    // (a pattern row adds its slabs c0 + vty, + ROWS, + 2 ROWS, ... one after the other: BATCH only says how many
    // of those loads are in flight together, it does not touch the order of the sum)
    constexpr int VR = ROWS / NTY;  // pattern rows per thread row: their loads share a round too
    double sv[VR][RG];
#pragma unroll
    for (int m = 0; m < VR; ++m)
#pragma unroll
      for (int j = 0; j < RG; ++j) sv[m][j] = 0.0;
    if (ok) {
      for (int off = 0; off < longest; off += ROWS * BATCH) {
        double v[VR][RG][BATCH];
#pragma unroll
        for (int m = 0; m < VR; ++m)
#pragma unroll
          for (int j = 0; j < RG; ++j)
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {  // every rating's slab loads issued together
              const int c = c0[j] + off + (ty + m * NTY) + i * ROWS;
              v[m][j][i] = (c < c1[j]) ? partial[static_cast<size_t>(c) * kl + col] : 0.0;
            }
#pragma unroll
        for (int m = 0; m < VR; ++m)
#pragma unroll
          for (int j = 0; j < RG; ++j)
#pragma unroll
            for (int i = 0; i < BATCH; ++i) sv[m][j] += v[m][j][i];
      }
    }
#pragma unroll
    for (int m = 0; m < VR; ++m)
#pragma unroll
      for (int j = 0; j < RG; ++j) red[j][ty + m * NTY][tx] = sv[m][j];
    STAMP(2);
    __syncthreads();
    STAMP(3);
    if (ty < 8) {  // ROWS rows -> 8 partial sums (fixed order)
#pragma unroll
      for (int j = 0; j < RG; ++j) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) t += red[j][ty * (ROWS / 8) + i][tx];
        s[j] = t;
      }
    }
    __syncthreads();
    if (ty < 8) {
#pragma unroll
      for (int j = 0; j < RG; ++j) red[j][ty][tx] = s[j];
    }
    __syncthreads();
    if (ty == 0 && ok) {
#pragma unroll
      for (int j = 0; j < RG; ++j) {
        if (j < nr) {
          double tot = red[j][0][tx];
#pragma unroll
          for (int i = 1; i < 8; ++i) tot += red[j][i][tx];
          const size_t e = static_cast<size_t>(r0 + j) * kl + col;
          const double raw = pold[j] * tot;
          npr[e] = raw;
          tot_all += raw;
          s[j] = raw;  // stays in registers for the single-group case below
        }
      }
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    if (n_ratings <= RG) {  // common case: normalise straight from registers
      if (ty == 0 && ok && normalize) {
        const double den = (tot_all == 0.0) ? 1.0 : tot_all;
        const int k = col / lp, l = col % lp;
#pragma unroll
        for (int j = 0; j < RG; ++j) {
          if (j < nr) {
            const double v = s[j] / den;
            p_new[static_cast<size_t>(j) * kl + col] = v;
            pt_new[static_cast<size_t>(j) * kl + static_cast<size_t>(l) * kp + k] = v;
          }
        }
      }
      return;
    }
  }
  if (ty == 0 && ok && normalize) {
    const double den = (tot_all == 0.0) ? 1.0 : tot_all;
    const int k = col / lp, l = col % lp;
    for (int r = 0; r < n_ratings; ++r) {
      const size_t e = static_cast<size_t>(r) * kl + col;
      const double v = npr[e] / den;  // this thread's own stores: program order suffices
      p_new[e] = v;
      pt_new[static_cast<size_t>(r) * kl + static_cast<size_t>(l) * kp + k] = v;
    }
  }
}

// ======================================================================================
// item_sum -- eta_new[i,:] = eta[i,:] * sum_{q in item i} T[q,:] / d_i  (src/mmsbm.py:249).
// One group of G lanes per item.
// ======================================================================================
// NB rows of T by the pair ids at row[0 .. NB) (-1: no such pair), all loads in flight together, added in order
template <int VEC, int NB>
__device__ __forceinline__ void grid_rows(const int32_t *__restrict__ row, const double *__restrict__ ttab, int lp,
                                          int lane_off, double (&acc)[VEC]) {
  int id[NB];
  double t[NB][VEC];
#pragma unroll
  for (int b = 0; b < NB; ++b) id[b] = row[b];
#pragma unroll
  for (int b = 0; b < NB; ++b) load_vec<VEC>(ttab + static_cast<size_t>(max(id[b], 0)) * lp + lane_off, t[b]);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (id[b] >= 0) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += t[b][v];
    }
  }
}

template <int G, int VEC>
__device__ __forceinline__ void item_sum_block(
    int block, const double *__restrict__ ttab, const int32_t *__restrict__ item_off,
    const int32_t *__restrict__ item_pairs, const int32_t *__restrict__ item_deg,
    const double *__restrict__ eta, double *__restrict__ eta_new, int n_items, int lp,
    int normalize, const int32_t *__restrict__ item_grid, int n_ratings) {
  // T rows in flight per group: 8 while a row piece is 4 doubles per lane; wide rows (8 or 16 doubles per lane) would
  // need 128 - 256 registers for that and spill (eta_p at L = 520: 780 us, most of it scratch traffic)
  constexpr int B = VEC <= 4 ? 8 : (VEC == 8 ? 4 : 2);
  const int it = block * (static_cast<int>(blockDim.x) / G) + threadIdx.x / G;
  const int gl = threadIdx.x % G;
  if (it >= n_items) return;
  // (rows of more than G x VEC = 1,024 groups: one more trip per 1,024 columns -- every column is its own sum)
  for (int lane_off = gl * VEC; lane_off < lp; lane_off += G * VEC) {
  double acc[VEC], e[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
  // (the item's own row: asked for up front where the registers allow it -- it is needed last; with 16 doubles per lane
  // it would sit beside two T rows in flight and the sums, 128 registers, and spill: fetched after the sums there)
  if (VEC <= 8) load_vec<VEC>(eta + static_cast<size_t>(it) * lp + lane_off, e);
  if (item_grid) {
    // dense data (most (item, rating) combinations occur): the item's pairs sit in a fixed-width
    // grid row (-1: no such pair; ascending rating like the CSR list, so the sums are the same), one
    // dependent load level less than offsets -> pair ids -> rows
    const int32_t *row = item_grid + static_cast<size_t>(it) * n_ratings;
    if (n_ratings <= B) {   // one round of loads (a rating beyond the last repeats the last one and is not added)
      int id[B];
      double t[B][VEC];
#pragma unroll
      for (int b = 0; b < B; ++b) id[b] = row[min(b, n_ratings - 1)];
#pragma unroll
      for (int b = 0; b < B; ++b) load_vec<VEC>(ttab + static_cast<size_t>(max(id[b], 0)) * lp + lane_off, t[b]);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if (b < n_ratings && id[b] >= 0) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += t[b][v];
        }
      }
    } else {                // full rounds, then the rest in rounds of 4, 2, 1: no row is asked for twice (R = 10: 8 + 2)
      int j = 0;
      for (; j + B <= n_ratings; j += B) grid_rows<VEC, B>(row + j, ttab, lp, lane_off, acc);
      const int rem = n_ratings - j;
      if (B > 4 && (rem & 4)) { grid_rows<VEC, 4>(row + j, ttab, lp, lane_off, acc); j += 4; }
      if (B > 2 && (rem & 2)) { grid_rows<VEC, 2>(row + j, ttab, lp, lane_off, acc); j += 2; }
      if (rem & 1) grid_rows<VEC, 1>(row + j, ttab, lp, lane_off, acc);
    }
  }
  const int beg = item_grid ? 0 : item_off[it], end = item_grid ? 0 : item_off[it + 1];
  for (int j = beg; j < end; j += B) {
    int id[B];
    double t[B][VEC];
#pragma unroll
    for (int b = 0; b < B; ++b) id[b] = item_pairs[min(j + b, end - 1)];
#pragma unroll
    for (int b = 0; b < B; ++b) load_vec<VEC>(ttab + static_cast<size_t>(id[b]) * lp + lane_off, t[b]);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      if (j + b < end) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += t[b][v];
      }
    }
  }
  if (VEC > 8) load_vec<VEC>(eta + static_cast<size_t>(it) * lp + lane_off, e);
  const double d = static_cast<double>(max(item_deg[it], 1));
#pragma unroll
  for (int v = 0; v < VEC; ++v) e[v] = normalize ? (e[v] * acc[v]) / d : e[v] * acc[v];
  store_vec<VEC>(eta_new + static_cast<size_t>(it) * lp + lane_off, e);
  }
}

// eta_p -- the two independent updates that follow the T / slab stage share ONE launch:
// blocks [0, nb_p) run p_update_block, the rest run item_sum_block.
struct EtaPArgs {
  const double *partial; const int32_t *chunk_off; const double *p_old;
  double *p_new; double *pt_new; double *npr;
  const double *ttab; const int32_t *item_off; const int32_t *item_pairs; const int32_t *item_deg;
  const double *eta; double *eta_new;
  int n_ratings, kp, lp, n_items, normalize, nb_p;
#ifdef MMSBM_ABLATE
  int abl;  // diagnostic build only: 64 skips p_update, 128 item_sum (mmsbm_hip_time_stage, stage >> 8)
#endif
  size_t bs_partial, bs_p, bs_t, bs_eta;  // restart slots (blockIdx.y): table strides
  const int32_t *item_grid;               // [n_items][n_ratings] pair id or -1 (dense data), else null
};

// The same two roles in workgroups of 256 threads, for launches of many rounds (BASELINE's config 5: 100,000 items).
// With 1,024 threads and 126 registers a CU holds ONE workgroup: 1,563 item_sum workgroups are 6.1 rounds, paid as 7, and
// the p_update workgroups take a CU each while they wait out their slab loads.  Four waves per workgroup are dealt
// wave by wave: the last round is a 24th of the launch, not a 7th.  p_update as tail_fused_kernel runs it (the
// 64-row pattern carried by 16 rows of threads: the same sums in the same order), two ratings per pass so that its
// registers and the 16 KB of LDS leave room for seven workgroups beside it.  Results are bitwise those of eta_p_kernel.
constexpr int kRedGroupW4 = 2;
template <int G, int VEC>
__global__ __launch_bounds__(kBlock) void eta_p_w4_kernel(EtaPArgs a) {
  __shared__ double red[kRedGroupW4][kRedRows][kRedCols];
  const size_t slot = blockIdx.y;
#ifdef MMSBM_ABLATE
  if (a.abl & (static_cast<int>(blockIdx.x) < a.nb_p ? 64 : 128)) return;  // skip a role
#endif
  if (static_cast<int>(blockIdx.x) < a.nb_p)
    p_update_block<kRedRows, kBlock / kRedCols, 2, kRedGroupW4>(red, blockIdx.x, a.partial + slot * a.bs_partial, a.chunk_off,
                                                                a.p_old + slot * a.bs_p, a.p_new + slot * a.bs_p,
                                                                a.pt_new + slot * a.bs_p, a.npr + slot * a.bs_p, a.n_ratings,
                                                                a.kp, a.lp, a.normalize);
  else
    item_sum_block<G, VEC>(blockIdx.x - a.nb_p, a.ttab + slot * a.bs_t, a.item_off, a.item_pairs,
                           a.item_deg, a.eta + slot * a.bs_eta, a.eta_new + slot * a.bs_eta,
                           a.n_items, a.lp, a.normalize, a.item_grid, a.n_ratings);
}

template <int G, int VEC>
__global__ __launch_bounds__(kRedThreads) void eta_p_kernel(EtaPArgs a) {
  __shared__ double red[kRedGroup][kRedRows][kRedCols];
  const size_t slot = blockIdx.y;
#ifdef MMSBM_ABLATE
  if (a.abl & (static_cast<int>(blockIdx.x) < a.nb_p ? 64 : 128)) return;  // skip a role
#endif
  if (static_cast<int>(blockIdx.x) < a.nb_p)
    p_update_block<kRedRows>(red, blockIdx.x, a.partial + slot * a.bs_partial, a.chunk_off,
                             a.p_old + slot * a.bs_p, a.p_new + slot * a.bs_p,
                             a.pt_new + slot * a.bs_p, a.npr + slot * a.bs_p, a.n_ratings, a.kp,
                             a.lp, a.normalize);
  else
    item_sum_block<G, VEC>(blockIdx.x - a.nb_p, a.ttab + slot * a.bs_t, a.item_off, a.item_pairs,
                           a.item_deg, a.eta + slot * a.bs_eta, a.eta_new + slot * a.bs_eta,
                           a.n_items, a.lp, a.normalize, a.item_grid, a.n_ratings);
}

// ---- host side: the argument blocks of these kernels from the context ----
EtaPArgs eta_p_args(const mmsbm_hip_ctx *c, bool commit, int cols_per_block) {
  const int cur = c->cur, nxt = cur ^ 1;
  EtaPArgs a;
  const int s = c->base_slot;
  a.partial = c->partial.at(s);
  a.chunk_off = c->mv_chunk_off.ptr;
  a.p_old = c->p[cur].at(s); a.p_new = c->p[nxt].at(s); a.pt_new = c->pt[nxt].at(s);
  a.npr = c->npr.at(s);
  a.ttab = c->ttab.at(s); a.item_off = c->item_off.ptr; a.item_pairs = c->item_pairs.ptr;
  a.item_deg = c->item_deg.ptr; a.eta = c->eta[cur].at(s); a.eta_new = c->eta[nxt].at(s);
  a.bs_partial = c->partial.stride; a.bs_p = c->p[0].stride; a.bs_t = c->ttab.stride;
  a.bs_eta = c->eta[0].stride;
  a.n_ratings = c->n_ratings; a.kp = c->kp; a.lp = c->lp; a.n_items = c->n_items;
  a.normalize = commit ? 1 : 0;
#ifdef MMSBM_ABLATE
  a.abl = c->ablate;
#endif
  a.nb_p = (c->kp * c->lp + cols_per_block - 1) / cols_per_block;
  a.item_grid = c->item_grid.count ? c->item_grid.ptr : nullptr;
  return a;
}

}  // namespace

// tu_once.hip -- translation unit of the once-per-run kernels (once_kernels.hpp, lik_fact.hpp): the likelihood in its
// forms, prod_dist / predict / score, compute_omegas, the random start -- with the host code that chooses among them
#include "prelude.hpp"
#include "pcg64.hpp"
#include "once_kernels.hpp"
#include "lik_fact.hpp"

namespace {

// likelihood_fast_kernel: tile + its logarithms in LDS?  (always with several lanes per triple)
bool lik_fast_tile_lds(const mmsbm_hip_ctx *c) {
  return c->lp > 20 || c->lik_g > 1 ||
         2 * static_cast<size_t>(c->kp) * c->lp * sizeof(double) > kScalarTileBytes;
}
bool lik_fast_usable(const mmsbm_hip_ctx *c) {
  const size_t lds = lik_fast_tile_lds(c) ? 2 * static_cast<size_t>(c->kp) * c->lp * sizeof(double) : 0;
  return c->lik_mode >= 1 && c->lp <= 160 && c->n_lik_units > 0 && lds <= kLdsMax - 4096;
}
// likelihood of the selected slot through the logarithm tables; returns the number of partial sums
int likelihood_fast(mmsbm_hip_ctx *c) {
  ensure_a(c);  // (s_n = theta_n . A[q_n])
  const int cur = c->cur, sl = c->sel;
  const size_t nt = static_cast<size_t>(c->n_users) * c->kp, ne = static_cast<size_t>(c->n_items) * c->lp;
  const size_t np = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  if (c->lg_theta.count < nt) c->lg_theta.alloc(nt);
  if (c->lg_eta.count < ne) c->lg_eta.alloc(ne);
  if (c->lg_p.count < np) c->lg_p.alloc(np);
  auto logs = [&](const double *in, double *out, size_t n) {
    if (n == 0) return;
    LAUNCH(log_table_kernel, static_cast<unsigned>((n + kBlock - 1) / kBlock), kBlock, 0, c->stream, in, out, n);
  };
  const RowTab th = theta_tab(c, cur);
  const RowTab lth{c->lg_theta.ptr, c->lg_theta.ptr + static_cast<size_t>(c->n_users) * th.mw, th.mw, th.tw,
                   th.mw, th.tw, 0, 0};  // main + tail like theta, one slot
  if (nt > 0)
    LAUNCH(log_rows_kernel, static_cast<unsigned>((nt + kBlock - 1) / kBlock), kBlock, 0, c->stream, th, lth,
           static_cast<size_t>(c->n_users), c->kp);
  logs(c->eta[cur].at(sl), c->lg_eta.ptr, ne);
  logs(c->p[cur].at(sl), c->lg_p.ptr, np);
  const int nb = c->n_lik_units;
  if (c->lik_part.count < static_cast<size_t>(nb)) c->lik_part.alloc(nb);
  // lanes per triple and columns per lane: at most ~20 columns (40 + 40 registers) per lane
  int G = c->lp <= 20 ? 1 : (c->lp <= 40 ? 2 : 4);
  if (c->lik_g > 0) G = c->lik_g;  // tuning override
  while (G < 8 && (c->lp + G - 1) / G > 20) G *= 2;
  const int LW = ((c->lp + G - 1) / G + 3) / 4 * 4;
  const bool tl = lik_fast_tile_lds(c);
  const size_t lds = tl ? 2 * static_cast<size_t>(c->kp) * c->lp * sizeof(double) : 0;
#define LIK_GO(LW_, G_, TL_)                                                                      \
  allow_big_lds(likelihood_fast_kernel<LW_, G_, TL_>, lds);                                       \
  LAUNCH((likelihood_fast_kernel<LW_, G_, TL_>), nb, kLikThreads, lds, c->stream,                 \
         c->lik_units.ptr, c->pair_off.ptr, c->pair_user.ptr, c->pair_item.ptr, th, lth, a_tab(c, cur), \
         c->eta[cur].at(sl), c->lg_eta.ptr, c->p[cur].at(sl), c->lg_p.ptr, c->lik_part.ptr, c->k, \
         c->l, c->kp, c->lp)
#define LIK_LW(G_, TL_)                                                                           \
  do {                                                                                            \
    switch (LW) {                                                                                 \
      case 4: LIK_GO(4, G_, TL_); break;                                                          \
      case 8: LIK_GO(8, G_, TL_); break;                                                          \
      case 12: LIK_GO(12, G_, TL_); break;                                                        \
      case 16: LIK_GO(16, G_, TL_); break;                                                        \
      default: LIK_GO(20, G_, TL_); break;                                                        \
    }                                                                                             \
  } while (0)
  if (G == 1) {
    if (tl) LIK_LW(1, true); else LIK_LW(1, false);
  } else if (G == 2) {
    LIK_LW(2, true);
  } else if (G == 4) {
    LIK_LW(4, true);
  } else {
    LIK_LW(8, true);
  }
#undef LIK_LW
#undef LIK_GO
  return nb;
}

// ---- the likelihood pair by pair (lik_fact.hpp): logarithm tables, then one wave per (item, rating) pair.
// For rows of more than 32 groups (one column per lane: narrower rows would leave most of a wave idle), tiles
// that fit the LDS beside their logarithms, and data with a few triples per pair (the pair's eta p products
// are shared by four triples at a time).  Returns the number of partial sums.
bool lik_pairs_usable(const mmsbm_hip_ctx *c) {
  return c->lik_mode == 2 && c->lp > 32 && c->lp <= 192 && c->kp <= 192 && c->n_lik_units > 0 && c->n_pairs > 0 &&
         (2 * static_cast<size_t>(c->kp) * c->lp + c->kp) * sizeof(double) <= kLdsMax - 4096 &&
         c->n_obs * 2 >= static_cast<int64_t>(c->n_pairs) * 5;
}
int likelihood_pairs(mmsbm_hip_ctx *c) {
  ensure_a(c);  // (s_t = theta_t . A[q])
  const int cur = c->cur, sl = c->sel;
  hipStream_t st = c->stream;
  const size_t nt = static_cast<size_t>(c->n_users) * c->kp, ne = static_cast<size_t>(c->n_items) * c->lp;
  const size_t np = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  if (c->lg_theta.count < 2 * nt) c->lg_theta.alloc(2 * nt);  // here: (theta, log theta) pairs, plain rows
  if (c->lg_eta.count < ne) c->lg_eta.alloc(ne);
  if (c->lg_p.count < np) c->lg_p.alloc(np);
  auto blocks = [](size_t n) { return static_cast<unsigned>((n + kBlock - 1) / kBlock); };
  const RowTab th = theta_tab(c, cur);
  double2 *tl = reinterpret_cast<double2 *>(c->lg_theta.ptr);
  const double *eta = c->eta[cur].at(sl), *p = c->p[cur].at(sl);
  if (nt > 0) LAUNCH(theta_log_pairs_kernel, blocks(nt), kBlock, 0, st, th, tl, static_cast<size_t>(c->n_users), c->kp);
  if (ne > 0) LAUNCH(log_table_kernel, blocks(ne), kBlock, 0, st, eta, c->lg_eta.ptr, ne);
  LAUNCH(log_table_kernel, blocks(np), kBlock, 0, st, p, c->lg_p.ptr, np);
  const int nb = c->n_lik_units;
  if (c->lik_part.count < static_cast<size_t>(nb)) c->lik_part.alloc(static_cast<size_t>(nb));
  const size_t lds = (2 * static_cast<size_t>(c->kp) * c->lp + c->kp) * sizeof(double);
#define WAVE_GO(LW_)                                                                                 \
  do {                                                                                               \
    allow_big_lds(lik_wave_kernel<LW_>, lds);                                                        \
    LAUNCH(lik_wave_kernel<LW_>, nb, kLikWaveThreads, lds, st,                                       \
           c->lik_units.ptr, c->pair_off.ptr, c->pair_user.ptr, c->pair_item.ptr, tl, a_tab(c, cur), \
           eta, c->lg_eta.ptr, p, c->lg_p.ptr, c->lik_part.ptr, c->k, c->l, c->kp, c->lp);           \
  } while (0)
  if (c->lp <= 64) WAVE_GO(1); else if (c->lp <= 128) WAVE_GO(2); else WAVE_GO(3);
#undef WAVE_GO
  return nb;
}

// ---- rows of up to 32 groups: a lane per triple, the tile through scalar loads (lik_fact.hpp: lik_lane_kernel) ----
constexpr int kLikLaneThreads = 128;
bool lik_lanes_usable(const mmsbm_hip_ctx *c) {
  return c->lik_mode == 2 && c->kp <= 32 && c->lp <= 32 && c->n_lik_units > 0 && c->n_pairs > 0;
}
int likelihood_lanes(mmsbm_hip_ctx *c) {
  ensure_a(c);  // (s_n = theta_n . A[q_n])
  const int cur = c->cur, sl = c->sel;
  hipStream_t st = c->stream;
  const size_t nt = static_cast<size_t>(c->n_users) * c->kp, ne = static_cast<size_t>(c->n_items) * c->lp;
  const size_t np = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  if (c->lg_theta.count < 2 * nt) c->lg_theta.alloc(2 * nt);  // (value, logarithm) pairs, plain rows
  if (c->lg_eta.count < 2 * ne) c->lg_eta.alloc(2 * ne);
  if (c->lg_p.count < 2 * np + 16) c->lg_p.alloc(2 * np + 16);   // (+ 16: lik_lane_kernel's scalar loads run one chunk ahead)
  auto blocks = [](size_t n) { return static_cast<unsigned>((n + kBlock - 1) / kBlock); };
  double2 *tl = reinterpret_cast<double2 *>(c->lg_theta.ptr), *el = reinterpret_cast<double2 *>(c->lg_eta.ptr);
  double2 *ptl = reinterpret_cast<double2 *>(c->lg_p.ptr);
  if (nt > 0) LAUNCH(theta_log_pairs_kernel, blocks(nt), kBlock, 0, st, theta_tab(c, cur), tl, static_cast<size_t>(c->n_users), c->kp);
  if (ne > 0) LAUNCH(theta_log_pairs_kernel, blocks(ne), kBlock, 0, st, plain_tab(c->eta[cur].at(sl), c->lp), el, static_cast<size_t>(c->n_items), c->lp);
  LAUNCH(theta_log_pairs_kernel, blocks(np), kBlock, 0, st, plain_tab(c->pt[cur].at(sl), c->kp), ptl,
         static_cast<size_t>(c->n_ratings) * c->lp, c->kp);   // pT: [R][lp][kp]
  const int nb = c->n_lik_units;
  if (c->lik_part.count < static_cast<size_t>(nb)) c->lik_part.alloc(static_cast<size_t>(nb));
#define LANE_GO(KP_)                                                                                          \
  LAUNCH((lik_lane_kernel<KP_, kLikLaneThreads>), nb, kLikLaneThreads, 0, st,                                 \
         c->lik_units.ptr, c->pair_off.ptr, c->pair_user.ptr, c->pair_item.ptr, tl, a_tab(c, cur), el, ptl,  \
         c->lik_part.ptr, c->k, c->l, c->lp)
  switch (c->kp) {
    case 4: LANE_GO(4); break;
    case 8: LANE_GO(8); break;
    case 12: LANE_GO(12); break;
    case 16: LANE_GO(16); break;
    case 20: LANE_GO(20); break;
    case 24: LANE_GO(24); break;
    case 28: LANE_GO(28); break;
    default: LANE_GO(32); break;
  }
#undef LANE_GO
  return nb;
}

}  // namespace

namespace mmsbm_hip_impl {

// likelihood of the selected slot (the caller holds a OneSlot): kernels onto the context's stream, no wait;
// returns the number of partial sums likelihood_finish adds up
int likelihood_enqueue(mmsbm_hip_ctx *ctx) {
  const int cur = ctx->cur, sl = ctx->sel;
  int nb;
  if (lik_pairs_usable(ctx)) {
    nb = likelihood_pairs(ctx);
  } else if (lik_lanes_usable(ctx)) {
    nb = likelihood_lanes(ctx);
  } else if (lik_fast_usable(ctx)) {
    nb = likelihood_fast(ctx);
  } else {
    const size_t lik_lds = static_cast<size_t>(ctx->kp + ctx->lp) * kLikThreads * sizeof(double);
    if (lik_lds <= kLdsMax - 2048 && ctx->n_lik_units > 0) {
      nb = ctx->n_lik_units;
      allow_big_lds(likelihood_units_kernel, lik_lds);
      if (ctx->lik_part.count < static_cast<size_t>(nb)) ctx->lik_part.alloc(nb);
      LAUNCH(likelihood_units_kernel, nb, kLikThreads, lik_lds, ctx->stream,
             ctx->lik_units.ptr, ctx->pair_off.ptr, ctx->pair_user.ptr, ctx->pair_item.ptr,
             theta_tab(ctx, cur), ctx->eta[cur].at(sl), ctx->p[cur].at(sl), ctx->lik_part.ptr, ctx->k,
             ctx->l, ctx->kp, ctx->lp);
    } else {
      nb = static_cast<int>(std::min<int64_t>((ctx->n_obs + kBlock - 1) / kBlock, 4096));
      nb = std::max(nb, 1);
      LAUNCH(likelihood_kernel, nb, kBlock, 0, ctx->stream,
             ctx->orig_u.ptr, ctx->orig_i.ptr, ctx->orig_r.ptr, theta_tab(ctx, cur),
             ctx->eta[cur].at(sl), ctx->p[cur].at(sl), ctx->lik_part.ptr, ctx->n_obs, ctx->k, ctx->l,
             ctx->kp, ctx->lp);
    }
  }
  HIP_CHECK(hipGetLastError());
  return nb;
}

// B for the selected slot (the caller holds a OneSlot and rows_fast_prepare said yes), then one group of
// lanes per row.
// mode 0: dist[m][r] = P[m, r];  mode 1: dist += P, block_out = the restart's six sums per workgroup
int rows_launch(mmsbm_hip_ctx *c, int mode, const int32_t *pu, const int32_t *pi, const int32_t *preal,
                const double *weights, double *dist, double *block_out, int64_t n_rows, int first) {
  stage_matvec_a(c, c->cur, c->cur, true);
  const int per = kBlock / group_lanes(c->code_k);
  const int nb = static_cast<int>((n_rows + per - 1) / per);
  const size_t rstride = static_cast<size_t>(c->n_items) * c->kp;
#define CALL(G, V)                                                                                           \
  do {                                                                                                       \
    if (mode == 0)                                                                                           \
      LAUNCH((predict_rows_kernel<G, V, 0>), nb, kBlock, 0, c->stream, pu, pi, preal, theta_tab(c, c->cur), c->btab.ptr, \
             rstride, weights, dist, block_out, n_rows, c->n_ratings, c->kp, first);                         \
    else                                                                                                     \
      LAUNCH((predict_rows_kernel<G, V, 1>), nb, kBlock, 0, c->stream, pu, pi, preal, theta_tab(c, c->cur), c->btab.ptr, \
             rstride, weights, dist, block_out, n_rows, c->n_ratings, c->kp, first);                         \
  } while (0)
  DISPATCH_GV(c->code_k, CALL);
#undef CALL
  HIP_CHECK(hipGetLastError());
  return nb;
}

// the per-row scoring kernels (fewer rows than items, or predict_fast = 0) and the final pass over the mean; the six sums
// per workgroup go to ps_part.  Returns the number of workgroups.
int score_rows_launch(mmsbm_hip_ctx *ctx, bool finish) {
  const int nb = static_cast<int>((ctx->ps_rows + kBlock - 1) / kBlock);
  if (nb <= 0) return 0;
  const int cur = ctx->cur, sl = ctx->sel;
  if (finish)
    LAUNCH(predict_score_kernel<true>, nb, kBlock, 0, ctx->stream,
           ctx->ps_u.ptr, ctx->ps_i.ptr, ctx->ps_r.ptr, theta_tab(ctx, cur), ctx->eta[cur].at(sl),
           ctx->p[cur].at(sl), ctx->ps_w.ptr, ctx->ps_sum.ptr, ctx->ps_part.ptr, ctx->ps_rows,
           ctx->n_ratings, ctx->k, ctx->l, ctx->kp, ctx->lp, 0, static_cast<double>(ctx->ps_added));
  else
    LAUNCH(predict_score_kernel<false>, nb, kBlock, 0, ctx->stream,
           ctx->ps_u.ptr, ctx->ps_i.ptr, ctx->ps_r.ptr, theta_tab(ctx, cur), ctx->eta[cur].at(sl),
           ctx->p[cur].at(sl), ctx->ps_w.ptr, ctx->ps_sum.ptr, ctx->ps_part.ptr, ctx->ps_rows,
           ctx->n_ratings, ctx->k, ctx->l, ctx->kp, ctx->lp, ctx->ps_added == 0 ? 1 : 0, 1.0);
  HIP_CHECK(hipGetLastError());
  return nb;
}
int score_stats_count() { return kScoreStats; }

// theta0 / eta0 of the selected slot drawn on the device from the restart's PCG64 stream (pcg64.hpp)
void init_rows_launch(mmsbm_hip_ctx *ctx, const uint64_t pcg64_state[4]) {
  const int cur = ctx->cur, sl = ctx->sel;
  // the reference draws theta (external users x K) first, then eta; internally the two sides
  // may be swapped, the stream offsets are not
  const uint64_t n_theta_ext = static_cast<uint64_t>(ctx->ext_users) * ctx->ext_k;
  const uint64_t off_users = ctx->swapped ? n_theta_ext : 0;  // internal users' table
  const uint64_t off_items = ctx->swapped ? 0 : n_theta_ext;
  const RowTab tt = theta_tab(ctx, cur), et = plain_tab(ctx->eta[cur].at(sl), ctx->lp);
  auto blocks = [](uint64_t total) {
    return static_cast<unsigned>((total + uint64_t(kBlock) * kDrawsPerThread - 1) / (uint64_t(kBlock) * kDrawsPerThread));
  };
  const uint64_t nu = static_cast<uint64_t>(ctx->n_users) * ctx->k, ni = static_cast<uint64_t>(ctx->n_items) * ctx->l;
  if (nu > 0)
    LAUNCH(init_rows_kernel, blocks(nu), kBlock, 0, ctx->stream, tt, ctx->user_off.ptr, nullptr, ctx->n_users, ctx->k,
           pcg64_state[0], pcg64_state[1], pcg64_state[2], pcg64_state[3], off_users);
  if (ni > 0)
    LAUNCH(init_rows_kernel, blocks(ni), kBlock, 0, ctx->stream, et, nullptr, ctx->item_deg.ptr, ctx->n_items, ctx->l,
           pcg64_state[0], pcg64_state[1], pcg64_state[2], pcg64_state[3], off_items);
  HIP_CHECK(hipGetLastError());
}

// omega[n,k,l] of the selected slot into dev_out (n_elems = N K L), in the caller's (external) orientation
void omegas_launch(mmsbm_hip_ctx *ctx, double *dev_out, int64_t n_elems) {
  const int cur = ctx->cur, sl = ctx->sel;
  // internal (k,l) -> external position: not swapped [k][l] strides (L,1); swapped the
  // external tensor is [l_int][k_int] so strides are (1, K_int)
  const int sk = ctx->swapped ? 1 : ctx->l;
  const int sl_stride = ctx->swapped ? ctx->k : 1;
  const int64_t nb = (n_elems + kBlock - 1) / kBlock;
  LAUNCH(omegas_kernel, static_cast<unsigned>(nb), kBlock, 0, ctx->stream,
         ctx->orig_u.ptr, ctx->orig_i.ptr, ctx->orig_r.ptr, theta_tab(ctx, cur),
         ctx->eta[cur].at(sl), ctx->p[cur].at(sl), dev_out, n_elems, ctx->k, ctx->l, ctx->kp,
         ctx->lp, sk, sl_stride);
  HIP_CHECK(hipGetLastError());
}

// P[m, r] of the selected slot for n_pairs (user, item) rows, a thread per element (few rows, or predict_fast = 0)
void prod_dist_launch(mmsbm_hip_ctx *ctx, const int32_t *du, const int32_t *di, double *dout, int64_t n_pairs) {
  const int cur = ctx->cur, sl = ctx->sel;
  const int64_t n_elems = n_pairs * ctx->n_ratings;
  const int64_t nb = (n_elems + kBlock - 1) / kBlock;
  LAUNCH(prod_dist_kernel, static_cast<unsigned>(nb), kBlock, 0, ctx->stream,
         du, di, theta_tab(ctx, cur), ctx->eta[cur].at(sl), ctx->p[cur].at(sl), dout,
         n_pairs, ctx->n_ratings, ctx->k, ctx->l, ctx->kp, ctx->lp);
  HIP_CHECK(hipGetLastError());
}

}  // namespace mmsbm_hip_impl

// tu_fused.hip -- translation unit of the two-launch iteration of small problems: pairs_fused_kernel and
// tail_fused_kernel (fused_small.hpp; they reuse device code of seg_pass.hpp, pair_block.hpp and eta_p.hpp)
#include "prelude.hpp"
#include "seg_pass.hpp"
#include "pair_block.hpp"
#include "eta_p.hpp"
#include "fused_small.hpp"

namespace mmsbm_hip_impl {

void stage_fused_pairs(mmsbm_hip_ctx *c) {
  LaunchScope ls(c, K_FUSED_PAIRS, true);
  const int s = c->base_slot, cur = c->cur;
  FusedPairArgs fa{};
  fa.pt_tiles = c->pt[cur].at(s); fa.p_tiles = c->p[cur].at(s); fa.eta = c->eta[cur].at(s);
  fa.theta = theta_tab(c, cur); fa.a_out = a_tab(c, cur);
  fa.pair_off = c->pair_off.ptr; fa.pair_user = c->pair_user.ptr; fa.pair_item = c->pair_item.ptr;
  fa.chunks = c->mv_chunks.ptr; fa.t_out = c->ttab.at(s); fa.partial = c->partial.at(s);
  fa.kp = c->kp; fa.lp = c->lp; fa.spb = c->pb_spb; fa.nsub = c->pb_nsub; fa.nt = nt_on(c) & 1;
  fa.bs_tiles = c->p[0].stride; fa.bs_eta = c->eta[0].stride; fa.bs_t = c->ttab.stride; fa.bs_partial = c->partial.stride;
  const bool split = c->fs_pairs && !c->lay.pair_work.splits.empty();
  fa.units = split ? c->fp_units.ptr : nullptr; fa.items = split ? c->fp_items.ptr : nullptr;
  fa.splits = split ? c->fp_splits.ptr : nullptr;
  const size_t lds = pairs_fused_lds(c->kp, c->lp, split ? c->fp_max_parts : 0);
  const dim3 grid = slot_grid(c, c->n_chunks);
#define PF(G, S)                                                                  \
  do {                                                                            \
    allow_big_lds(pairs_fused_kernel<G, 4, S>, lds);                              \
    LAUNCH_IN(ls, (pairs_fused_kernel<G, 4, S>), grid, kBlock, lds, c->stream, fa); \
  } while (0)
  if (c->code_k == 0) { if (split) PF(4, true); else PF(4, false); }
  else { if (split) PF(8, true); else PF(8, false); }
#undef PF
  ls.done();
}
void stage_fused_tail(mmsbm_hip_ctx *c, bool commit) {
  LaunchScope ls(c, K_FUSED_TAIL, true);
  const SegArgs su = seg_users_args(c, commit, c->n_users);
  const EtaPArgs a = eta_p_args(c, commit, kRedCols);
  const int per_u = kBlock / group_lanes(c->code_k), per_i = kBlock / group_lanes(c->code_l);
  const bool split = c->fs_users && !c->lay.user_work.splits.empty();
  const int bu = split ? c->fu_blocks : (su.nseg + per_u - 1) / per_u, nb_i = (c->n_items + per_i - 1) / per_i;
  const dim3 grid = slot_grid(c, bu + a.nb_p + nb_i);
  const FusedUserArgs fu{split ? c->fu_units.ptr : nullptr, split ? c->fu_items.ptr : nullptr, split ? c->fu_splits.ptr : nullptr};
  const size_t lds = split ? static_cast<size_t>(c->fu_max_parts) * c->kp * sizeof(double) : 0;
  // (32 rows in flight per user segment leave one workgroup per CU: only while that is a single round)
  const bool deep = static_cast<long long>(grid.x) * grid.y <= c->n_cus;
#define TAIL(G, V, GL, VL)                                                                                        \
  do {                                                                                                            \
    if (split) {                                                                                                  \
      allow_big_lds(tail_fused_kernel<G, V, GL, VL, 16, true>, lds);                                              \
      LAUNCH_IN(ls, (tail_fused_kernel<G, V, GL, VL, 16, true>), grid, kBlock, lds, c->stream, su, a, bu, c->kp, fu); \
    } else if (deep) LAUNCH_IN(ls, (tail_fused_kernel<G, V, GL, VL, 32, false>), grid, kBlock, 0, c->stream, su, a, bu, c->kp, fu); \
    else LAUNCH_IN(ls, (tail_fused_kernel<G, V, GL, VL, 16, false>), grid, kBlock, 0, c->stream, su, a, bu, c->kp, fu);      \
  } while (0)
  switch (c->code_k * 2 + c->code_l) {
    case 0: TAIL(4, 4, 4, 4); break;
    case 1: TAIL(4, 4, 8, 4); break;
    case 2: TAIL(8, 4, 4, 4); break;
    default: TAIL(8, 4, 8, 4); break;
  }
#undef TAIL
  ls.done();
}

}  // namespace mmsbm_hip_impl

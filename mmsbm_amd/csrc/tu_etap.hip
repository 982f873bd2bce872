// tu_etap.hip -- translation unit of eta_p: p_update and item_sum in one launch (eta_p.hpp)
#include "prelude.hpp"
#include "eta_p.hpp"

namespace mmsbm_hip_impl {

void stage_eta_p(mmsbm_hip_ctx *c, bool commit) {  // eta_new ; p_new, pT_new, raw n_p
  LaunchScope ls(c, K_ETAP, true);
  const EtaPArgs a = eta_p_args(c, commit, kRedCols);
  const int per = kRedThreads / group_lanes(c->code_l);
  const int nb_i = (c->n_items + per - 1) / per;
  // launches of several rounds of 1,024-thread workgroups (one per CU at a time): 256-thread workgroups instead, dealt
  // wave by wave (eta_p.hpp: eta_p_w4_kernel; C5 105 -> 88 us); bitwise the same results
  if (static_cast<long long>(a.nb_p + nb_i) * c->launch_slots > 2LL * c->n_cus) {
    const int per4 = kBlock / group_lanes(c->code_l);
    const int nb_i4 = (c->n_items + per4 - 1) / per4;
#define CALL(G, V) LAUNCH_IN(ls, (eta_p_w4_kernel<G, V>), slot_grid(c, a.nb_p + nb_i4), kBlock, 0, c->stream, a)
    DISPATCH_GV(c->code_l, CALL);
#undef CALL
    ls.done();
    return;
  }
#define CALL(G, V) LAUNCH_IN(ls, (eta_p_kernel<G, V>), slot_grid(c, a.nb_p + nb_i), kRedThreads, 0, c->stream, a)
  DISPATCH_GV(c->code_l, CALL);
#undef CALL
  ls.done();
}

}  // namespace mmsbm_hip_impl

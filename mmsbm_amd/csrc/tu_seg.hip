// tu_seg.hip -- translation unit of the two triple passes: seg_pass_kernel, seg_pass_slots_kernel, seg_wide_kernel and
// the combine kernels of split segments (seg_pass.hpp), with the host code that chooses among them
#include "prelude.hpp"
#include "seg_pass.hpp"

namespace mmsbm_hip_impl {

// The two triple passes.  with_pairs / with_users select the segment sets of this launch (both: one
// launch, the pair segments' workgroups first); `st` is the stream it goes to.
void stage_seg(mmsbm_hip_ctx *c, bool commit, bool with_pairs, bool with_users, hipStream_t st) {
  const SegArgs sp = seg_pairs_args(c), su = seg_users_args(c, commit, c->n_users);
  const int per = kBlock / group_lanes(c->code_k);
  if (c->kp > kMaxGroupRow) {  // rows of more than 1,024 groups: a wave per segment, whole segments (no work lists)
    LaunchScope ls(c, K_SEG, st == c->stream);
    const int bp = with_pairs ? (sp.nseg + per - 1) / per : 0, bu = with_users ? (su.nseg + per - 1) / per : 0;
    if (bp + bu > 0) {
      // up to 2,048: the group-of-lanes kernel once more, 32 doubles per lane, one row in flight per wave (the row,
      // the fixed row and the sums are 192 registers); beyond: seg_wide_kernel, the row in blocks of 1,024 columns
      if (c->kp <= 2 * kMaxGroupRow) LAUNCH_IN(ls, (seg_pass_kernel<64, 32, 1>), slot_grid(c, bp + bu), kBlock, 0, st, sp, su, bp, c->kp);
      else LAUNCH_IN(ls, (seg_wide_kernel<16>), slot_grid(c, bp + bu), kBlock, 0, st, sp, su, bp, c->kp);
    }
    ls.done();
    return;
  }
  // several restart slots: a super-group of SW x G lanes per segment (seg_pass_slots_kernel)
  int sw = 1;
  if (c->launch_slots > 1) {
    const int room = 64 / group_lanes(c->code_k);
    while (sw * 2 <= room && sw < c->launch_slots) sw *= 2;
  }
  // (the stage is one kernel -- and is timed as that kernel -- when no segment was split and the slots share a launch)
  const bool one_kernel = sw == 1 && c->lay.pair_work.splits.empty() && c->lay.user_work.splits.empty() && st == c->stream &&
                          (with_pairs ? sp.nseg : 0) + (with_users ? su.nseg : 0) > 0;
  LaunchScope ls(c, K_SEG, one_kernel);
  if (sw > 1) {
    const int per_s = kBlock / (group_lanes(c->code_k) * sw);
    const int bps = with_pairs ? (sp.nseg + per_s - 1) / per_s : 0;
    const int bus = with_users ? (su.nseg + per_s - 1) / per_s : 0;
    const dim3 grid(static_cast<unsigned>(bps + bus), static_cast<unsigned>((c->launch_slots + sw - 1) / sw), 1);
    if (bps + bus > 0) {
#define CALL_S(G, V, S) LAUNCH((seg_pass_slots_kernel<G, V, 4, S>), grid, kBlock, 0, st, sp, su, bps, c->kp, c->launch_slots)
      switch (c->code_k * 100 + sw) {
        case 2: CALL_S(4, 4, 2); break;
        case 4: CALL_S(4, 4, 4); break;
        case 8: CALL_S(4, 4, 8); break;
        case 16: CALL_S(4, 4, 16); break;
        case 102: CALL_S(8, 4, 2); break;
        case 104: CALL_S(8, 4, 4); break;
        case 108: CALL_S(8, 4, 8); break;
        case 202: CALL_S(16, 4, 2); break;
        case 204: CALL_S(16, 4, 4); break;
        case 302: CALL_S(32, 4, 2); break;
        default: throw ApiError(MMSBM_E_INTERNAL, "seg_pass_slots: no instantiation");
      }
#undef CALL_S
    }
  }
  const int bp = (with_pairs && sw == 1) ? (sp.nseg + per - 1) / per : 0;
  const int bu = (with_users && sw == 1) ? (su.nseg + per - 1) / per : 0;
  if (bp + bu > 0) {  // one slot per workgroup (blockIdx.y = slot)
#define SEG_GO(G, V, B) LAUNCH_IN(ls, (seg_pass_kernel<G, V, B>), slot_grid(c, bp + bu), kBlock, 0, st, sp, su, bp, c->kp)
    if (c->seg_batch == 8 && c->code_k <= 4) {  // (eight row gathers in flight per group: small problems; not with
                                                  // 8 or 16 doubles per lane and row: that is 128 - 256 registers)
      switch (c->code_k) {
        case 0: SEG_GO(4, 4, 8); break;
        case 1: SEG_GO(8, 4, 8); break;
        case 2: SEG_GO(16, 4, 8); break;
        case 3: SEG_GO(32, 4, 8); break;
        default: SEG_GO(64, 4, 8); break;
      }
    } else {
#define CALL(G, V) SEG_GO(G, V, 4)
      DISPATCH_GV(c->code_k, CALL);
#undef CALL
    }
#undef SEG_GO
  }
  // long segments were processed in pieces: add the pieces up (fixed order) and finish them
  // (splits with few pieces come first in the lists: one group of lanes each; the rest: a workgroup each)
  const mmsbm::WorkList &wp = c->lay.pair_work, &wu = c->lay.user_work;
  const int nsp_s = with_pairs ? wp.n_small : 0;
  const int nsp_b = with_pairs ? static_cast<int>(wp.splits.size()) - wp.n_small : 0;
  const int nsu_s = with_users ? wu.n_small : 0;
  const int nsu_b = with_users ? static_cast<int>(wu.splits.size()) - wu.n_small : 0;
  const CombineArgs cps{c->pair_splits.ptr, sp.parts, c->pair_off.ptr, sp.fixed, sp.out, nsp_s, sp.mode, sp.bs_parts};
  const CombineArgs cus{c->user_splits.ptr, su.parts, c->user_off.ptr, su.fixed, su.out, nsu_s, su.mode, su.bs_parts};
  const CombineArgs cpb{c->pair_splits.ptr + wp.n_small, sp.parts, c->pair_off.ptr, sp.fixed, sp.out, nsp_b, sp.mode, sp.bs_parts};
  const CombineArgs cub{c->user_splits.ptr + wu.n_small, su.parts, c->user_off.ptr, su.fixed, su.out, nsu_b, su.mode, su.bs_parts};
  const int ba = (nsp_s + per - 1) / per, bb = (nsu_s + per - 1) / per;
  const size_t lds = static_cast<size_t>(per) * c->kp * sizeof(double);
  if (nsp_b + nsu_b > 0) {  // some split has many pieces: one launch for both kinds (the small ones' blocks first, if any)
#define CALL(G, V) \
  LAUNCH((seg_combine_both_kernel<G, V>), slot_grid(c, ba + bb + nsp_b + nsu_b), kBlock, lds, st, cps, cus, ba, ba + bb, cpb, cub, nsp_b, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  } else if (nsp_s + nsu_s > 0) {
#define CALL(G, V) LAUNCH((seg_combine_small_kernel<G, V>), slot_grid(c, ba + bb), kBlock, 0, st, cps, cus, ba, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  }
  ls.done();
}

}  // namespace mmsbm_hip_impl

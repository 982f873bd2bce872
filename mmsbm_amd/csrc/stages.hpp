// stages.hpp -- host side of the iteration: argument blocks, kernel selection and the four stage launchers
// Part of the single translation unit mmsbm_hip.hip (included there, in order; not a stand-alone header).
#pragma once

namespace {

// One kernel launch of a stage: with the stage's event pair attached to the launch when the stage is timed as
// that kernel (LaunchScope::ext), else plain.  KERNEL in parentheses when its template arguments hold commas.
#define LAUNCH_IN(LS, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                  \
  do {                                                                                                       \
    if ((LS).ext())                                                                                          \
      hipExtLaunchKernelGGL(KERNEL, GRID, dim3(BLOCK), static_cast<uint32_t>(LDS), STREAM, (LS).e0, (LS).e1, 0, __VA_ARGS__); \
    else                                                                                                     \
      hipLaunchKernelGGL(KERNEL, GRID, dim3(BLOCK), static_cast<uint32_t>(LDS), STREAM, __VA_ARGS__);        \
  } while (0)

constexpr size_t kLdsMax = 160 * 1024;  // with hipFuncAttributeMaxDynamicSharedMemorySize

// dynamic LDS of pair_block: transposed rows + output rows (shared with the eta rows)
constexpr size_t kScalarTileBytes = 8 * 1024;  // larger tiles thrash the scalar cache: stage in LDS
bool tile_in_lds(int dinp, int doutp) {
  return static_cast<size_t>(dinp) * doutp * sizeof(double) > kScalarTileBytes;
}
size_t pair_block_lds(int dinp, int doutp, bool tile_lds) {
  // transposed input rows + one region shared by the eta rows and the output rows [+ the tile]
  const size_t d = static_cast<size_t>(dinp) * (kUnitPairs + 1) + static_cast<size_t>(kUnitPairs) * doutp +
                   (tile_lds ? static_cast<size_t>(dinp) * doutp : 0);
  return d * sizeof(double);  // the S hand-over area reuses it (create() bounds the copies by it)
}

template <class K>
void allow_big_lds(K kernel, size_t bytes) {
  if (bytes > kLdsBudget)
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(bytes)));
}

// ---- the stages of one EM iteration ---------------------------------------------------------
// commit: parameters advance (theta, eta, p normalised, A refreshed); otherwise the
// un-normalised numerators are left in the "next" buffers / npr.
// split (main/tail) tables -- see RowTab: theta and A are the gathered ones, eta/C/T stream
RowTab plain_tab(double *base, int width, size_t slot_stride = 0) {
  return RowTab{base, base, width, 0, width, 0, slot_stride, 0};
}
// A gathered table (theta, A) as seen from restart slot `slot`: the n_slots copies of every row are
// interleaved (RowTab), main parts of all rows first, then the tail parts.
RowTab gather_tab(const mmsbm_hip_ctx *c, double *base, size_t rows, int slot) {
  int mw = c->split_rows ? (c->kp / 16) * 16 : c->kp;  // (split_rows is always on today)
  if (mw == 0) mw = c->kp;
  const int tw = c->kp - mw, ns = c->n_slots;
  return RowTab{base + static_cast<size_t>(slot) * mw,
                base + rows * static_cast<size_t>(ns) * mw + static_cast<size_t>(slot) * tw,
                mw, tw, ns * mw, ns * tw, static_cast<size_t>(mw), static_cast<size_t>(tw)};
}
// (`b` = which of the two ping-pong buffers; the restart slot is c->base_slot)
RowTab theta_tab(const mmsbm_hip_ctx *c, int b) {
  return gather_tab(c, c->theta[b].ptr, static_cast<size_t>(c->n_users), c->base_slot);
}
RowTab a_tab(const mmsbm_hip_ctx *c, int b) {
  return gather_tab(c, c->atab[b].ptr, static_cast<size_t>(c->n_pairs), c->base_slot);
}
dim3 slot_grid(const mmsbm_hip_ctx *c, int blocks) {
  return dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(c->launch_slots), 1);
}
// Single-restart entry points: launches and copies cover the selected slot only.
struct OneSlot {
  mmsbm_hip_ctx *c;
  int b, n;
  explicit OneSlot(mmsbm_hip_ctx *ctx) : c(ctx), b(ctx->base_slot), n(ctx->launch_slots) {
    c->base_slot = c->sel;
    c->launch_slots = 1;
  }
  ~OneSlot() { c->base_slot = b; c->launch_slots = n; }
};

// Non-temporal stores for the rows the next launch gathers (T, A, theta'): one restart per launch and rows of up to
// 32 groups.  Measured per iteration, plain -> non-temporal (scripts/ab_fused.sh, variants side by side on one box): C1
// 11.3 -> 11.2 us, C2 20.3 -> 19.5, 600k ratings at K = L = 20 67.2 -> 66.0, C3 95.3 -> 93.8, 3M 302.3 -> 297.2, 10M
// 886 -> 880; with restart slots nothing or a loss (C3 x 2 165.8 -> 166.9, C3 x 8 591 -> 597), theta' at K = L = 50
// +0.3 %.  C rows, eta' and the slabs the same way: +0.3, +0.3 and +1.0 us at C3 -- they stay plain stores.  Loads:
// the segments' own rows in seg_pass as non-temporal loads C3 94.1 -> 93.5 (kept, same condition); the T rows in
// item_sum +1.6 us, the index stream +0.3, C rows in T + S and the slabs in p_update nothing.
// Only for data whose segments are all short and alike (no work lists on either side): with heavy-tailed degrees the
// rows of busy users and popular items are gathered again and again and a long segment's own row is read by every one
// of its pieces -- there every one of these hints costs (1M ratings, K = L = 20, per iteration, none / T and A rows /
// theta' rows / own-row loads / all: log-normal degrees 90.2 / 90.6 / 91.5 / 95.9 / 96.8 us, Zipf(1.2) 93.9 / 95.5 /
// 99.2 / 103.6 / 104.7; 100k ratings of 943 users: 28.2 / 29.2 / 28.2 / 28.4 / 29.4; scripts/nt_time.py).
int nt_on(const mmsbm_hip_ctx *c) {
  const bool plain_data = (c->lay.pair_work.items.empty() && c->lay.user_work.items.empty()) || (c->nt_out & 8);  // (8: tuning, whatever the data)
  return (plain_data && c->launch_slots == 1 && c->kp <= 32 && c->lp <= 32) ? (c->nt_out & 7) : 0;
}

SegArgs seg_pairs_args(const mmsbm_hip_ctx *c) {  // C = sum over a pair's triples
  const bool it = !c->lay.pair_work.items.empty();
  return SegArgs{a_tab(c, c->cur), theta_tab(c, c->cur), c->pair_off.ptr, c->pair_user.ptr,
                 plain_tab(c->ctab.at(c->base_slot), c->kp, c->ctab.stride),
                 it ? static_cast<int32_t>(c->lay.pair_work.items.size()) : c->n_pairs, 0,
                 it ? c->pair_items.ptr : nullptr, c->pair_parts.at(c->base_slot), c->pair_parts.stride,
                 nt_on(c) >> 1};
}
SegArgs seg_users_args(const mmsbm_hip_ctx *c, bool commit, int seg_end) {  // theta_new
  const bool it = !c->lay.user_work.items.empty();
  return SegArgs{theta_tab(c, c->cur),     a_tab(c, c->cur), c->user_off.ptr, c->user_pair.ptr,
                 theta_tab(c, c->cur ^ 1),
                 it ? static_cast<int32_t>(c->lay.user_work.items.size()) : seg_end,
                 commit ? 1 : 2,
                 it ? c->user_items.ptr : nullptr, c->user_parts.at(c->base_slot), c->user_parts.stride,
                 nt_on(c) >> 1};
}
PairBlockArgs pair_block_t_args(const mmsbm_hip_ctx *c) {
  const int s = c->base_slot;
  PairBlockArgs pa{};
  pa.tiles = c->p[c->cur].at(s); pa.in_tab = c->ctab.at(s); pa.e_tab = c->eta[c->cur].at(s);
  pa.pair_item = c->pair_item.ptr; pa.chunks = c->mv_chunks.ptr;
  pa.out = c->ttab.at(s); pa.partial = c->partial.at(s);
  pa.din = c->k; pa.dinp = c->kp; pa.doutp = c->lp; pa.spb = c->pb_spb; pa.nsub = c->pb_nsub; pa.abl = c->ablate;
  pa.nt = nt_on(c) & 1;
  pa.out_mw = c->lp; pa.out_rs_m = c->lp; pa.out_rs_t = 0; pa.out_tail = c->ttab.at(s);
  pa.bs_tiles = c->p[0].stride; pa.bs_in = c->ctab.stride; pa.bs_e = c->eta[0].stride;
  pa.bs_out = c->ttab.stride; pa.bs_out_t = 0; pa.bs_partial = c->partial.stride;
  pa.mg0 = pa.mg1 = 0;  // (the T + S launch runs padded 16-tiles: its 128 registers do not hold the 4 x 4 blocks)
  return pa;
}
PairBlockArgs pair_block_a_args(const mmsbm_hip_ctx *c, int param_slot, int a_slot) {
  const int s = c->base_slot;  // (param_slot / a_slot: ping-pong buffer indices)
  const RowTab at = a_tab(c, a_slot);
  PairBlockArgs pa{};
  pa.tiles = c->pt[param_slot].at(s); pa.in_tab = c->eta[param_slot].at(s); pa.e_tab = nullptr;
  pa.pair_item = c->pair_item.ptr; pa.chunks = c->mv_chunks.ptr;
  pa.out = at.main; pa.partial = nullptr;
  pa.din = c->l; pa.dinp = c->lp; pa.doutp = c->kp; pa.spb = kBlock; pa.nsub = 1; pa.abl = c->ablate;
  pa.nt = (nt_on(c) & 1) | (((c->nt_out & 1) && c->launch_slots == 1 && c->lay.pair_work.items.empty() && c->lay.user_work.items.empty()) ? 2 : 0);  // bit 1: the matrix-core A rows (C5 2,236 -> 2,215 us; T rows there: nothing)
  pa.out_mw = at.mw; pa.out_rs_m = at.rs_m; pa.out_rs_t = at.rs_t; pa.out_tail = at.tail;
  pa.bs_tiles = c->pt[0].stride; pa.bs_in = c->eta[0].stride; pa.bs_e = 0;
  pa.bs_out = at.so_m; pa.bs_out_t = at.so_t; pa.bs_partial = 0;
  mfma_geometry(pa.dinp, pa.doutp, &pa.mg0, &pa.mg1);
  return pa;
}
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
  a.abl = c->ablate;
  a.nb_p = (c->kp * c->lp + cols_per_block - 1) / cols_per_block;
  a.item_grid = c->item_grid.count ? c->item_grid.ptr : nullptr;
  return a;
}

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
#define CALL_S(G, V, S) \
  seg_pass_slots_kernel<G, V, 4, S><<<grid, kBlock, 0, st>>>(sp, su, bps, c->kp, c->launch_slots)
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
#define CALL(G, V) SEG_GO(G, V, 8)
      DISPATCH_GV(c->code_k, CALL);
#undef CALL
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
  if (nsp_s + nsu_s > 0 && nsp_b + nsu_b > 0) {  // both kinds: one launch
#define CALL(G, V) \
  seg_combine_both_kernel<G, V><<<slot_grid(c, ba + bb + nsp_b + nsu_b), kBlock, lds, st>>>(cps, cus, ba, ba + bb, cpb, cub, nsp_b, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  } else if (nsp_s + nsu_s > 0) {
#define CALL(G, V) \
  seg_combine_small_kernel<G, V><<<slot_grid(c, ba + bb), kBlock, 0, st>>>(cps, cus, ba, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  } else if (nsp_b + nsu_b > 0) {
#define CALL(G, V) \
  seg_combine_kernel<G, V><<<slot_grid(c, nsp_b + nsu_b), kBlock, lds, st>>>(cpb, cub, nsp_b, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  }
  ls.done();
}

// Units per workgroup for a launch whose workgroups each walk a run of 64-pair units of one rating, `slots` of them
// resident at a time: the launch lasts rounds x (units + a prologue of ~0.6 unit-times: the tile into LDS, the item ids,
// the first rows' latency -- measured at C5, EXPERIMENTS.md), and rounds is an INTEGER (with 768 slots C5's 15,616 units in
// runs of 8 are 1,952 workgroups = 2.54 rounds, paid as 3 = 24 unit-times; in runs of 11 they are 1.9 rounds, paid as 2 = 22).
// Every rating's run count is rounded up to a multiple of 8 as in build_mv_chunks.
inline int balanced_run_units(const std::vector<int32_t> &rating_off, int slots, int lo, int hi) {
  int best = hi;
  double best_cost = 1e300;
  for (int u = hi; u >= lo; --u) {
    long long wgs = 0;
    for (size_t r = 0; r + 1 < rating_off.size(); ++r) {
      const long long units = (rating_off[r + 1] - rating_off[r] + kUnitPairs - 1) / kUnitPairs;
      const long long runs = (units + u - 1) / u;
      wgs += rating_off.size() > 2 ? (runs + 7) / 8 * 8 : runs;
    }
    const double cost = static_cast<double>((wgs + slots - 1) / std::max(slots, 1)) * (u + 0.6);
    if (cost < best_cost * 0.97) { best_cost = cost; best = u; }   // (longer runs win ties: fewer prologues)
  }
  return best;
}

bool mfma_possible(const mmsbm_hip_ctx *c) {
  return !c->wide && c->kp <= kMfmaMaxDim && c->lp <= kMfmaMaxDim && c->lds_mt <= kLdsMax && c->lds_ma <= kLdsMax;
}

void stage_dense(mmsbm_hip_ctx *c) {  // T = P^T C  and the K x L slabs for p
  if (c->n_chunks == 0) return;
  if (c->mfma_big) {
    LaunchScope ls(c, K_DENSE);
    const int nb = static_cast<int>(c->lay.mv_chunks.size());
    const PairBlockArgs pa = pair_block_t_args(c);
    const int subs = (c->mv_chunk_pairs + kRowsUnits * kUnitPairs - 1) / (kRowsUnits * kUnitPairs);  // groups of units per chunk
    const int n_kb = (c->kp + kMfmaBlk - 1) / kMfmaBlk, n_lb = (c->lp + kMfmaBlk - 1) / kMfmaBlk;
    allow_big_lds(mfma_rows_kernel<false>, kMfmaRowsLds);
    allow_big_lds(mfma_slab_kernel, kMfmaSlabLds);
    mfma_rows_kernel<false><<<slot_grid(c, nb * subs * n_lb), kPairBlockMax, kMfmaRowsLds, c->stream>>>(pa, pa.tiles, subs, n_lb);
    mfma_slab_kernel<<<slot_grid(c, nb * n_kb * n_lb), kPairBlockMax, kMfmaSlabLds, c->stream>>>(pa, n_kb, n_lb);
    ls.done();
    return;
  }
  if (c->wide) {
    LaunchScope ls(c, K_DENSE);
    const int nb = static_cast<int>(c->lay.mv_chunks.size());
    const PairBlockArgs pa = pair_block_t_args(c);
    const int subs = kWideChunkPairs / kWidePairs;
    const int kgs = (c->kp + kWideKG - 1) / kWideKG, lbs = (c->lp + kBlock - 1) / kBlock;
    const size_t lds = static_cast<size_t>(kWidePairs) * c->kp * sizeof(double);
    allow_big_lds(wide_matvec_kernel<false>, lds);
    wide_matvec_kernel<false><<<slot_grid(c, nb * subs), kBlock, lds, c->stream>>>(pa, subs);
    wide_slab_kernel<<<slot_grid(c, nb * kgs * lbs), kBlock, 0, c->stream>>>(pa, kgs, lbs);
    ls.done();
    return;
  }
  if (c->mfma) {
    LaunchScope ls(c, K_DENSE, true);
    const int nb = static_cast<int>(c->lay.mv_chunks.size());
    const PairBlockArgs pa = pair_block_t_args(c);
    // (padded 16-tiles, eight waves: the 4 x 4 blocks of the A launch do not fit this launch's 128 registers -- measured,
    // EXPERIMENTS.md -- and four waves leave one workgroup per CU)
    allow_big_lds(pair_mfma_kernel<false, true, kPairBlockMax, false>, c->lds_mt);
    LAUNCH_IN(ls, (pair_mfma_kernel<false, true, kPairBlockMax, false>), slot_grid(c, nb), kPairBlockMax, c->lds_mt, c->stream, pa, pa.tiles);
    ls.done();
    return;
  }
  {
    LaunchScope ls(c, K_DENSE, true);
    const int nb = static_cast<int>(c->lay.mv_chunks.size());
    const PairBlockArgs pa = pair_block_t_args(c);
#define PB_D(N, TL, NT, KT, D)                                                              \
  do {                                                                                      \
    allow_big_lds(pair_block_kernel<false, true, N, TL, NT, KT, D>, c->lds_t);              \
    LAUNCH_IN(ls, (pair_block_kernel<false, true, N, TL, NT, KT, D>), slot_grid(c, nb), NT, c->lds_t, c->stream, pa, pa.tiles); \
  } while (0)
#define PB_KT(N, TL, NT, KT)                                                                \
  do {                                                                                      \
    if (c->direct_out) PB_D(N, TL, NT, KT, true); else PB_D(N, TL, NT, KT, false);          \
  } while (0)
#define PB_GO(N, TL, NT)                                                                    \
  do {                                                                                      \
    if (c->pb_kt == 2) PB_KT(N, TL, NT, 2); else PB_KT(N, TL, NT, 4);                        \
  } while (0)
#define PB(N)                                                                               \
  do {                                                                                      \
    const bool tl = c->tl_t, big = c->pb_threads_t > kBlock;                                \
    if (tl && big) PB_GO(N, true, kPairBlockMax);                                           \
    else if (tl) PB_GO(N, true, kBlock);                                                    \
    else if (big) PB_GO(N, false, kPairBlockMax);                                           \
    else PB_GO(N, false, kBlock);                                                           \
  } while (0)
    switch (c->pb_nacc) {
      case 1: PB(1); break;
      case 2: PB(2); break;
      default: PB(4); break;
    }
#undef PB
#undef PB_GO
#undef PB_KT
#undef PB_D
    ls.done();
  }
}

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

// A[q,:] from (eta, pT) of parameter slot `slot` into atab[a_slot] -- or, with `grid` set, the same
// mat-vec over every (item, rating) combination into the plain table btab (prod_dist / predict)
void stage_matvec_a(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid = false) {
  int nb = grid ? c->grid_n_chunks : static_cast<int>(c->lay.mv_chunks.size());
  if (nb == 0) return;
  LaunchScope ls(c, K_MATVEC_A, true);
  PairBlockArgs pa = pair_block_a_args(c, slot, a_slot);
  if (!grid && c->mfma && c->n_a_chunks > 0) {  // the same units in runs of its own (create(): balanced_run_units)
    pa.chunks = c->a_chunks.ptr;
    nb = c->n_a_chunks;
  }
  if (grid) {
    pa.pair_item = c->grid_item.ptr; pa.chunks = c->grid_chunks.ptr;
    pa.out = c->btab.ptr; pa.out_tail = nullptr;
    pa.out_mw = pa.doutp; pa.out_rs_m = pa.doutp; pa.out_rs_t = 0; pa.bs_out = 0; pa.bs_out_t = 0;
  }
  if (c->mfma_big) {
    const int subs = (c->mv_chunk_pairs + kRowsUnits * kUnitPairs - 1) / (kRowsUnits * kUnitPairs);
    const int n_kb = (c->kp + kMfmaBlk - 1) / kMfmaBlk;  // (outputs: K columns)
    allow_big_lds(mfma_rows_kernel<true>, kMfmaRowsLds);
    LAUNCH_IN(ls, (mfma_rows_kernel<true>), slot_grid(c, nb * subs * n_kb), kPairBlockMax, kMfmaRowsLds, c->stream, pa, pa.tiles, subs, n_kb);
  } else if (c->wide) {
    const int subs = kWideChunkPairs / kWidePairs;
    const size_t lds = static_cast<size_t>(kWidePairs) * c->lp * sizeof(double);
    allow_big_lds(wide_matvec_kernel<true>, lds);
    LAUNCH_IN(ls, (wide_matvec_kernel<true>), slot_grid(c, nb * subs), kBlock, lds, c->stream, pa, subs);
  } else if (c->mfma) {
    // (eight waves like the T + S launch: 84 registers against the four-wave form's 148, so that two to three
    // workgroups = 16 to 24 waves share a CU instead of 12 -- C5 150 -> 139-146 us, 4M ratings at K = L = 50 unchanged)
    allow_big_lds(pair_mfma_kernel<true, false, kPairBlockMax, true>, c->lds_ma);
    LAUNCH_IN(ls, (pair_mfma_kernel<true, false, kPairBlockMax, true>), slot_grid(c, nb), kPairBlockMax, c->lds_ma, c->stream, pa, pa.tiles);
  } else if (c->quad_a) {
    const dim3 grid = slot_grid(c, std::min(nb, c->n_cus));
#define QA(NL)                                                                                    \
  do {                                                                                            \
    allow_big_lds(pair_quad_a_kernel<NL>, c->lds_qa);                                             \
    LAUNCH_IN(ls, (pair_quad_a_kernel<NL>), grid, kPairBlockMax, c->lds_qa, c->stream, pa, pa.tiles, nb); \
  } while (0)
    const int nl = (c->lp + 3) / 4;  // dinp of the A launch = lp
    if (nl <= 8) QA(8); else if (nl <= 10) QA(10); else if (nl <= 12) QA(12);
    else if (nl <= 13) QA(13); else QA(14);   // (kQuadMaxL: 15 or 16 double2 per thread spill at 256 registers)
#undef QA
  } else {
#define PA_D(TL, NT, D)                                                                     \
  do {                                                                                      \
    allow_big_lds(pair_block_kernel<true, false, 1, TL, NT, 4, D>, c->lds_a);               \
    LAUNCH_IN(ls, (pair_block_kernel<true, false, 1, TL, NT, 4, D>), slot_grid(c, nb), NT, c->lds_a, c->stream, pa, pa.tiles); \
  } while (0)
#define PA_GO(TL, NT)                                                                       \
  do {                                                                                      \
    if (c->direct_out) PA_D(TL, NT, true); else PA_D(TL, NT, false);                        \
  } while (0)
    const bool tl = c->tl_a, big = c->pb_threads_a > kBlock;
    if (tl && big) PA_GO(true, kPairBlockMax);
    else if (tl) PA_GO(true, kBlock);
    else if (big) PA_GO(false, kPairBlockMax);
    else PA_GO(false, kBlock);
#undef PA_GO
#undef PA_D
  }
  ls.done();
}

// ---- small problems: the iteration in two launches (fused_small.hpp) -------------------------------------
constexpr size_t kFusedSplitLds = 96 * 1024;  // partial rows of a workgroup's split user segments, at most
bool fused_shape_ok(const mmsbm_hip_ctx *c) {  // (everything but the data: the kernels exist for this shape)
  // (rows of up to 24 groups: beyond that the four-launch pair stage runs 512-thread workgroups, whose split of a
  // unit's pairs among the copies of the slab grid -- hence the association order of S -- 256 threads cannot mirror)
  return c->code_k <= 1 && c->code_l <= 1 && c->pb_threads_t == kBlock && c->pb_threads_a == kBlock && !c->tl_t &&
         !c->tl_a && c->pb_nacc == 1 && c->pb_kt == 2 && c->pb_spb * c->pb_nsub <= kBlock && !c->wide &&
         !c->mfma && !c->mfma_big && !c->direct_out && c->mv_chunk_pairs == mmsbm::kMvChunkPairs && c->n_chunks > 0 &&
         pairs_fused_lds(c->kp, c->lp) <= kLdsBudget;
}
bool fused_possible(const mmsbm_hip_ctx *c) {
  // A work list that only ORDERS whole segments is fine (the pair units ignore it -- every segment's result is its
  // own -- and the user pass follows it as before); segments cut into pieces need all pieces of a segment inside one
  // workgroup: the lists create() builds for small problems (fs_pairs / fs_users; round 4)
  return fused_shape_ok(c) && (c->lay.pair_work.splits.empty() || c->fs_pairs) &&
         (c->lay.user_work.splits.empty() || c->fs_users);
}
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

// atab[cur] = A of the current parameters, for every slot the next launches cover (see mmsbm_hip_ctx::a_ok)
void ensure_a(mmsbm_hip_ctx *c) {
  bool ok = true;
  for (int s = c->base_slot; s < c->base_slot + c->launch_slots; ++s) ok = ok && c->a_ok[static_cast<size_t>(s)];
  if (ok) return;
  const bool prof = c->profiling;  // (not a launch of the iteration being profiled)
  c->profiling = false;
  stage_matvec_a(c, c->cur, c->cur);
  c->profiling = prof;
  for (int s = c->base_slot; s < c->base_slot + c->launch_slots; ++s) c->a_ok[static_cast<size_t>(s)] = 1;
}
void mark_a(mmsbm_hip_ctx *c, bool ok) {
  for (int s = c->base_slot; s < c->base_slot + c->launch_slots; ++s) c->a_ok[static_cast<size_t>(s)] = ok ? 1 : 0;
}

// Two launches while the launches' work is small: ratings x restart slots x (K + L, padded) <= 14M (measured per
// iteration, two / four launches.  One slot, K = L = 10: 100k ratings 20.2 / 29.3 us, 300k 28.7 / 37.0, 500k 37.8 /
// 43.5; K = L = 16: 400k 36.0 / 41.7; K = L = 20: 100k 28.4 / 37.6, 300k 50.1 / 50.1, 600k 69.4 / 67.5, 1M 105.0 / 95.6.  100k ratings at K = L = 10
// with 2 / 4 / 8 / 16 slots: 26.5 / 32.4, 40.9 / 42.8, 68.0 / 64.9, 119.9 / 115.9 -- with several slots the four-launch
// form shares the index stream among them)
// Data whose segments are cut into pieces pays a combine launch in the separate form (five launches): there the two
// launches win a little further out (log-normal popularity, separate / two, scripts/fused_split_sweep.py: 400k ratings
// K = L = 10 38.4 / 31.3 us, 700k 43.8 / 39.8; 300k K = L = 20 45.1 / 41.4, 600k 54.0 / 65.5; 1M x 100k x 20k K = L = 10
// 77.3 / 92.6) -- up to 18M.
constexpr long long kFusedWorkMax = 14000000, kFusedWorkMaxSplit = 18000000, kFusedRatingsMax = 1500000;
bool use_fused(const mmsbm_hip_ctx *c) {
  // (fused_possible again: options set after create() -- "mfma", "direct" ... -- change what it depends on)
  const bool cut = !c->lay.pair_work.splits.empty() || !c->lay.user_work.splits.empty();
  return c->fused && fused_possible(c) &&
         (c->fused_forced || c->n_obs * c->launch_slots * (c->kp + c->lp) <= (cut ? kFusedWorkMaxSplit : kFusedWorkMax));
}

void launch_iteration(mmsbm_hip_ctx *c, bool commit) {
  if (use_fused(c)) {
    stage_fused_pairs(c);  // (writes A of the current parameters on its way)
    stage_fused_tail(c, commit);
    if (commit) c->cur ^= 1;
    mark_a(c, !commit);
    return;
  }
  ensure_a(c);
  stage_seg(c, commit, true, true, c->stream);
  stage_dense(c);
  stage_eta_p(c, commit);
  if (commit) {
    stage_matvec_a(c, c->cur ^ 1, c->cur ^ 1);
    c->cur ^= 1;
    mark_a(c, true);
  }
}

// n committed iterations: graph replays of two iterations each when enabled, the rest eager
void run_iterations(mmsbm_hip_ctx *c, int n) {
  if (c->graph_mode && !c->profiling) {
    while (n >= 2) {
      const int slot = c->cur;
      if (!c->graph_exec[slot]) {
        if (!use_fused(c)) ensure_a(c);  // (outside the capture: a replay must not repeat it)
        hipGraph_t graph = nullptr;
        HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        try {
          launch_iteration(c, true);
          launch_iteration(c, true);
        } catch (...) {
          (void)hipStreamEndCapture(c->stream, &graph);
          if (graph) (void)hipGraphDestroy(graph);
          c->cur = slot;
          throw;
        }
        HIP_CHECK(hipStreamEndCapture(c->stream, &graph));
        hipError_t e = hipGraphInstantiate(&c->graph_exec[slot], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess)
          throw ApiError(MMSBM_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        // capture only records: cur is back where it started and nothing has run yet
      }
      HIP_CHECK(hipGraphLaunch(c->graph_exec[slot], c->stream));
      // (a replay runs none of launch_iteration's host code: the bookkeeping of "atab[cur] holds A of the current
      // parameters" has to be repeated here -- true after the four-launch form, false after the two-launch one)
      mark_a(c, !use_fused(c));
      n -= 2;
    }
  }
  for (; n > 0; --n) launch_iteration(c, true);
}

void require_params(const mmsbm_hip_ctx *c) {  // the selected slot
  if (!c) throw std::invalid_argument("null context");
  if (!c->have[c->sel]) throw std::invalid_argument("set_params has not been called");
}
void require_all_params(const mmsbm_hip_ctx *c) {  // every slot: the iteration advances all of them
  if (!c) throw std::invalid_argument("null context");
  for (int s = 0; s < c->n_slots; ++s)
    if (!c->have[s])
      throw std::invalid_argument(c->n_slots == 1 ? std::string("set_params has not been called")
                                                  : "set_params has not been called for slot " +
                                                        std::to_string(s));
}

// (Re)allocate the per-restart state for `slots` parameter sets; nothing is kept.
void alloc_state(mmsbm_hip_ctx *c, int slots) {
  hipStream_t s = c->stream;
  HIP_CHECK(hipStreamSynchronize(s));
  c->drop_graphs();
  const size_t klr = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  auto zeroed = [&](SlotBuf &b, size_t per_slot) {
    b.alloc_slots(per_slot, slots);
    HIP_CHECK(hipMemsetAsync(b.ptr, 0, sizeof(double) * std::max<size_t>(b.count, 1), s));
  };
  for (int b = 0; b < 2; ++b) {
    zeroed(c->theta[b], static_cast<size_t>(c->n_users) * c->kp);
    zeroed(c->eta[b], static_cast<size_t>(c->n_items) * c->lp);
    zeroed(c->p[b], klr);
    zeroed(c->pt[b], klr);
    zeroed(c->atab[b], static_cast<size_t>(c->n_pairs) * c->kp);
  }
  // (the scratch tables too: every entry a kernel reads is written by the launch before it -- audited, EXPERIMENTS.md
  // -- but padding rows and the slabs of empty units then hold zeros rather than whatever the pages held before)
  zeroed(c->ctab, static_cast<size_t>(c->n_pairs) * c->kp);
  zeroed(c->ttab, static_cast<size_t>(c->n_pairs) * c->lp);
  zeroed(c->partial, c->lay.mv_chunks.size() * c->kp * c->lp);
  zeroed(c->npr, klr);
  zeroed(c->pair_parts, static_cast<size_t>(c->lay.pair_work.n_parts) * c->kp);
  zeroed(c->user_parts, static_cast<size_t>(c->lay.user_work.n_parts) * c->kp);
  HIP_CHECK(hipStreamSynchronize(s));
  c->n_slots = slots;
  c->sel = 0;
  c->base_slot = 0;
  c->launch_slots = slots;
  c->cur = 0;
  c->have.assign(static_cast<size_t>(slots), 0);
  c->a_ok.assign(static_cast<size_t>(slots), 0);
}

// host (rows, d) row-major  <->  device RowTab (rows, dp) zero-padded, main + tail parts, staged
// through pinned memory in the one-slot layout (main rows, then tail rows): one contiguous copy when
// the context has one slot, a strided (2-D) copy per part when the slots' rows are interleaved
bool tab_is_packed(const RowTab &t, int rows) {
  return t.rs_m == t.mw && t.rs_t == t.tw && (t.tw == 0 || t.tail == t.main + static_cast<size_t>(rows) * t.mw);
}
void copy_rows(mmsbm_hip_ctx *c, const RowTab &t, double *stage, int rows, bool to_device) {
  const size_t e = sizeof(double);
  hipStream_t xs = c->xfer ? c->xfer : c->stream;
  if (rows == 0) return;
  if (tab_is_packed(t, rows)) {
    if (to_device)
      HIP_CHECK(hipMemcpyAsync(t.main, stage, e * rows * (t.mw + t.tw), hipMemcpyHostToDevice, xs));
    else
      HIP_CHECK(hipMemcpyAsync(stage, t.main, e * rows * (t.mw + t.tw), hipMemcpyDeviceToHost, xs));
    return;
  }
  double *stage_t = stage + static_cast<size_t>(rows) * t.mw;
  if (to_device) {
    HIP_CHECK(hipMemcpy2DAsync(t.main, e * t.rs_m, stage, e * t.mw, e * t.mw, rows, hipMemcpyHostToDevice, xs));
    if (t.tw > 0)
      HIP_CHECK(hipMemcpy2DAsync(t.tail, e * t.rs_t, stage_t, e * t.tw, e * t.tw, rows, hipMemcpyHostToDevice, xs));
  } else {
    HIP_CHECK(hipMemcpy2DAsync(stage, e * t.mw, t.main, e * t.rs_m, e * t.mw, rows, hipMemcpyDeviceToHost, xs));
    if (t.tw > 0)
      HIP_CHECK(hipMemcpy2DAsync(stage_t, e * t.tw, t.tail, e * t.rs_t, e * t.tw, rows, hipMemcpyDeviceToHost, xs));
  }
}
void zero_rows(mmsbm_hip_ctx *c, const RowTab &t, int rows) {
  const size_t e = sizeof(double);
  if (rows == 0) return;
  if (tab_is_packed(t, rows)) {
    HIP_CHECK(hipMemsetAsync(t.main, 0, e * rows * (t.mw + t.tw), c->stream));
    return;
  }
  HIP_CHECK(hipMemset2DAsync(t.main, e * t.rs_m, 0, e * t.mw, rows, c->stream));
  if (t.tw > 0) HIP_CHECK(hipMemset2DAsync(t.tail, e * t.rs_t, 0, e * t.tw, rows, c->stream));
}
void upload_rows(mmsbm_hip_ctx *c, const RowTab &t, const double *host, int rows, int d) {
  const int dp = t.mw + t.tw, mw = t.mw, tw = t.tw;
  double *stage = c->pin.take(static_cast<size_t>(rows) * dp);
  double *tail = stage + static_cast<size_t>(rows) * mw;
  const int wm = std::min(d, mw), wt = std::max(0, d - mw);
  for_row_blocks(rows, dp, [=](int a, int b) {
    for (int r = a; r < b; ++r) {
      const double *src = host + static_cast<size_t>(r) * d;
      double *m = stage + static_cast<size_t>(r) * mw;
      std::memcpy(m, src, sizeof(double) * wm);
      for (int j = wm; j < mw; ++j) m[j] = 0.0;
      if (tw > 0) {
        double *tl = tail + static_cast<size_t>(r) * tw;
        std::memcpy(tl, src + mw, sizeof(double) * wt);
        for (int j = wt; j < tw; ++j) tl[j] = 0.0;
      }
    }
  });
  copy_rows(c, t, stage, rows, true);
}
// enqueue the device -> pinned copy; unpack_rows after the stream has been synchronised
double *download_rows(mmsbm_hip_ctx *c, const RowTab &t, int rows) {
  const int dp = t.mw + t.tw;
  double *stage = c->pin.take(static_cast<size_t>(rows) * dp);
  copy_rows(c, t, stage, rows, false);
  return stage;
}
void unpack_rows(double *host, const double *stage, const RowTab &t, int rows, int d) {
  const int dp = t.mw + t.tw, mw = t.mw, tw = t.tw;
  const double *tail = stage + static_cast<size_t>(rows) * mw;
  const int wm = std::min(d, mw), wt = std::max(0, d - mw);
  for_row_blocks(rows, dp, [=](int a, int b) {
    for (int r = a; r < b; ++r) {
      double *dst = host + static_cast<size_t>(r) * d;
      std::memcpy(dst, stage + static_cast<size_t>(r) * mw, sizeof(double) * wm);
      if (tw > 0 && wt > 0) std::memcpy(dst + mw, tail + static_cast<size_t>(r) * tw, sizeof(double) * wt);
    }
  });
}
size_t rows_doubles(const mmsbm_hip_ctx *c) {  // staging for theta + eta + p + pT of one slot
  return static_cast<size_t>(c->n_users) * c->kp + static_cast<size_t>(c->n_items) * c->lp +
         2 * static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
}

// device p layout [R][kp][lp] (internal k, l)  <->  host pr (K, L, R) external
void p_host_to_dev(const mmsbm_hip_ctx *c, const double *pr, double *p, double *pt) {
  const int R = c->n_ratings, K = c->k, L = c->l, kp = c->kp, lp = c->lp;
  std::fill(p, p + static_cast<size_t>(R) * kp * lp, 0.0);
  std::fill(pt, pt + static_cast<size_t>(R) * kp * lp, 0.0);
  for (int k = 0; k < K; ++k)
    for (int l = 0; l < L; ++l)
      for (int r = 0; r < R; ++r) {
        // internal (k,l) == external (l,k) when swapped
        const size_t h = c->swapped ? (static_cast<size_t>(l) * c->ext_l + k) * R + r
                                    : (static_cast<size_t>(k) * c->ext_l + l) * R + r;
        const double v = pr[h];
        p[(static_cast<size_t>(r) * kp + k) * lp + l] = v;
        pt[(static_cast<size_t>(r) * lp + l) * kp + k] = v;
      }
}
void p_dev_to_host(const mmsbm_hip_ctx *c, const double *p, double *pr) {
  const int R = c->n_ratings, K = c->k, L = c->l, kp = c->kp, lp = c->lp;
  for (int k = 0; k < K; ++k)
    for (int l = 0; l < L; ++l)
      for (int r = 0; r < R; ++r) {
        const size_t h = c->swapped ? (static_cast<size_t>(l) * c->ext_l + k) * R + r
                                    : (static_cast<size_t>(k) * c->ext_l + l) * R + r;
        pr[h] = p[(static_cast<size_t>(r) * kp + k) * lp + l];
      }
}

// (theta, eta, p) tables of one slot -> host arrays in host layout; any output may be null
void fetch_params(mmsbm_hip_ctx *c, const RowTab &tt, const RowTab &et, const double *p_dev,
                  double *theta, double *eta, double *pr) {
  hipStream_t xs = c->xfer ? c->xfer : c->stream;
  HIP_CHECK(hipStreamSynchronize(xs));
  c->pin.reset(rows_doubles(c));
  const size_t klr = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  const double *st = theta ? download_rows(c, tt, c->n_users) : nullptr;
  const double *se = eta ? download_rows(c, et, c->n_items) : nullptr;
  double *sp = nullptr;
  if (pr) {
    sp = c->pin.take(klr);
    HIP_CHECK(hipMemcpyAsync(sp, p_dev, sizeof(double) * klr, hipMemcpyDeviceToHost, xs));
  }
  HIP_CHECK(hipStreamSynchronize(xs));
  if (theta) unpack_rows(theta, st, tt, c->n_users, c->k);
  if (eta) unpack_rows(eta, se, et, c->n_items, c->l);
  if (pr) p_dev_to_host(c, sp, pr);
}

void collect_profile(mmsbm_hip_ctx *c, float *mean_us, int *launches, int n_iters) {
  std::vector<double> tot(K_COUNT, 0.0);
  std::vector<int> cnt(K_COUNT, 0);
  for (auto &pe : c->prof_events) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, pe.second.first, pe.second.second));
    tot[pe.first] += ms * 1000.0;
    cnt[pe.first]++;
    (void)hipEventDestroy(pe.second.first);
    (void)hipEventDestroy(pe.second.second);
  }
  c->prof_events.clear();
  for (int i = 0; i < K_COUNT; ++i) {
    mean_us[i] = cnt[i] ? static_cast<float>(tot[i] / cnt[i]) : 0.f;
    if (launches) launches[i] = n_iters > 0 ? cnt[i] / n_iters : 0;
  }
}


}  // namespace

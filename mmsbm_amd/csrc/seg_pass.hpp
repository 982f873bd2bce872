// seg_pass.hpp -- kernel 0: the two triple passes (pair segments, user segments) and the combine kernels of split segments
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
#pragma once

namespace {

// ======================================================================================
// kernel 0: seg_pass -- the two triple passes (user segments and pair segments), fused
// into one launch.  One group of G lanes per segment; lane gl owns VEC consecutive
// entries of the K-vector.
//
//   acc[:] = sum_{n in segment} gath[idx[n], :] / max(fixed[seg, :] . gath[idx[n], :], eps)
//
//   user segments: fixed = theta, gath = A, out = theta * acc / d_u   (src/mmsbm.py:248)
//   pair segments: fixed = A,     gath = theta, out = C = acc
// ======================================================================================
struct SegArgs {
  RowTab fixed;
  RowTab gath;
  const int32_t *off;
  const int32_t *idx;
  RowTab out;
  int32_t nseg;  // number of work units: segments, or work items when `items` is set
  int32_t mode;  // 0: out = acc   1: out = fixed*acc/max(len,1)   2: out = fixed*acc
  const mmsbm::WorkItem *items;  // null: unit w is segment w.  Else unit w is a piece of a segment
  double *parts;                 // [n_parts][dp] partial rows of the split segments
  size_t bs_parts;               // restart slots: distance in doubles between the slots' partial rows
  int32_t nt_out = 0;            // bit 0: finished rows (mode != 0: theta') as non-temporal stores, bit 1: the segment's own row as a non-temporal load
};
// One body for both forms.  SW == 1: a group of G lanes per segment, the restart slot is blockIdx.y.
// SW > 1: a "super-group" of SW x G lanes walks one segment for SW slots (lane = slot * G + gl).  The
// slots' copies of a gathered row are neighbours in memory (RowTab), so one index load serves all of
// them and the gather of a row is one contiguous piece of SW * 8 * dp bytes -- whole cache lines, no
// separate 32-byte tail access; blockIdx.y = group of SW slots.  Per (segment, slot) the arithmetic
// does not depend on SW: slot s of a batch is bitwise what a one-slot context computes.
//
// Lane gl of a group owns the VEC consecutive doubles gl * VEC .. of the row (two 16-byte loads per row
// at VEC = 4).  Dealing the columns out interleaved instead (load instruction j of lane gl = double2
// number j * G + gl, so that one instruction of a group covers one whole 128-byte line and the two
// loads of a row never wait on each other's pending miss in the L1 -- TCP_PENDING_STALL_CYCLES is 40 %
// of the launch) was measured: bitwise-different sums, same accuracy, 98.3 vs 96.9 us per iteration at
// C3 (16 more VGPRs for the per-instruction addresses); not kept.
template <int G, int VEC, int B, int SW>
__device__ __forceinline__ void seg_body(const SegArgs &a, int unit, int dp, int n_slots) {
  constexpr int GS = G * SW;
  const int sl = threadIdx.x % GS, gl = sl % G;
  const int slot = SW == 1 ? static_cast<int>(blockIdx.y) : static_cast<int>(blockIdx.y) * SW + sl / G;
  if (unit >= a.nseg) return;  // whole (super-)groups leave together
  const bool slot_ok = SW == 1 || slot < n_slots;
  const size_t sidx = slot_ok ? slot : 0;
  const RowTab fixed = slot_tab(a.fixed, sidx), gath = slot_tab(a.gath, sidx), outt = slot_tab(a.out, sidx);
  int seg = unit, beg, end, part = -1;
  if (a.items) {
    const mmsbm::WorkItem it = a.items[unit];
    seg = it.seg; beg = it.begin; end = it.end; part = it.part;
    if (seg < 0) return;  // padding of an XCD-local work list
  } else {
    beg = a.off[unit];
    end = a.off[unit + 1];
  }
  const bool act = gl * VEC < dp && slot_ok;
  const int lane_off = gl * VEC < dp ? gl * VEC : 0;
  double f[VEC], acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
  load_vec_in<VEC>(rowtab_ptr(fixed, seg, lane_off), f, (a.nt_out & 2) != 0);
  if (!act) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) f[v] = 0.0;
  }
  // this lane's part of every gathered row: main or tail, fixed for the whole kernel
  const bool g_main = lane_off < gath.mw;
  const double *gbase = g_main ? gath.main + lane_off : gath.tail + (lane_off - gath.mw);
  const size_t gstride = g_main ? gath.rs_m : gath.rs_t;

  // Every lane of the (super-)group fetches one index of the segment (one coalesced load per GS
  // triples, two per lane in small groups); the indices are then broadcast with ds_bpermute, so the
  // dependent chain is offsets -> indices -> rows instead of one index load per batch.
  constexpr int CH = (GS < 16) ? 2 * GS : GS;
  for (int c0 = beg; c0 < end; c0 += CH) {
    const int cnt = min(CH, end - c0);
    const int mine0 = a.idx[c0 + min(sl, cnt - 1)];
    const int mine1 = (CH > GS) ? a.idx[c0 + min(GS + sl, cnt - 1)] : 0;
    for (int n = 0; n < cnt; n += B) {
      double g[B][VEC];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int jj = min(n + b, cnt - 1);
        const int id = __shfl((CH > GS && jj >= GS) ? mine1 : mine0, jj, GS);
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

  if (!act) return;
  if (part >= 0) {  // a piece of a long segment: raw partial sum, finished by the combine kernels below
    store_vec<VEC>(a.parts + sidx * a.bs_parts + static_cast<size_t>(part) * dp + lane_off, acc);
    return;
  }
  double o[VEC];
  if (a.mode == 0) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = acc[v];
  } else if (a.mode == 1) {
    const double d = static_cast<double>(max(end - beg, 1));
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = (f[v] * acc[v]) / d;
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = f[v] * acc[v];
  }
  store_vec_out<VEC>(rowtab_ptr(outt, seg, lane_off), o, (a.nt_out & 1) != 0 && a.mode != 0);
}

// blocks [0, blocks_a) work on segment set `sa`, the rest on `sb`
template <int G, int VEC, int B>
__global__ __launch_bounds__(kBlock) void seg_pass_kernel(SegArgs sa, SegArgs sb,
                                                          int blocks_a, int dp) {
  const bool first = static_cast<int>(blockIdx.x) < blocks_a;
  const int blk = first ? blockIdx.x : blockIdx.x - blocks_a;
  seg_body<G, VEC, B, 1>(first ? sa : sb, blk * (kBlock / G) + threadIdx.x / G, dp, 1);
}

template <int G, int VEC, int B, int SW>
__global__ __launch_bounds__(kBlock) void seg_pass_slots_kernel(SegArgs sa, SegArgs sb, int blocks_a,
                                                                int dp, int n_slots) {
  const bool first = static_cast<int>(blockIdx.x) < blocks_a;
  const int blk = first ? blockIdx.x : blockIdx.x - blocks_a;
  seg_body<G, VEC, B, SW>(first ? sa : sb, blk * (kBlock / (G * SW)) + threadIdx.x / (G * SW), dp, n_slots);
}

// Rows of more than 1,024 groups (the widest instantiation above is 64 lanes x 16 doubles).  The reference has no
// size limit (src/kernels_numpy.py:21-79), so these run too -- in a plain form: one WAVE per segment, the row walked in
// blocks of 64 x 16 doubles.  A triple's weight needs the dot product over the WHOLE row before any column can be
// added up, so a chunk of up to 64 triples is handled in two sweeps: first every triple's weight (lane j keeps the
// weight of triple j), then, block by block, acc[block] += row[block] * weight over the chunk's triples -- the rows
// are read twice (the second time from L2).  Between the chunks of a long segment the raw sums wait in the output row.
// Same sums as seg_body in the same order per column; only the dot product is associated per block.
template <int VEC>
__global__ __launch_bounds__(kBlock) void seg_wide_kernel(SegArgs sa, SegArgs sb, int blocks_a, int dp) {
  constexpr int G = 64, BW = G * VEC;  // columns per block
  const bool first = static_cast<int>(blockIdx.x) < blocks_a;
  const SegArgs &a = first ? sa : sb;
  const int blk = first ? blockIdx.x : blockIdx.x - blocks_a;
  const int lane = threadIdx.x % G, unit = blk * (kBlock / G) + threadIdx.x / G;
  if (unit >= a.nseg) return;  // whole waves
  const size_t slot = blockIdx.y;
  const RowTab fixed = slot_tab(a.fixed, slot), gath = slot_tab(a.gath, slot), outt = slot_tab(a.out, slot);
  const int seg = unit, beg = a.off[unit], end = a.off[unit + 1];  // (no work lists for these shapes)
  const int nblk = (dp + BW - 1) / BW;
  for (int c0 = beg; c0 < end; c0 += G) {
    const int cnt = min(G, end - c0);
    const int myidx = a.idx[c0 + min(lane, cnt - 1)];
    double myw = 0.0;
    for (int j = 0; j < cnt; ++j) {
      const size_t id = static_cast<size_t>(__shfl(myidx, j, G));
      double pt = 0.0;
      for (int b = 0; b < nblk; ++b) {
        const int off = b * BW + lane * VEC;
        if (off < dp) {
          double g[VEC], f[VEC];
          load_vec<VEC>(rowtab_ptr(gath, id, off), g);
          load_vec<VEC>(rowtab_ptr(fixed, static_cast<size_t>(seg), off), f);
#pragma unroll
          for (int v = 0; v < VEC; ++v) pt = fma(g[v], f[v], pt);
        }
      }
      const double w = 1.0 / fmax(group_sum<G>(pt), kEps);
      if (lane == j) myw = w;
    }
    for (int b = 0; b < nblk; ++b) {
      const int off = b * BW + lane * VEC;
      if (off >= dp) continue;  // (no cross-lane operation below: each lane owns its columns)
      double acc[VEC];
      if (c0 == beg) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
      } else {
        load_vec<VEC>(rowtab_ptr(outt, static_cast<size_t>(seg), off), acc);
      }
      for (int j = 0; j < cnt; ++j) {
        // (shuffles by every lane of the wave that is still here: lanes past the row's end left together above)
        const size_t id = static_cast<size_t>(__builtin_amdgcn_readlane(myidx, j));
        const double w = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(myw), j),
                                          __builtin_amdgcn_readlane(__double2loint(myw), j));
        double g[VEC];
        load_vec<VEC>(rowtab_ptr(gath, id, off), g);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fma(g[v], w, acc[v]);
      }
      if (c0 + G >= end && a.mode != 0) {  // the segment is complete: its epilogue
        double f[VEC];
        load_vec<VEC>(rowtab_ptr(fixed, static_cast<size_t>(seg), off), f);
        const double d = static_cast<double>(max(end - beg, 1));
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = a.mode == 1 ? (f[v] * acc[v]) / d : f[v] * acc[v];
      }
      store_vec<VEC>(rowtab_ptr(outt, static_cast<size_t>(seg), off), acc);
    }
  }
  if (beg == end) {  // an id that never occurs: a zero row (what the loop-free epilogue of seg_body gives)
    for (int b = 0; b < nblk; ++b) {
      const int off = b * BW + lane * VEC;
      if (off >= dp) continue;
      double z[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) z[v] = 0.0;
      store_vec<VEC>(rowtab_ptr(outt, static_cast<size_t>(seg), off), z);
    }
  }
}

// Long segments: add the pieces' partial rows in piece order and apply the epilogue.
struct CombineArgs {
  const mmsbm::SplitSeg *splits;
  const double *parts;
  const int32_t *off;
  RowTab fixed, out;
  int32_t n_splits, mode;
  size_t bs_parts;  // restart slots, as in SegArgs
};

// One workgroup per split segment: its kBlock/G groups add the pieces j = g, g + NG, ... (four
// loads in flight), the per-group sums meet in LDS and are added in group order.
template <int G, int VEC>
__device__ __forceinline__ void seg_combine_big(const CombineArgs &ca, const CombineArgs &cb, int block, int blocks_a, int dp) {
  extern __shared__ double lds[];  // [kBlock / G][dp]
  constexpr int NG = kBlock / G;
  const bool first = block < blocks_a;
  const CombineArgs &a = first ? ca : cb;
  const int w = first ? block : block - blocks_a;
  const int grp = threadIdx.x / G, gl = threadIdx.x % G;
  const bool act = gl * VEC < dp;
  const int lane_off = act ? gl * VEC : 0;
  const mmsbm::SplitSeg sp = a.splits[w];
  const double *parts = a.parts + blockIdx.y * a.bs_parts;
  double acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
  for (int j0 = grp; j0 < sp.n_parts; j0 += NG * 4) {
    double t[4][VEC];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      load_vec<VEC>(parts + static_cast<size_t>(sp.first_part + min(j0 + i * NG, sp.n_parts - 1)) * dp +
                        lane_off, t[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (j0 + i * NG < sp.n_parts) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += t[i][v];
      }
  }
  if (act) store_vec<VEC>(lds + grp * dp + lane_off, acc);
  __syncthreads();
  if (grp != 0 || !act) return;
  // (four loads in flight, not all NG - 1 of them: fully unrolled with its loads hoisted this loop took 256 registers
  // and spilled at G = 4 -- NG = 64 rows of VEC doubles; the additions stay in group order)
#pragma unroll 4
  for (int g = 1; g < NG; ++g) {
    double t[VEC];
    load_vec<VEC>(lds + g * dp + lane_off, t);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += t[v];
  }
  double f[VEC], o[VEC];
  load_vec<VEC>(rowtab_ptr(slot_tab(a.fixed, blockIdx.y), sp.seg, lane_off), f);
  const double d = static_cast<double>(max(a.off[sp.seg + 1] - a.off[sp.seg], 1));
#pragma unroll
  for (int v = 0; v < VEC; ++v)
    o[v] = a.mode == 0 ? acc[v] : (a.mode == 1 ? (f[v] * acc[v]) / d : f[v] * acc[v]);
  store_vec<VEC>(rowtab_ptr(slot_tab(a.out, blockIdx.y), sp.seg, lane_off), o);
}

// Split segments with few pieces (the usual case when MANY segments are cut: dense data): one group
// of lanes per split segment adds its pieces in piece order, four loads in flight.
template <int G, int VEC>
__device__ __forceinline__ void seg_combine_small(const CombineArgs &ca, const CombineArgs &cb, int block, int blocks_a, int dp) {
  const bool first = block < blocks_a;
  const CombineArgs &a = first ? ca : cb;
  const int blk = first ? block : block - blocks_a;
  const int w = blk * (kBlock / G) + threadIdx.x / G, gl = threadIdx.x % G;
  if (w >= a.n_splits || gl * VEC >= dp) return;
  const int lane_off = gl * VEC;
  const mmsbm::SplitSeg sp = a.splits[w];
  const double *parts = a.parts + blockIdx.y * a.bs_parts + static_cast<size_t>(sp.first_part) * dp + lane_off;
  double acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
  for (int j0 = 0; j0 < sp.n_parts; j0 += 4) {
    double t[4][VEC];
#pragma unroll
    for (int i = 0; i < 4; ++i) load_vec<VEC>(parts + static_cast<size_t>(min(j0 + i, sp.n_parts - 1)) * dp, t[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (j0 + i < sp.n_parts) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += t[i][v];
      }
  }
  double f[VEC], o[VEC];
  load_vec<VEC>(rowtab_ptr(slot_tab(a.fixed, blockIdx.y), sp.seg, lane_off), f);
  const double d = static_cast<double>(max(a.off[sp.seg + 1] - a.off[sp.seg], 1));
#pragma unroll
  for (int v = 0; v < VEC; ++v)
    o[v] = a.mode == 0 ? acc[v] : (a.mode == 1 ? (f[v] * acc[v]) / d : f[v] * acc[v]);
  store_vec<VEC>(rowtab_ptr(slot_tab(a.out, blockIdx.y), sp.seg, lane_off), o);
}

template <int G, int VEC>
__global__ __launch_bounds__(kBlock) void seg_combine_small_kernel(CombineArgs ca, CombineArgs cb, int blocks_a, int dp) {
  seg_combine_small<G, VEC>(ca, cb, static_cast<int>(blockIdx.x), blocks_a, dp);
}
// Splits of many pieces -- and, in the same launch (a launch costs more than what it does here), the splits of few
// pieces if there are any: blocks [0, n_small) take those, a group of lanes each; the rest take one split of many pieces
// each.
template <int G, int VEC>
__global__ __launch_bounds__(kBlock) void seg_combine_both_kernel(CombineArgs sa, CombineArgs sb, int small_a, int n_small,
                                                                  CombineArgs ba, CombineArgs bb, int big_a, int dp) {
  const int bx = static_cast<int>(blockIdx.x);
  if (bx < n_small) seg_combine_small<G, VEC>(sa, sb, bx, small_a, dp);
  else seg_combine_big<G, VEC>(ba, bb, bx - n_small, big_a, dp);
}

// ---- host side: the argument blocks of these kernels from the context ----
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

}  // namespace

// fused_small.hpp -- small problems: the EM iteration in TWO launches instead of four (round 3)
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
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
  int kp, lp, spb, nsub, nt;  // nt: A and T rows as non-temporal stores
  size_t bs_tiles, bs_eta, bs_t, bs_partial;  // restart slots (blockIdx.y): strides of the streamed tables
  // data whose long pair segments are cut into pieces (SPLIT): the unit's work items in (pair, piece) order and
  // its split pairs (layout.hpp: FusedLists); null otherwise
  const mmsbm::FusedUnit *units;
  const mmsbm::WorkItem *items;
  const mmsbm::FusedSplit *splits;
};

// The pieces' partial rows of a split segment, added in the order the combine kernels of the separate launches use
// (seg_pass.hpp) -- seg_combine_small: one after the other; seg_combine_big (more than kSmallSplitParts pieces): group
// g of a workgroup's NG adds the pieces g, g + NG, ..., then the groups' sums are added in group order -- so that the
// result is bit for bit theirs.  parts: the unit's partial rows in LDS, [piece][dp].
template <int G, int VEC>
__device__ __forceinline__ void combine_lds(const double *parts, const mmsbm::FusedSplit &sp, int dp, int lane_off,
                                            double (&tot)[VEC]) {
  constexpr int NG = kBlock / G;
  const double *src = parts + static_cast<size_t>(sp.first_part) * dp + lane_off;
#pragma unroll
  for (int v = 0; v < VEC; ++v) tot[v] = 0.0;
  if (!sp.big || sp.n_parts <= NG) {
    for (int j = 0; j < sp.n_parts; ++j) {
      double t[VEC];
      load_vec<VEC>(src + static_cast<size_t>(j) * dp, t);
#pragma unroll
      for (int v = 0; v < VEC; ++v) tot[v] += t[v];
    }
    return;
  }
  for (int g = 0; g < NG; ++g) {
    double a[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[v] = 0.0;
    for (int j = g; j < sp.n_parts; j += NG) {
      double t[VEC];
      load_vec<VEC>(src + static_cast<size_t>(j) * dp, t);
#pragma unroll
      for (int v = 0; v < VEC; ++v) a[v] += t[v];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) tot[v] += a[v];
  }
}

template <int G, int VEC, bool SPLIT>
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
  double *parts = cT + static_cast<size_t>(kp) * CS;        // (SPLIT) [pieces of the unit's split pairs][kp]
  STAMP(0);
  STAMP_WHERE(9);
  const mmsbm::Chunk ch = fa.chunks[blockIdx.x];
  mmsbm::FusedUnit fu{0, 0, 0, 0};
  if (SPLIT) fu = fa.units[blockIdx.x];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q0 = ch.q_begin, np = ch.q_end - ch.q_begin;  // (np <= 64; 0 for the padding of the unit lists)
  typedef const double __attribute__((address_space(4))) * const_tile_ptr;
  const const_tile_ptr gpt = (const_tile_ptr)(reinterpret_cast<uintptr_t>(pt_tiles + static_cast<size_t>(ch.rating) * lp * kp));
  const const_tile_ptr gp = (const_tile_ptr)(reinterpret_cast<uintptr_t>(p_tiles + static_cast<size_t>(ch.rating) * kp * lp));

  // ---- everything the unit needs from memory, asked for in dependency LEVELS (the vector-memory wait counter is
  // in order: what is asked for first must arrive first, so independent loads of one level go out together and
  // nothing waits on a younger load): offsets + item ids -> first indices + eta rows -> first theta rows of round 0,
  // which then travel while A is multiplied ----
  constexpr int NGRP = kBlock / G, ROUNDS = (kUnitPairs + NGRP - 1) / NGRP;
  const int grp = tid / G, gl = tid % G;
#ifdef MMSBM_FUSED_CH  // (tuning aid, scripts/ab_fused.sh)
  constexpr int CH = MMSBM_FUSED_CH > G ? MMSBM_FUSED_CH : G;
#else
  constexpr int CH = B;        // triples per step of a segment = row gathers in flight per group
#endif
  constexpr int IPL = CH / G;  // indices a lane holds per step (G = 4: four, G = 8: two)
  static_assert(IPL == 1 || IPL == 2 || IPL == 4, "groups of 4 or 8 lanes");
  constexpr int NE = 4;  // eta loads per thread: 64 pairs x <= 32 entries / 2 per load / 256 threads
  const int tot = np * lp;
  int beg[ROUNDS], end[ROUNDS], mine[ROUNDS][IPL], nxt[ROUNDS][IPL];
  int col[ROUNDS], prt[ROUNDS];  // (SPLIT) the item's pair within the unit (-1: no item) and its partial row (-1: whole pair)
  int pr[NE], d[NE], ids[NE];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {  // level 1
    const int p0 = grp + r * NGRP;
    if (SPLIT) {  // work item p0 of the unit (at most kUnitPairs of them: the unit list is built that way)
      const bool on = fu.it_begin + p0 < fu.it_end;
      mmsbm::WorkItem it{-1, 0, 0, -1};
      if (on) it = fa.items[fu.it_begin + p0];
      beg[r] = it.begin; end[r] = it.end; prt[r] = it.part;
      col[r] = on ? it.seg - q0 : -1;
    } else {
      const bool on = p0 < np;
      beg[r] = on ? fa.pair_off[q0 + p0] : 0;
      end[r] = on ? fa.pair_off[q0 + p0 + 1] : 0;
    }
  }
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int t = min(tid * 2 + j * kBlock * 2, max(tot - 2, 0));
    pr[j] = t / lp;
    d[j] = t - pr[j] * lp;
    ids[j] = tot > 0 ? fa.pair_item[q0 + pr[j]] : 0;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {  // level 2: the indices of a segment's first two steps
    const int len = end[r] - beg[r];
    const int last = max(len - 1, 0);
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
      mine[r][i] = len > i * G ? fa.pair_user[beg[r] + min(i * G + gl, last)] : 0;
      nxt[r][i] = len > CH + i * G ? fa.pair_user[beg[r] + min(CH + i * G + gl, last)] : 0;
    }
  }
  double2 ev[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) ev[j] = *reinterpret_cast<const double2 *>(eta + static_cast<size_t>(ids[j]) * lp + d[j]);
  __builtin_amdgcn_sched_barrier(0);  // (the compiler moved one of these behind the gathers below, and with it the wait)
  const bool act = gl * VEC < kp;
  const int lane_off = act ? gl * VEC : 0;
  const bool g_main = lane_off < theta.mw;
  const double *gbase = g_main ? theta.main + lane_off : theta.tail + (lane_off - theta.mw);
  const size_t gstride = g_main ? theta.rs_m : theta.rs_t;
  double gpre[CH][VEC];
  {  // level 3: the rows of round 0's first step (only those the segment has: a group that is done asks for nothing)
    const int len0 = end[0] - beg[0];
#pragma unroll
    for (int b = 0; b < CH; ++b) {
      const int id = __shfl(mine[0][b / G], b % G, G);
      if (b < len0) load_vec<VEC>(gbase + static_cast<size_t>(id) * gstride, gpre[b]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  {  // eta rows -> etaT (transposed) and es (row-major)
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      if (tid * 2 + j * kBlock * 2 < tot) {
        etaT[d[j] * CS + pr[j]] = ev[j].x;
        etaT[(d[j] + 1) * CS + pr[j]] = ev[j].y;
        *reinterpret_cast<double2 *>(es + pr[j] * lp + d[j]) = ev[j];
      }
    }
    if (np < kUnitPairs) {  // ragged tail of a rating: the missing pairs are zero columns / zero rows
      for (int t = tid; t < (kUnitPairs - np) * lp; t += kBlock) {
        etaT[(t / (kUnitPairs - np)) * CS + np + t % (kUnitPairs - np)] = 0.0;
        es[np * lp + t] = 0.0;
      }
    }
  }
  STAMP(1);
  __syncthreads();
  STAMP(2);
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
  STAMP(3);
  // the unit's A rows go to memory for launch 2 (the user pass gathers them)
  for (int t = tid * 2; t < np * kp; t += kBlock * 2) {
    const int pr = t / kp, j = t - pr * kp;
    store_out2(rowtab_ptr(a_out, static_cast<size_t>(q0 + pr), j), *reinterpret_cast<const double2 *>(aout + t), fa.nt != 0);
  }
  // ---- the pair segments (seg_body's arithmetic): a group of G lanes per pair, C row into cT ----
  {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int pr = SPLIT ? col[r] : grp + r * NGRP;
      if (SPLIT ? pr < 0 : pr >= kUnitPairs) continue;  // (whole groups)
      double f[VEC], acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        acc[v] = 0.0;
        f[v] = act ? aout[pr * kp + lane_off + v] : 0.0;
      }
      // One step = CH = 16 triples, all of its rows in flight together.  The rows of round 0's first step were asked
      // for at the very top of the kernel and travelled while A was multiplied: a segment of up to 16 triples -- all
      // but a handful at the sizes this kernel serves -- waits for nothing here.  Longer ones: the next step's
      // indices are in registers a step ahead, its rows are asked for when the current step has been added up.
      double g[CH][VEC];
      if (r == 0) {
#pragma unroll
        for (int b = 0; b < CH; ++b)
#pragma unroll
          for (int v = 0; v < VEC; ++v) g[b][v] = gpre[b][v];
      } else {
        const int len = end[r] - beg[r];
#pragma unroll
        for (int b = 0; b < CH; ++b) {
          const int id = __shfl(mine[r][b / G], b % G, G);
          if (b < len) load_vec<VEC>(gbase + static_cast<size_t>(id) * gstride, g[b]);
        }
      }
      int n[IPL];
#pragma unroll
      for (int i = 0; i < IPL; ++i) n[i] = nxt[r][i];
      for (int c0 = beg[r]; c0 < end[r]; c0 += CH) {
        const int cnt = min(CH, end[r] - c0);
        int nn[IPL];
#pragma unroll
        for (int i = 0; i < IPL; ++i) nn[i] = 0;
        if (c0 + 2 * CH < end[r]) {  // (whole groups)
          const int last = end[r] - 1;
#pragma unroll
          for (int i = 0; i < IPL; ++i) nn[i] = fa.pair_user[min(c0 + 2 * CH + i * G + gl, last)];
        }
#pragma unroll
        for (int b = 0; b < CH; ++b) {
          if (b < cnt) {  // (weights of 0 for the empty slots instead of this branch: 2.36 vs 1.98 us for the phase)
            double pt = 0.0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) pt = fma(g[b][v], f[v], pt);
            const double s = group_sum<G>(pt);
            const double w = 1.0 / fmax(s, kEps);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fma(g[b][v], w, acc[v]);
          }
        }
        if (c0 + CH < end[r]) {
          const int left = end[r] - c0 - CH;
#pragma unroll
          for (int b = 0; b < CH; ++b) {
            const int id = __shfl(n[b / G], b % G, G);
            if (b < left) load_vec<VEC>(gbase + static_cast<size_t>(id) * gstride, g[b]);
          }
        }
#pragma unroll
        for (int i = 0; i < IPL; ++i) n[i] = nn[i];
      }
      if (act) {
        if (SPLIT && prt[r] >= 0) {  // a piece of a long pair segment: its partial row, added up below
          store_vec<VEC>(parts + static_cast<size_t>(prt[r]) * kp + lane_off, acc);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) cT[(lane_off + v) * CS + pr] = acc[v];  // (pairs >= np: zero columns)
        }
      }
    }
  }
  if (SPLIT) {
    // (the columns no item writes: those beyond the unit's pairs are zero; a split pair's column comes from its pieces)
    for (int t = tid; t < (kUnitPairs - np) * kp; t += kBlock) cT[(t / (kUnitPairs - np)) * CS + np + t % (kUnitPairs - np)] = 0.0;
    __syncthreads();
    for (int s = fu.sp_begin + grp; s < fu.sp_end; s += NGRP) {
      const mmsbm::FusedSplit sp = fa.splits[s];
      double tot[VEC];
      combine_lds<G, VEC>(parts, sp, kp, lane_off, tot);
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) cT[(lane_off + v) * CS + (sp.seg - q0)] = tot[v];
      }
    }
  }
  __syncthreads();
  STAMP(4);
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
  STAMP(5);
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
  STAMP(6);
  {  // the unit's T rows are contiguous in memory
    double *dst = t_out + static_cast<size_t>(q0) * lp;
    for (int t = tid * 2; t < np * lp; t += kBlock * 2)
      store_out2(dst + t, *reinterpret_cast<const double2 *>(tout + t), fa.nt != 0);
  }
  // ---- the slab: the copies hand their sums over through LDS; one thread per CELL of the slab then adds the copies
  // in copy order (the order of the four-launch form, where copy 0's threads do it for their 8 cells each) and the
  // slab goes out as one contiguous piece ----
  if (nsub > 1) {
    __syncthreads();
    if (s_active && slot0 < nslot) {
#pragma unroll
      for (int j = 0; j < TV; ++j) lds[(j * nsub + sub) * nslot + slot0] = sacc[j];
    }
    __syncthreads();
    double *slab = partial + static_cast<size_t>(blockIdx.x) * kp * lp;
    for (int t = tid; t < kp * lp; t += kBlock) {
      const int k = t / lp, l = t - k * lp;
      const int sl = (k / KT) * nch + (l >> 2), j = (k % KT) * 4 + (l & 3);
      const double *src = lds + static_cast<size_t>(j) * nsub * nslot + sl;
      double v = src[0];
      for (int oo = 1; oo < nsub; ++oo) v += src[oo * nslot];
      slab[t] = v;
    }
  } else if (sub == 0 && slot0 < nslot) {
    double *cell = partial + static_cast<size_t>(blockIdx.x) * kp * lp + (o / nch) * KT * lp + eoff;
#pragma unroll
    for (int h = 0; h < KT; ++h) {
      double2 x, y;
      x.x = sacc[4 * h]; x.y = sacc[4 * h + 1]; y.x = sacc[4 * h + 2]; y.y = sacc[4 * h + 3];
      *reinterpret_cast<double2 *>(cell + h * lp) = x;
      *reinterpret_cast<double2 *>(cell + h * lp + 2) = y;
    }
  }
#ifdef MMSBM_STAMPS
  STAMP(7);
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(8);
#endif
}

// seg_body (seg_pass.hpp) for the sizes of this file: the same sums in the same order (one triple after the other), but
// the indices of a segment's first CH triples in ONE load level and all CH rows in flight together -- a user with up
// to CH ratings costs offsets -> indices -> rows instead of an index and a row round trip per 8 triples; the next
// step's indices travel a step ahead.  (Rows a segment does not have are not asked for.)  Occupancy does not matter
// here -- the launch has fewer workgroups than the chip has CUs.
template <int G, int VEC, int CH>
__device__ __forceinline__ void seg_body_small(const SegArgs &a, int unit, int dp) {
  constexpr int IPL = CH / G;
  static_assert(IPL >= 1 && CH % G == 0, "groups of up to 16 lanes");
  const int gl = threadIdx.x % G;
  const size_t sidx = blockIdx.y;
  if (unit >= a.nseg) return;  // whole groups leave together
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
  const bool act = gl * VEC < dp;
  const int lane_off = act ? gl * VEC : 0;
  double f[VEC], acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
  load_vec<VEC>(rowtab_ptr(fixed, seg, lane_off), f);
  if (!act) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) f[v] = 0.0;
  }
  const bool g_main = lane_off < gath.mw;
  const double *gbase = g_main ? gath.main + lane_off : gath.tail + (lane_off - gath.mw);
  const size_t gstride = g_main ? gath.rs_m : gath.rs_t;
  if (end >= beg) STAMP(1);  // (diagnostic builds: the segment's range has arrived)
  const int last = max(end - 1, beg);
  int n[IPL];
#pragma unroll
  for (int i = 0; i < IPL; ++i) n[i] = end - beg > i * G ? a.idx[min(beg + i * G + gl, last)] : 0;
  for (int c0 = beg; c0 < end; c0 += CH) {
    const int cnt = min(CH, end - c0);
    double g[CH][VEC];
#pragma unroll
    for (int b = 0; b < CH; ++b) {
      const int id = __shfl(n[b / G], b % G, G);
      if (b < cnt) load_vec<VEC>(gbase + static_cast<size_t>(id) * gstride, g[b]);
    }
    if (n[0] >= 0) STAMP(2);  // (the indices have arrived, the rows are asked for)
    if (c0 + CH < end) {
#pragma unroll
      for (int i = 0; i < IPL; ++i) n[i] = a.idx[min(c0 + CH + i * G + gl, last)];
    }
#pragma unroll
    for (int b = 0; b < CH; ++b) {
      if (b < cnt) {
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
  if (acc[0] >= 0.0) STAMP(3);  // (rows arrived and added up)
  if (!act) return;
  if (part >= 0) {  // (a piece of a long segment: not at the sizes that take this path, kept for completeness)
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

// User segments cut into pieces, every segment's pieces inside ONE workgroup (layout.hpp: build_fused_users): the
// workgroup's groups of lanes take its work items round robin -- whole segments are finished as in seg_body_small,
// a piece leaves its raw partial row in LDS -- and after a barrier one group per split segment adds the pieces up in
// the combine kernels' order (combine_lds) and applies their epilogue.  Bit for bit seg_pass + seg_combine.
struct FusedUserArgs {
  const mmsbm::FusedUnit *units;
  const mmsbm::WorkItem *items;
  const mmsbm::FusedSplit *splits;
};
template <int G, int VEC, int CH>
__device__ __forceinline__ void users_split_block(const SegArgs &a, const FusedUserArgs &fu_args, int blk, int dp, double *parts) {
  constexpr int IPL = CH / G, NGRP = kBlock / G;
  static_assert(IPL >= 1 && CH % G == 0, "groups of up to 16 lanes");
  const int gl = threadIdx.x % G, grp = threadIdx.x / G;
  const size_t sidx = blockIdx.y;
  const RowTab fixed = slot_tab(a.fixed, sidx), gath = slot_tab(a.gath, sidx), outt = slot_tab(a.out, sidx);
  const mmsbm::FusedUnit fu = fu_args.units[blk];
  const bool act = gl * VEC < dp;
  const int lane_off = act ? gl * VEC : 0;
  const bool g_main = lane_off < gath.mw;
  const double *gbase = g_main ? gath.main + lane_off : gath.tail + (lane_off - gath.mw);
  const size_t gstride = g_main ? gath.rs_m : gath.rs_t;
  for (int it = fu.it_begin + grp; it < fu.it_end; it += NGRP) {  // (whole groups)
    const mmsbm::WorkItem w = fu_args.items[it];
    const int beg = w.begin, end = w.end;
    double f[VEC], acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
    load_vec<VEC>(rowtab_ptr(fixed, w.seg, lane_off), f);
    if (!act) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) f[v] = 0.0;
    }
    const int last = max(end - 1, beg);
    int n[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) n[i] = end - beg > i * G ? a.idx[min(beg + i * G + gl, last)] : 0;
    for (int c0 = beg; c0 < end; c0 += CH) {
      const int cnt = min(CH, end - c0);
      double g[CH][VEC];
#pragma unroll
      for (int b = 0; b < CH; ++b) {
        const int id = __shfl(n[b / G], b % G, G);
        if (b < cnt) load_vec<VEC>(gbase + static_cast<size_t>(id) * gstride, g[b]);
      }
      if (c0 + CH < end) {
#pragma unroll
        for (int i = 0; i < IPL; ++i) n[i] = a.idx[min(c0 + CH + i * G + gl, last)];
      }
#pragma unroll
      for (int b = 0; b < CH; ++b) {
        if (b < cnt) {
          double pt = 0.0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) pt = fma(g[b][v], f[v], pt);
          const double s = group_sum<G>(pt);
          const double wgt = 1.0 / fmax(s, kEps);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = fma(g[b][v], wgt, acc[v]);
        }
      }
    }
    if (!act) continue;
    if (w.part >= 0) {
      store_vec<VEC>(parts + static_cast<size_t>(w.part) * dp + lane_off, acc);
      continue;
    }
    double o[VEC];
    const double d = static_cast<double>(max(end - beg, 1));
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = a.mode == 0 ? acc[v] : (a.mode == 1 ? (f[v] * acc[v]) / d : f[v] * acc[v]);
    store_vec<VEC>(rowtab_ptr(outt, w.seg, lane_off), o);
  }
  __syncthreads();
  for (int s = fu.sp_begin + grp; s < fu.sp_end; s += NGRP) {
    const mmsbm::FusedSplit sp = fu_args.splits[s];
    double tot[VEC], f[VEC], o[VEC];
    combine_lds<G, VEC>(parts, sp, dp, lane_off, tot);
    if (!act) continue;
    load_vec<VEC>(rowtab_ptr(fixed, sp.seg, lane_off), f);
    const double d = static_cast<double>(max(a.off[sp.seg + 1] - a.off[sp.seg], 1));
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = a.mode == 0 ? tot[v] : (a.mode == 1 ? (f[v] * tot[v]) / d : f[v] * tot[v]);
    store_vec<VEC>(rowtab_ptr(outt, sp.seg, lane_off), o);
  }
}

// Rows in flight per user segment (UCH): 32 while the launch is ONE round of workgroups at one workgroup per CU (32 rows
// of 4 doubles per lane are 256 registers); 16 beyond.  Measured at BASELINE's config 1 on one box, per iteration:
// 8 (seg_body) 20.8 us, 16 20.4, 24 20.6, 32 19.8, 48 20.7; with the 157 user workgroups split into 314 smaller ones
// (a second round at 32) 23.5.
// launch 2: blocks [0, nb_p) p_update, [nb_p, nb_p + bu) user segments, the rest item_sum
template <int G, int VEC, int GL, int VECL, int UCH, bool SPLIT>
__global__ __launch_bounds__(kBlock) void tail_fused_kernel(SegArgs su, EtaPArgs a, int bu, int dp, FusedUserArgs fu) {
  __shared__ double red[kRedGroup][kRedRows][kRedCols];  // (eta_p_kernel's pattern: the same sums in the same order)
  extern __shared__ double split_parts[];                // (SPLIT) the partial rows of the workgroup's split user segments
  const int bx = static_cast<int>(blockIdx.x);
  const size_t slot = blockIdx.y;
  STAMP(0);
  STAMP_WHERE(9);
  if (bx < a.nb_p) {  // (the longest chain of the three first)
    p_update_block<kRedRows, kBlock / kRedCols, 2>(red, bx, a.partial + slot * a.bs_partial, a.chunk_off,
                                      a.p_old + slot * a.bs_p, a.p_new + slot * a.bs_p, a.pt_new + slot * a.bs_p,
                                      a.npr + slot * a.bs_p, a.n_ratings, a.kp, a.lp, a.normalize);
  } else if (bx < a.nb_p + bu) {
    if (SPLIT) users_split_block<G, VEC, (UCH > 16 ? 16 : UCH)>(su, fu, bx - a.nb_p, dp, split_parts);
    else seg_body_small<G, VEC, UCH>(su, (bx - a.nb_p) * (kBlock / G) + threadIdx.x / G, dp);
  } else {
    item_sum_block<GL, VECL>(bx - bu - a.nb_p, a.ttab + slot * a.bs_t, a.item_off, a.item_pairs, a.item_deg,
                             a.eta + slot * a.bs_eta, a.eta_new + slot * a.bs_eta, a.n_items, a.lp, a.normalize,
                             a.item_grid, a.n_ratings);
  }
#ifdef MMSBM_STAMPS
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(8);
#endif
}

}  // namespace

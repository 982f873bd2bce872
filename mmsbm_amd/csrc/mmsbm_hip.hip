// mmsbm_hip.hip -- MI355X (gfx950 / CDNA4) EM core for the Mixed-Membership Stochastic
// Block Model, behind the C ABI of include/mmsbm_hip.h.
//
// What the reference computes per EM iteration (src/kernels_numpy.py:43-79 followed by
// src/mmsbm.py:248-250), for every observed triple n = (u, i, r):
//
//   omega[n,k,l] = theta[u,k] eta[i,l] p[k,l,r]        s_n = sum_kl omega[n,k,l]
//   n_theta[u,k] = sum_{n: u_n=u} sum_l omega/max(s_n,eps)      (and the same for eta, p)
//
// materialising the (N,K,L) tensor several times.  This file never builds omega.  With
// a "pair" q = a distinct (item, rating) combination and
//
//   A[q,k]   = sum_l p[k,l,r_q] eta[i_q,l]                       (pair_matvec, K-side)
//   w_n      = 1 / max(theta[u_n,:] . A[q_n,:], eps)
//   n_theta[u,k] = theta[u,k] * sum_{n in user u} A[q_n,k] w_n   (seg_pass, user segments)
//   C[q,k]   = sum_{n in pair q} theta[u_n,k] w_n                (seg_pass, pair segments)
//   T[q,l]   = sum_k p[k,l,r_q] C[q,k]                           (pair_matvec, L-side)
//   n_eta[i,l] = eta[i,l] * sum_{q in item i} T[q,l]             (item_sum)
//   n_p[k,l,r] = p[k,l,r] * sum_{q: r_q=r} C[q,k] eta[i_q,l]     (p_partial, p_update)
//
// which is the same sum re-associated: O(N K + Q K L) flops instead of O(N K L), two
// row-gather passes over the triples (sorted user-major and (rating,item)-major), every
// reduction a segmented reduction in registers (no atomics: results are bitwise
// reproducible), all arithmetic float64 on the vector ALU (no MFMA).
//
// Written for gfx950 only: 64-wide wavefronts, DPP cross-lane reductions, LDS-staged
// rating tiles.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mmsbm_hip.h"
#include "layout.hpp"
#include "layout_gpu.hpp"
#include "pcg64.hpp"

namespace {

// ======================================================================================
// errors
// ======================================================================================
thread_local std::string g_last_error;

struct ApiError : std::runtime_error {
  int code;
  ApiError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(expr)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      throw ApiError(MMSBM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

template <class F>
int guarded(F &&f) {
  try {
    f();
    return MMSBM_OK;
  } catch (const ApiError &e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::invalid_argument &e) {
    g_last_error = e.what();
    return MMSBM_E_INVALID;
  } catch (const std::bad_alloc &) {
    g_last_error = "host allocation failed";
    return MMSBM_E_INTERNAL;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return MMSBM_E_INTERNAL;
  }
}

constexpr double kEps = 2.220446049250313e-16;  // np.finfo(float).eps, src/kernels_numpy.py:51
constexpr int64_t kGpuLayoutMin = 100'000;       // triples from which the layout's sorts run on the device
constexpr int kBlock = 256;

// ======================================================================================
// device helpers
// ======================================================================================
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Sum over the G consecutive lanes of a group (G a power of two, groups aligned to G).
// Every step adds a value to its mirror image, so all lanes of a group end with the
// bitwise-identical sum.  Up to 16 lanes stay inside one DPP row (no LDS traffic).
template <int G>
__device__ __forceinline__ double group_sum(double x) {
  if (G >= 2) x += dpp_move<0xB1>(x);    // quad_perm [1,0,3,2]
  if (G >= 4) x += dpp_move<0x4E>(x);    // quad_perm [2,3,0,1]
  if (G >= 8) x += dpp_move<0x141>(x);   // row_half_mirror
  if (G >= 16) x += dpp_move<0x140>(x);  // row_mirror
  if (G >= 32) x += __shfl_xor(x, 16, 64);
  if (G >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

template <int VEC>
__device__ __forceinline__ void load_vec(const double *__restrict__ p, double (&v)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    const double2 t = *reinterpret_cast<const double2 *>(p + j);
    v[j] = t.x;
    v[j + 1] = t.y;
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(double *__restrict__ p, const double (&v)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    double2 t;
    t.x = v[j];
    t.y = v[j + 1];
    *reinterpret_cast<double2 *>(p + j) = t;
  }
}

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
// A table of rows that are GATHERED by index (theta, A).  A 160-byte row (K = 20) straddles two
// 128-byte cache lines; the table is therefore kept as a "main" part of whole 128-byte lines
// (mw = 16 * floor(Kp/16) doubles per row, line aligned) plus a compact "tail" part
// (tw = Kp - mw doubles per row), so a gather misses on one line of the big main part and
// hits the small, cache-resident tail part.  mw == row width and tw == 0 describes a plain table.
struct RowTab {
  double *main;      // row r, off < mw:  main + r * rs_m + off
  double *tail;      // row r, off >= mw: tail + r * rs_t + (off - mw)
  int mw, tw;        // widths of the two parts (tw == 0: a plain table)
  int rs_m, rs_t;    // row strides in doubles
  size_t so_m, so_t; // distance between the copies of two consecutive restart slots (see below)
};
// Restart slots.  The two GATHERED tables (theta, A) keep the slots' copies of a row side by side:
// row r = [slot 0 | slot 1 | ...], so rs_m = n_slots * mw, so_m = mw (and the same for the tail part).
// One index then serves every slot and a gather of row r for all slots is ONE contiguous piece of
// n_slots * 160 bytes at K = 20 -- whole 128-byte lines, no separate 32-byte tail access.  Streamed
// tables (C, T, eta) are plain per-slot copies: rs_m = width, so_m = the table's size.
__device__ __forceinline__ double *rowtab_ptr(const RowTab &t, size_t row, int off) {
  return off < t.mw ? t.main + row * t.rs_m + off : t.tail + row * t.rs_t + (off - t.mw);
}

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
};
__device__ __forceinline__ RowTab slot_tab(RowTab t, size_t slot) {
  t.main += slot * t.so_m;
  t.tail += slot * t.so_t;
  return t;
}

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
  load_vec<VEC>(rowtab_ptr(fixed, seg, lane_off), f);
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
  if (part >= 0) {  // a piece of a long segment: raw partial sum, finished by seg_combine_kernel
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
  store_vec<VEC>(rowtab_ptr(outt, seg, lane_off), o);
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
__global__ __launch_bounds__(kBlock) void seg_combine_kernel(CombineArgs ca, CombineArgs cb,
                                                             int blocks_a, int dp) {
  extern __shared__ double lds[];  // [kBlock / G][dp]
  constexpr int NG = kBlock / G;
  const bool first = static_cast<int>(blockIdx.x) < blocks_a;
  const CombineArgs &a = first ? ca : cb;
  const int w = first ? blockIdx.x : blockIdx.x - blocks_a;
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
__global__ __launch_bounds__(kBlock) void seg_combine_small_kernel(CombineArgs ca, CombineArgs cb,
                                                                   int blocks_a, int dp) {
  const bool first = static_cast<int>(blockIdx.x) < blocks_a;
  const CombineArgs &a = first ? ca : cb;
  const int blk = first ? blockIdx.x : blockIdx.x - blocks_a;
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

// ======================================================================================
// pair_block -- the fused dense stage.  A block takes a unit of <= 64 consecutive pairs of
// ONE rating (more if its chunk is longer) and, per unit, stages in LDS (coalesced flat copies;
// the rating's Din x Dout tile p[r] / pT[r] is lane-uniform and comes through SGPRs instead):
//   cst[d][pair]    the 64 input rows, transposed (C rows, or gathered eta rows),
//   es[pair][:]     (DO_S) the 64 gathered eta rows (the region is reused for the output rows),
// then
//   mat-vec : out[q,:] = sum_d in[q,d] tile[d,:]  -- lane = pair, wave = chunk of 4 outputs,
//             results transposed through LDS and written as one contiguous 64-row block;
//   DO_S    : S[k][l] += sum_q C[q,k] eta[i_q,l]  -- thread = (k, 4 l) slot, kept in registers
//             across the block's units, one K x L slab per block at the end (combined by
//             p_update in a fixed order: deterministic, no atomics).
// T-mode: in = C (contiguous), tile = p[r] as [Kp][Lp], out = T, DO_S on.
// A-mode: in = eta gathered by pair_item, tile = pT[r] as [Lp][Kp], out = A.
// ======================================================================================
constexpr int kUnitPairs = 64;

// Diagnostic build only (-DMMSBM_STAMPS): thread 0 of every workgroup of the pair stage records the
// 100 MHz wall clock at its phase borders; nothing else in the kernels reads the buffer.
#ifdef MMSBM_STAMPS
constexpr int kStampSlots = 16, kStampBlocks = 8192;
__device__ unsigned long long g_stamps[kStampBlocks * kStampSlots];
#define STAMP(i)                                                                             \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < kStampBlocks && blockIdx.y == 0)                    \
      g_stamps[blockIdx.x * kStampSlots + (i)] = wall_clock64();                             \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

struct PairBlockArgs {
  const double *tiles; const double *in_tab; const double *e_tab; const int32_t *pair_item;
  const mmsbm::Chunk *chunks; double *out; double *partial;
  int din, dinp, doutp, spb, nsub, abl;
  // output rows: `out` is a plain [rows][doutp] table (T: out_mw == doutp, out_rs == doutp) or the
  // main part of a RowTab whose tail part starts at out_tail (A)
  int out_mw, out_rs_m, out_rs_t;
  double *out_tail;
  size_t bs_tiles, bs_in, bs_e, bs_out, bs_out_t, bs_partial;  // restart slots (blockIdx.y): offsets
};
// element j (a multiple of 2) of output row q
__device__ __forceinline__ double *pair_out_ptr(const PairBlockArgs &pa, double *out, double *out_tail,
                                                size_t q, int j) {
  return j < pa.out_mw ? out + q * pa.out_rs_m + j : out_tail + q * pa.out_rs_t + (j - pa.out_mw);
}

template <bool GATHER, bool DO_S, int NACC, bool TLDS, int NT, int KT, bool DIRECT>
__device__ __forceinline__ void pair_block_body(const PairBlockArgs &pa,
                                                const double *__restrict__ tiles0, int block) {
  const size_t slot = blockIdx.y;
  const double *__restrict__ tiles = tiles0 + slot * pa.bs_tiles;
  const double *__restrict__ in_tab = pa.in_tab + slot * pa.bs_in;
  const double *__restrict__ e_tab = pa.e_tab + slot * pa.bs_e;
  const int32_t *__restrict__ pair_item = pa.pair_item;
  double *__restrict__ out = pa.out + slot * pa.bs_out;
  double *__restrict__ out_tail = pa.out_tail + slot * pa.bs_out_t;
  double *__restrict__ partial = pa.partial + slot * pa.bs_partial;
  const int dinp = pa.dinp, doutp = pa.doutp, spb = pa.spb, abl = pa.abl;  // (rows >= din are zero)
  const int nsub = pa.nsub;
  // abl: tuning aid, normally 0 -- bit0 rows, bit1 eta rows, bit2 S, bit3 mat-vec, bit4 output
  // copy, bit5 slab store, bit6 tile staging are skipped when set
  extern __shared__ double lds[];
  constexpr int CS = kUnitPairs + 1;  // odd stride: conflict-free column AND row reads
  double *cst = lds;                                  // [dinp][CS]
  double *es = cst + static_cast<size_t>(dinp) * CS;  // [64][doutp]  gathered eta rows (DO_S) ...
  double *tout = es;                                  // ... then the mat-vec's output rows
  STAMP(0);
  // (the descriptor comes from memory: passing the unit -> pair-range map with the kernel arguments
  // instead was measured slower at C3 -- A launch 11.4 vs 10.0 us -- bigger argument blocks cost more
  // than the one dependent load they save)
  const mmsbm::Chunk ch = pa.chunks[block];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int nthr = NT;  // 256, or 512 for long rows (more waves to share the output chunks)
  const int nch = doutp >> 2;

  // The rating's tile is the same for every lane: it is read through the scalar cache
  // (s_load into SGPRs) and never touches LDS.
  // (constant address space: the tiles are never written by this launch, and AS4 loads
  // with a uniform address are always selected as scalar loads.)
  typedef const double __attribute__((address_space(4))) * const_tile_ptr;
  const const_tile_ptr gtile = (const_tile_ptr)(reinterpret_cast<uintptr_t>(
      tiles + static_cast<size_t>(ch.rating) * dinp * doutp));
  // Tiles too big for the scalar cache (TLDS) are staged in LDS once per workgroup instead and
  // read with broadcast ds_read_b128; the unit loop's first barrier orders this staging.
  double *tile_l = es + static_cast<size_t>(kUnitPairs) * doutp;  // [dinp][doutp], TLDS only
  if (TLDS) {
    const double *src = tiles + static_cast<size_t>(ch.rating) * dinp * doutp;
    for (int t = tid * 2; t < dinp * doutp; t += nthr * 2)
      *reinterpret_cast<double2 *>(tile_l + t) = *reinterpret_cast<const double2 *>(src + t);
  }
  // S slots (DO_S): a slot is a KT (k) x 4 (l) register tile (KT = 4: 16 FMAs per 4 + 2 LDS
  // reads; KT = 2 keeps more copies busy when K x L is small); `spb` threads form one copy of
  // the K x L slot grid and the block's nsub copies split each unit's pairs.
  constexpr int TV = KT * 4;
  const int nslot = (dinp / KT) * nch;
  const int sub = tid / spb, slot0 = tid % spb;
  const bool s_active = sub < nsub;
  int coff[NACC], eoff[NACC];
  double acc[NACC][TV];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int o = min(slot0 + a * spb, nslot - 1);
    coff[a] = (o / nch) * KT * CS;
    eoff[a] = (o % nch) * 4;
#pragma unroll
    for (int j = 0; j < TV; ++j) acc[a][j] = 0.0;
  }

  for (int q0 = ch.q_begin; q0 < ch.q_end; q0 += kUnitPairs) {
    const int np = min(kUnitPairs, ch.q_end - q0);
    if (q0 != ch.q_begin) __syncthreads();  // previous unit fully consumed
    STAMP(1);
    STAMP(2);
    // input rows -> cst (transposed) and, for S, the gathered eta rows -> es (row-major).  Every
    // thread fetches the item ids of its own elements itself (L1/L2 hits, no LDS hand-over and no
    // barrier between ids and rows) and all loads of a round -- two double2 of each table per thread
    // -- are in flight before any of them is stored to LDS.
    {
      const int tot_c = (abl & 1) ? 0 : np * dinp;
      const int tot_e = (DO_S && !(abl & 2)) ? np * doutp : 0;
      for (int t0 = tid * 2; t0 < max(tot_c, tot_e); t0 += nthr * 4) {
        double2 v[2], w[2];
        int pr[2], d[2], te[2];
        size_t row_c[2], row_e[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int t = min(t0 + j * nthr * 2, max(tot_c - 2, 0));
          pr[j] = t / dinp;
          d[j] = t - pr[j] * dinp;
          row_c[j] = GATHER ? static_cast<size_t>(pair_item[q0 + pr[j]]) : static_cast<size_t>(q0 + pr[j]);
          if (DO_S) {
            te[j] = min(t0 + j * nthr * 2, max(tot_e - 2, 0));
            row_e[j] = static_cast<size_t>(pair_item[q0 + te[j] / doutp]);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
          v[j] = *reinterpret_cast<const double2 *>(in_tab + row_c[j] * dinp + d[j]);
        if (DO_S) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            w[j] = *reinterpret_cast<const double2 *>(e_tab + row_e[j] * doutp + te[j] % doutp);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (t0 + j * nthr * 2 < tot_c) {
            cst[d[j] * CS + pr[j]] = v[j].x;
            cst[(d[j] + 1) * CS + pr[j]] = v[j].y;
          }
        }
        if (DO_S) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int t = t0 + j * nthr * 2;
            if (t < tot_e) *reinterpret_cast<double2 *>(es + t) = w[j];
          }
        }
      }
      if (np < kUnitPairs && !(abl & 1))  // ragged tail of a rating: zero the missing columns
        for (int t = tid; t < (kUnitPairs - np) * dinp; t += nthr)
          cst[(t / (kUnitPairs - np)) * CS + np + t % (kUnitPairs - np)] = 0.0;
    }
    STAMP(3);
    __syncthreads();
    STAMP(4);
    // ---- S: thread = (k, 4 l) slot, copies split the unit's pairs --------------------------------
    if (DO_S) {
      if (!(abl & 4) && s_active) {
#pragma unroll 2
        for (int j = sub; j < np; j += nsub) {
#pragma unroll
          for (int a = 0; a < NACC; ++a) {
            double cv[KT];
#pragma unroll
            for (int i = 0; i < KT; ++i) cv[i] = cst[coff[a] + i * CS + j];
            const double2 e0 = *reinterpret_cast<const double2 *>(es + j * doutp + eoff[a]);
            const double2 e1 = *reinterpret_cast<const double2 *>(es + j * doutp + eoff[a] + 2);
#pragma unroll
            for (int i = 0; i < KT; ++i) {
              acc[a][4 * i + 0] = fma(cv[i], e0.x, acc[a][4 * i + 0]);
              acc[a][4 * i + 1] = fma(cv[i], e0.y, acc[a][4 * i + 1]);
              acc[a][4 * i + 2] = fma(cv[i], e1.x, acc[a][4 * i + 2]);
              acc[a][4 * i + 3] = fma(cv[i], e1.y, acc[a][4 * i + 3]);
            }
          }
        }
      }
      if (!DIRECT) __syncthreads();  // es is dead: its space becomes tout
    }
    STAMP(5);
    // ---- mat-vec: lane = pair, wave = output chunk ------------------------------------------
    // (readfirstlane: tell the compiler the wave index is uniform so the tile loads scalarise)
    for (int c = __builtin_amdgcn_readfirstlane(wave); c < nch && !(abl & 8); c += nthr / 64) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      // rows d >= din of the tile and of the inputs are zero padding, dinp is a multiple of 4:
      // four rows' operands are fetched from LDS before any of them is used
      for (int d = 0; d < dinp; d += 4) {
        double x[4];
        double2 m0[4], m1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = cst[(d + i) * CS + lane];
          if (TLDS) {
            m0[i] = *reinterpret_cast<const double2 *>(tile_l + (d + i) * doutp + c * 4);
            m1[i] = *reinterpret_cast<const double2 *>(tile_l + (d + i) * doutp + c * 4 + 2);
          } else {
            const const_tile_ptr row = gtile + static_cast<size_t>(d + i) * doutp + c * 4;  // uniform
            m0[i].x = row[0]; m0[i].y = row[1];
            m1[i].x = row[2]; m1[i].y = row[3];
          }
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
      if (DIRECT) {  // each lane stores its 32 bytes of row q0 + lane straight from registers
        if (lane < np && !(abl & 16)) {
          double *dst = pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + lane), c * 4);
          *reinterpret_cast<double2 *>(dst) = w0;
          *reinterpret_cast<double2 *>(dst + 2) = w1;
        }
      } else {
        *reinterpret_cast<double2 *>(tout + lane * doutp + c * 4) = w0;
        *reinterpret_cast<double2 *>(tout + lane * doutp + c * 4 + 2) = w1;
      }
    }
    if (DIRECT) continue;
    __syncthreads();
    STAMP(6);
    if (!(abl & 16)) {  // the unit's 64 output rows are contiguous in memory: flat coalesced copy
      const int total = np * doutp;
      if (pa.out_mw == doutp && pa.out_rs_m == doutp) {  // plain table: the unit's rows are one block
        double *dst = out + static_cast<size_t>(q0) * doutp;
        for (int t = tid * 2; t < total; t += nthr * 2)
          *reinterpret_cast<double2 *>(dst + t) = *reinterpret_cast<const double2 *>(tout + t);
      } else {  // RowTab output (A): row by row, main part and tail part
        for (int t = tid * 2; t < total; t += nthr * 2) {
          const int pr = t / doutp, j = t - pr * doutp;
          *reinterpret_cast<double2 *>(pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + pr), j)) =
              *reinterpret_cast<const double2 *>(tout + t);
        }
      }
    }
  }
  STAMP(7);
  if (DO_S) {
    if (nsub > 1) {  // the other copies hand their sums over through LDS, added in copy order
      __syncthreads();
      if (s_active && sub > 0 && slot0 < nslot) {
#pragma unroll
        for (int j = 0; j < TV; ++j)  // [value][copy][slot]: consecutive lanes, consecutive words
          lds[(j * (nsub - 1) + sub - 1) * nslot + slot0] = acc[0][j];
      }
      __syncthreads();
      if (sub == 0 && slot0 < nslot) {
        for (int o = 1; o < nsub; ++o)
#pragma unroll
          for (int j = 0; j < TV; ++j) acc[0][j] += lds[(j * (nsub - 1) + o - 1) * nslot + slot0];
      }
    }
    if (sub == 0 && !(abl & 32)) {
      double *dst = partial + static_cast<size_t>(block) * dinp * doutp;
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        const int o = slot0 + a * spb;
        if (o < nslot) {
          double *cell = dst + (o / nch) * KT * doutp + eoff[a];
#pragma unroll
          for (int h = 0; h < KT; ++h) {
            double2 x, y;
            x.x = acc[a][4 * h]; x.y = acc[a][4 * h + 1]; y.x = acc[a][4 * h + 2]; y.y = acc[a][4 * h + 3];
            *reinterpret_cast<double2 *>(cell + h * doutp) = x;
            *reinterpret_cast<double2 *>(cell + h * doutp + 2) = y;
          }
        }
      }
    }
  }
#ifdef MMSBM_STAMPS
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(8);
#endif
}

constexpr int kPairBlockMax = 512;

constexpr int kQuadUnits = 4;  // 64-pair units a workgroup of pair_quad_a_kernel multiplies jointly

// (amdgpu_num_sgpr: a 256-thread workgroup is admitted floor(800 / (ceil(sgpr/16)*16 + 16)) times per
// CU -- 106 SGPRs: 6, 96: 7 (MI355X_MICROARCH.md, residency).  At C3 the stage has 1,565 workgroups:
// with 6 per CU (1,536 slots) 29 of them ran as a second round that doubled the launch's time.)
// (second launch bound: the small-tile instantiations must stay at 7 waves per SIMD, i.e. <= 72 VGPRs,
// for the same reason.)
template <bool GATHER, bool DO_S, int NACC, bool TLDS, int NT, int KT, bool DIRECT>
__global__ __launch_bounds__(NT, (NACC == 1 && KT == 2 && !TLDS) ? 7 : 1)
__attribute__((amdgpu_num_sgpr(96))) void pair_block_kernel(PairBlockArgs pa,
                                                        const double *__restrict__ tiles) {
  pair_block_body<GATHER, DO_S, NACC, TLDS, NT, KT, DIRECT>(pa, tiles, blockIdx.x);
}

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

template <int ROWS>
__device__ __forceinline__ void p_update_block(
    double (*red)[ROWS][kRedCols], int block, const double *__restrict__ partial,
    const int32_t *__restrict__ chunk_off, const double *__restrict__ p_old,
    double *__restrict__ p_new, double *__restrict__ pt_new, double *__restrict__ npr,
    int n_ratings, int kp, int lp, int normalize) {
  const int tx = threadIdx.x % kRedCols, ty = threadIdx.x / kRedCols;
  const int kl = kp * lp;
  const int col = block * kRedCols + tx;
  const bool ok = col < kl;
  double tot_all = 0.0;  // meaningful for ty == 0
  for (int r0 = 0; r0 < n_ratings; r0 += kRedGroup) {
    const int nr = min(kRedGroup, n_ratings - r0);
    int c0[kRedGroup], c1[kRedGroup];
    double s[kRedGroup], pold[kRedGroup];
    int longest = 0;
#pragma unroll
    for (int j = 0; j < kRedGroup; ++j)  // (needed at the very end: fetched up front, off the tail of the chain)
      pold[j] = (ty == 0 && ok && j < nr) ? p_old[static_cast<size_t>(r0 + j) * kl + col] : 0.0;
#pragma unroll
    for (int j = 0; j < kRedGroup; ++j) {
      const int r = min(r0 + j, n_ratings - 1);
      c0[j] = chunk_off[r];
      c1[j] = (j < nr) ? chunk_off[r + 1] : c0[j];
      longest = max(longest, c1[j] - c0[j]);
      s[j] = 0.0;
    }
    if (ok) {
      for (int off = ty; off < longest; off += ROWS * kRedBatch) {
        double v[kRedGroup][kRedBatch];
#pragma unroll
        for (int j = 0; j < kRedGroup; ++j)
#pragma unroll
          for (int i = 0; i < kRedBatch; ++i) {  // every rating's slab loads issued together
            const int c = c0[j] + off + i * ROWS;
            v[j][i] = (c < c1[j]) ? partial[static_cast<size_t>(c) * kl + col] : 0.0;
          }
#pragma unroll
        for (int j = 0; j < kRedGroup; ++j)
#pragma unroll
          for (int i = 0; i < kRedBatch; ++i) s[j] += v[j][i];
      }
    }
#pragma unroll
    for (int j = 0; j < kRedGroup; ++j) red[j][ty][tx] = s[j];
    __syncthreads();
    if (ty < 8) {  // ROWS rows -> 8 partial sums (fixed order)
#pragma unroll
      for (int j = 0; j < kRedGroup; ++j) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) t += red[j][ty * (ROWS / 8) + i][tx];
        s[j] = t;
      }
    }
    __syncthreads();
    if (ty < 8) {
#pragma unroll
      for (int j = 0; j < kRedGroup; ++j) red[j][ty][tx] = s[j];
    }
    __syncthreads();
    if (ty == 0 && ok) {
#pragma unroll
      for (int j = 0; j < kRedGroup; ++j) {
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
    __syncthreads();
    if (n_ratings <= kRedGroup) {  // common case: normalise straight from registers
      if (ty == 0 && ok && normalize) {
        const double den = (tot_all == 0.0) ? 1.0 : tot_all;
        const int k = col / lp, l = col % lp;
#pragma unroll
        for (int j = 0; j < kRedGroup; ++j) {
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
template <int G, int VEC>
__device__ __forceinline__ void item_sum_block(
    int block, const double *__restrict__ ttab, const int32_t *__restrict__ item_off,
    const int32_t *__restrict__ item_pairs, const int32_t *__restrict__ item_deg,
    const double *__restrict__ eta, double *__restrict__ eta_new, int n_items, int lp,
    int normalize, const int32_t *__restrict__ item_grid, int n_ratings) {
  constexpr int B = 8;
  const int it = block * (static_cast<int>(blockDim.x) / G) + threadIdx.x / G;
  const int gl = threadIdx.x % G;
  if (it >= n_items || gl * VEC >= lp) return;
  const int lane_off = gl * VEC;
  double acc[VEC], e[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
  load_vec<VEC>(eta + static_cast<size_t>(it) * lp + lane_off, e);
  if (item_grid) {
    // dense data (most (item, rating) combinations occur): the item's pairs sit in a fixed-width
    // grid row (-1: no such pair; ascending rating like the CSR list, so the sums are the same), one
    // dependent load level less than offsets -> pair ids -> rows
    const int32_t *row = item_grid + static_cast<size_t>(it) * n_ratings;
    for (int j = 0; j < n_ratings; j += B) {
      int id[B];
      double t[B][VEC];
#pragma unroll
      for (int b = 0; b < B; ++b) id[b] = row[min(j + b, n_ratings - 1)];
#pragma unroll
      for (int b = 0; b < B; ++b) load_vec<VEC>(ttab + static_cast<size_t>(max(id[b], 0)) * lp + lane_off, t[b]);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if (j + b < n_ratings && id[b] >= 0) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += t[b][v];
        }
      }
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
  const double d = static_cast<double>(max(item_deg[it], 1));
#pragma unroll
  for (int v = 0; v < VEC; ++v) e[v] = normalize ? (e[v] * acc[v]) / d : e[v] * acc[v];
  store_vec<VEC>(eta_new + static_cast<size_t>(it) * lp + lane_off, e);
}

// eta_p -- the two independent updates that follow the T / slab stage share ONE launch:
// blocks [0, nb_p) run p_update_block, the rest run item_sum_block.
struct EtaPArgs {
  const double *partial; const int32_t *chunk_off; const double *p_old;
  double *p_new; double *pt_new; double *npr;
  const double *ttab; const int32_t *item_off; const int32_t *item_pairs; const int32_t *item_deg;
  const double *eta; double *eta_new;
  int n_ratings, kp, lp, n_items, normalize, nb_p, abl;
  size_t bs_partial, bs_p, bs_t, bs_eta;  // restart slots (blockIdx.y): table strides
  const int32_t *item_grid;               // [n_items][n_ratings] pair id or -1 (dense data), else null
};

template <int G, int VEC>
__global__ __launch_bounds__(kRedThreads) void eta_p_kernel(EtaPArgs a) {
  __shared__ double red[kRedGroup][kRedRows][kRedCols];
  const size_t slot = blockIdx.y;
  if (a.abl & (static_cast<int>(blockIdx.x) < a.nb_p ? 64 : 128)) return;  // tuning aid: skip a role
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

// ======================================================================================
// once-per-run kernels: likelihood, prod_dist, compute_omegas (element-wise forms that
// follow the reference's association order)
// ======================================================================================
// src/expectation_maximization.py:157-167.  One thread per triple (original order).
__global__ __launch_bounds__(kBlock) void likelihood_kernel(
    const int32_t *__restrict__ tu, const int32_t *__restrict__ ti,
    const int32_t *__restrict__ tr, RowTab theta,
    const double *__restrict__ eta, const double *__restrict__ p, double *__restrict__ block_out,
    int64_t n_obs, int k_groups, int l_groups, int kp, int lp) {
  __shared__ double red[kBlock];
  double total = 0.0;
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; n < n_obs;
       n += static_cast<int64_t>(gridDim.x) * kBlock) {
    const size_t urow = static_cast<size_t>(tu[n]);
    const double *et = eta + static_cast<size_t>(ti[n]) * lp;
    const double *pr = p + static_cast<size_t>(tr[n]) * kp * lp;
    double s = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = *rowtab_ptr(theta, urow, k);
      for (int l = 0; l < l_groups; ++l) s += (tk * et[l]) * pr[k * lp + l];
    }
    const double ls = log(fmax(s, kEps));
    double acc = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = *rowtab_ptr(theta, urow, k);
      for (int l = 0; l < l_groups; ++l) {
        const double w = fmax((tk * et[l]) * pr[k * lp + l], kEps);
        acc += w * log(w) - w * ls;
      }
    }
    total += acc;
  }
  red[threadIdx.x] = total;
  __syncthreads();
  for (int h = kBlock / 2; h > 0; h >>= 1) {
    if (static_cast<int>(threadIdx.x) < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_out[blockIdx.x] = red[0];
}

// Faster form of the same sum for the usual sizes: one thread per triple in PAIR order, a
// workgroup per unit of <= 64 pairs of one rating.  Each thread parks its theta row and its
// pair's eta row in LDS (transposed: conflict-free column reads), the rating tile is
// lane-uniform and comes through scalar loads.  Element-wise formula and association order
// are the reference's; only the order of the outer sum differs.
constexpr int kLikThreads = 128;
static_assert(mmsbm::kMvChunkPairs == kUnitPairs, "likelihood units are built with kMvChunkPairs pairs");

__global__ __launch_bounds__(kLikThreads) void likelihood_units_kernel(
    const mmsbm::Chunk *__restrict__ units, const int32_t *__restrict__ pair_off,
    const int32_t *__restrict__ pair_user, const int32_t *__restrict__ pair_item, RowTab theta,
    const double *__restrict__ eta, const double *__restrict__ p, double *__restrict__ block_out,
    int k_groups, int l_groups, int kp, int lp) {
  extern __shared__ double lds[];
  double *ths = lds;                                          // [kp][kLikThreads]
  double *ets = lds + static_cast<size_t>(kp) * kLikThreads;  // [lp][kLikThreads]
  __shared__ int32_t poff[kUnitPairs + 4];  // (+4: keeps the dynamic LDS base 16-byte aligned)
  __shared__ double red[kLikThreads];
  const mmsbm::Chunk ch = units[blockIdx.x];
  const int tid = threadIdx.x;
  const int npairs = ch.q_end - ch.q_begin;
  if (tid <= npairs) poff[tid] = pair_off[ch.q_begin + tid];
  __syncthreads();
  typedef const double __attribute__((address_space(4))) * const_tile_ptr;
  const const_tile_ptr tile = (const_tile_ptr)(reinterpret_cast<uintptr_t>(
      p + static_cast<size_t>(ch.rating) * kp * lp));
  const int t0 = poff[0], t1 = poff[npairs];
  double total = 0.0;
  for (int base = t0; base < t1; base += kLikThreads) {
    const int n = base + tid;
    const bool have = n < t1;
    int lo = 0, hi = npairs;  // pair of triple n: last q with poff[q] <= n
    const int nn = have ? n : t1 - 1;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (poff[mid] <= nn) lo = mid; else hi = mid;
    }
    const size_t urow = static_cast<size_t>(pair_user[nn]);
    const double *erow = eta + static_cast<size_t>(pair_item[ch.q_begin + lo]) * lp;
    for (int k = 0; k < kp; k += 2) {
      const double2 v = *reinterpret_cast<const double2 *>(rowtab_ptr(theta, urow, k));
      ths[k * kLikThreads + tid] = v.x;
      ths[(k + 1) * kLikThreads + tid] = v.y;
    }
    for (int l = 0; l < lp; l += 2) {
      const double2 v = *reinterpret_cast<const double2 *>(erow + l);
      ets[l * kLikThreads + tid] = v.x;
      ets[(l + 1) * kLikThreads + tid] = v.y;
    }
    // own column only: no workgroup barrier needed between the writes above and the reads below
    double s = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = ths[k * kLikThreads + tid];
      for (int l = 0; l < l_groups; ++l) s += (tk * ets[l * kLikThreads + tid]) * tile[k * lp + l];
    }
    const double ls = log(fmax(s, kEps));
    double acc = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = ths[k * kLikThreads + tid];
      for (int l = 0; l < l_groups; ++l) {
        const double w = fmax((tk * ets[l * kLikThreads + tid]) * tile[k * lp + l], kEps);
        acc += w * log(w) - w * ls;
      }
    }
    if (have) total += acc;
  }
  red[tid] = total;
  __syncthreads();
  for (int h = kLikThreads / 2; h > 0; h >>= 1) {
    if (tid < h) red[tid] += red[tid + h];
    __syncthreads();
  }
  if (tid == 0) block_out[blockIdx.x] = red[0];
}

// The same likelihood without a logarithm per element.  With w = max(omega, eps) and
// ls = log max(s, eps), a triple contributes
//   sum_{omega >= eps} omega (log omega - ls)  +  #{omega < eps} * eps (log eps - ls)
// and log omega = log theta_k + log eta_l + log p_kl comes from tables of logarithms that
// log_table_kernel fills once per evaluation (U*K + I*L + R*K*L logs instead of N*K*L).  One
// pass gathers A = sum omega log omega, W = sum omega over the unclamped elements, their count
// and s; ls enters at the end: A - ls W + count eps (log eps - ls).  The element sum is
// re-associated relative to the reference (agreement ~1e-15 relative), the formula is not
// changed.  G lanes share a triple, each holding LW columns of the eta row and of its logarithms
// in registers; the rating's tile (and its logarithms) is lane-uniform for G = 1 (scalar loads)
// or sits in LDS.
__global__ __launch_bounds__(kBlock) void log_table_kernel(const double *__restrict__ in,
                                                           double *__restrict__ out, size_t n) {
  const size_t e = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e < n) out[e] = log(in[e]);  // log(0) = -inf belongs to elements that are clamped, never used
}

// the same for a RowTab (theta): `out` has the one-slot layout whatever `in` has
__global__ __launch_bounds__(kBlock) void log_rows_kernel(RowTab in, RowTab out, size_t rows, int dp) {
  const size_t e = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= rows * dp) return;
  const size_t row = e / dp;
  const int off = static_cast<int>(e - row * dp);
  *rowtab_ptr(out, row, off) = log(*rowtab_ptr(in, row, off));
}

template <int LW, int G, bool TLDS>
__global__ __launch_bounds__(kLikThreads) void likelihood_fast_kernel(
    const mmsbm::Chunk *__restrict__ units, const int32_t *__restrict__ pair_off,
    const int32_t *__restrict__ pair_user, const int32_t *__restrict__ pair_item, RowTab theta,
    RowTab ltheta, const double *__restrict__ eta, const double *__restrict__ leta,
    const double *__restrict__ p, const double *__restrict__ logp, double *__restrict__ block_out,
    int k_groups, int l_groups, int kp, int lp) {
  extern __shared__ double lds[];  // TLDS: [kp*lp] tile, [kp*lp] its logarithms
  __shared__ int32_t poff[kUnitPairs + 4];
  __shared__ double red[kLikThreads];
  const mmsbm::Chunk ch = units[blockIdx.x];
  const int tid = threadIdx.x;
  const int npairs = ch.q_end - ch.q_begin;
  if (tid <= npairs) poff[tid] = pair_off[ch.q_begin + tid];
  const size_t toff = static_cast<size_t>(ch.rating) * kp * lp;
  if (TLDS) {
    for (int t = tid * 2; t < kp * lp; t += kLikThreads * 2) {
      *reinterpret_cast<double2 *>(lds + t) = *reinterpret_cast<const double2 *>(p + toff + t);
      *reinterpret_cast<double2 *>(lds + kp * lp + t) = *reinterpret_cast<const double2 *>(logp + toff + t);
    }
  }
  __syncthreads();
  typedef const double __attribute__((address_space(4))) * const_tile_ptr;
  const const_tile_ptr gtile = (const_tile_ptr)(reinterpret_cast<uintptr_t>(p + toff));
  const const_tile_ptr gltile = (const_tile_ptr)(reinterpret_cast<uintptr_t>(logp + toff));
  constexpr int TPB = kLikThreads / G;  // triples per round
  const int grp = tid / G, g = tid % G;
  // lane g of a group owns the column PAIRS 2g, 2g + 2G, 2g + 4G, ...: one (16-byte) read instruction of a
  // group then covers 2G consecutive tile entries in LDS.  (Blocks of LW consecutive columns per lane put the
  // lanes 8 LW bytes apart -- a two-way bank conflict on every tile read: SQ_LDS_BANK_CONFLICT was twice
  // SQ_ACTIVE_INST_LDS at C5; single columns g, g + G, ... are conflict-free too but cannot be read as
  // 16-byte pairs: 52 vs 39 ms.)
#define LIK_COL(j) (2 * g + ((j) & 1) + 2 * G * ((j) >> 1))
  const double log_eps = log(kEps);
  const int t0 = poff[0], t1 = poff[npairs];
  double total = 0.0;
  for (int base = t0; base < t1; base += TPB) {
    const int n = base + grp;
    const bool have = n < t1;
    const int nn = have ? n : t1 - 1;
    int lo = 0, hi = npairs;  // pair of triple nn: last q with poff[q] <= nn
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (poff[mid] <= nn) lo = mid; else hi = mid;
    }
    const size_t urow = static_cast<size_t>(pair_user[nn]);
    const size_t irow = static_cast<size_t>(pair_item[ch.q_begin + lo]);
    double e[LW], le[LW];
#pragma unroll
    for (int j = 0; j < LW; j += 2) {  // (columns past lp: any in-range address, masked below)
      const int cc = min(LIK_COL(j), lp - 2);
      const double2 v = *reinterpret_cast<const double2 *>(eta + irow * lp + cc);
      const double2 lv = *reinterpret_cast<const double2 *>(leta + irow * lp + cc);
      e[j] = v.x; e[j + 1] = v.y;
      le[j] = lv.x; le[j + 1] = lv.y;
    }
    double s = 0.0, a_sum = 0.0, w_sum = 0.0, clamped = 0.0;
    for (int k = 0; k < k_groups; ++k) {
      const double tk = *rowtab_ptr(theta, urow, k);
      const double ltk = *rowtab_ptr(ltheta, urow, k);
#pragma unroll
      for (int j = 0; j < LW; ++j) {
        const int l = LIK_COL(j);
        const bool real = l < l_groups;
        const int lc = min(l, lp - 1);
        double pv, lpv;
        if (TLDS) {
          pv = lds[k * lp + lc];
          lpv = lds[kp * lp + k * lp + lc];
        } else {  // G == 1: l is the same for every lane
          pv = gtile[k * lp + lc];
          lpv = gltile[k * lp + lc];
        }
        const double w = (tk * e[j]) * pv;
        const bool big = real && w >= kEps;
        s += real ? w : 0.0;
        a_sum += big ? w * ((ltk + le[j]) + lpv) : 0.0;
        w_sum += big ? w : 0.0;
        clamped += (real && !big) ? 1.0 : 0.0;
      }
    }
    s = group_sum<G>(s);
    a_sum = group_sum<G>(a_sum);
    w_sum = group_sum<G>(w_sum);
    clamped = group_sum<G>(clamped);
    const double ls = log(fmax(s, kEps));
    if (have && g == 0) total += (a_sum - ls * w_sum) + clamped * (kEps * (log_eps - ls));
  }
  red[tid] = total;
  __syncthreads();
  for (int h = kLikThreads / 2; h > 0; h >>= 1) {
    if (tid < h) red[tid] += red[tid + h];
    __syncthreads();
  }
  if (tid == 0) block_out[blockIdx.x] = red[0];
}
#undef LIK_COL

// ======================================================================================
// Initial parameters on the device (src/mmsbm.py:224-233): table[row][j] = U / degree(row) with
// U the (offset + row*d + j)-th double of the restart's PCG64 stream -- bit for bit what
// ``default_rng(child_seed).random((rows, d)) / degree`` gives on the host.  Each thread jumps
// the stream to its own 8 consecutive draws.
// ======================================================================================
constexpr int kDrawsPerThread = 8;

__global__ __launch_bounds__(kBlock) void init_rows_kernel(
    RowTab out, const int32_t *__restrict__ off, const int32_t *__restrict__ deg, int rows, int d,
    uint64_t s_hi, uint64_t s_lo, uint64_t i_hi, uint64_t i_lo, uint64_t stream_offset) {
  const uint64_t total = static_cast<uint64_t>(rows) * d;
  const uint64_t f0 = (static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x) * kDrawsPerThread;
  if (f0 >= total) return;
  pcg64::Stream g{pcg64::make128(s_hi, s_lo), pcg64::make128(i_hi, i_lo)};
  pcg64::advance(g, stream_offset + f0);
  int row = static_cast<int>(f0 / d), j = static_cast<int>(f0 % d);
  for (int e = 0; e < kDrawsPerThread && f0 + e < total; ++e) {
    const int cnt = off ? off[row + 1] - off[row] : deg[row];  // rows of this user / item
    *rowtab_ptr(out, static_cast<size_t>(row), j) = pcg64::next_double(g) / static_cast<double>(max(cnt, 1));
    if (++j == d) {
      j = 0;
      ++row;
    }
  }
}

// P[m,r] = sum_kl theta[u,k] eta[i,l] p[k,l,r] for one (row, rating): src/kernels_numpy.py:94-96.
__device__ __forceinline__ double prod_dist_elem(const RowTab &theta, size_t urow,
                                                 const double *__restrict__ et,
                                                 const double *__restrict__ pr, int k_groups,
                                                 int l_groups, int lp) {
  double acc = 0.0;
  for (int k = 0; k < k_groups; ++k) {
    const double tk = *rowtab_ptr(theta, urow, k);
    double inner = 0.0;
    for (int l = 0; l < l_groups; ++l) inner = fma(et[l], pr[k * lp + l], inner);
    acc = fma(tk, inner, acc);
  }
  return acc;
}

// src/kernels_numpy.py:86-96.  One thread per (pair, rating).
__global__ __launch_bounds__(kBlock) void prod_dist_kernel(
    const int32_t *__restrict__ pu, const int32_t *__restrict__ pi,
    RowTab theta, const double *__restrict__ eta,
    const double *__restrict__ p, double *__restrict__ out, int64_t n_pairs, int n_ratings,
    int k_groups, int l_groups, int kp, int lp) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n_pairs * n_ratings) return;
  const int64_t m = e / n_ratings;
  const int r = static_cast<int>(e % n_ratings);
  const size_t urow = static_cast<size_t>(pu[m]);
  const double *et = eta + static_cast<size_t>(pi[m]) * lp;
  const double *pr = p + static_cast<size_t>(r) * kp * lp;
  out[e] = prod_dist_elem(theta, urow, et, pr, k_groups, l_groups, lp);
}

// ======================================================================================
// predict / score on the device (src/mmsbm.py:297-315 and 488-539): the rating distribution
// of every test row for ONE restart is added into a running sum (restart order = call order,
// the order numpy's mean over the restart axis adds in) and reduced on the spot to the
// reference's indicators, so only six numbers per restart travel back:
//   [0] rows kept (distribution not all zero)   [1] argmax == real   [2] |argmax - real| <= 1
//   [3] sum |argmax - real|   [4] real == round(P . w)   [5] sum |P . w - real|
// One thread per test row; fixed-order tree per workgroup, workgroup sums added by the host in
// block order.  FINISH: the distribution is the running sum divided by the number of restarts
// (written back in place), nothing new is computed.
// ======================================================================================
constexpr int kScoreStats = 6;

template <bool FINISH>
__global__ __launch_bounds__(kBlock) void predict_score_kernel(
    const int32_t *__restrict__ pu, const int32_t *__restrict__ pi, const int32_t *__restrict__ preal,
    RowTab theta, const double *__restrict__ eta, const double *__restrict__ p,
    const double *__restrict__ weights, double *__restrict__ sum, double *__restrict__ block_out,
    int64_t n_rows, int n_ratings, int k_groups, int l_groups, int kp, int lp, int first,
    double n_added) {
  __shared__ double red[kScoreStats][kBlock];
  const int64_t m = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  double st[kScoreStats];
#pragma unroll
  for (int j = 0; j < kScoreStats; ++j) st[j] = 0.0;
  if (m < n_rows) {
    const size_t urow = static_cast<size_t>(FINISH ? 0 : pu[m]);
    const double *et = FINISH ? nullptr : eta + static_cast<size_t>(pi[m]) * lp;
    double *srow = sum + m * n_ratings;
    int best = 0;
    double bestv = 0.0, tot = 0.0, pond = 0.0;
    for (int r = 0; r < n_ratings; ++r) {
      double v;
      if (FINISH) {
        v = srow[r] / n_added;
        srow[r] = v;
      } else {
        v = prod_dist_elem(theta, urow, et, p + static_cast<size_t>(r) * kp * lp, k_groups, l_groups, lp);
        srow[r] = first ? v : srow[r] + v;
      }
      if (r == 0 || v > bestv) {  // np.argmax: the first maximum
        bestv = v;
        best = r;
      }
      tot += v;
      pond += v * weights[r];
    }
    if (tot != 0.0) {  // src/mmsbm.py:505-510: rows whose distribution is all zero are dropped
      const int real = preal[m];
      const int dist = abs(best - real);
      st[0] = 1.0;
      st[1] = dist == 0 ? 1.0 : 0.0;
      st[2] = dist <= 1 ? 1.0 : 0.0;
      st[3] = static_cast<double>(dist);
      st[4] = (static_cast<double>(real) == rint(pond)) ? 1.0 : 0.0;  // np.round: half to even
      st[5] = fabs(pond - static_cast<double>(real));
    }
  }
#pragma unroll
  for (int j = 0; j < kScoreStats; ++j) red[j][threadIdx.x] = st[j];
  __syncthreads();
  for (int h = kBlock / 2; h > 0; h >>= 1) {
    if (static_cast<int>(threadIdx.x) < h) {
#pragma unroll
      for (int j = 0; j < kScoreStats; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + h];
    }
    __syncthreads();
  }
  if (threadIdx.x < kScoreStats) block_out[blockIdx.x * kScoreStats + threadIdx.x] = red[threadIdx.x][0];
}

// ======================================================================================
// prod_dist / predict through the factorisation (round 2).  P[m, r] = theta_u . (p_r eta_i): the inner
// vector B[(i, r), :] = p_r eta_i is the A launch's mat-vec over EVERY (item, rating) combination (a
// rating-major pair list q = r I + i built once per context), after which a test row costs R dot
// products of length K instead of R K L multiply-adds behind dependent loads: 1M rows at K = L = 50,
// R = 10 took 357 ms per restart in predict_score_kernel, now the B launch (0.2 ms) plus this kernel.
// A group of G lanes per test row (lane gl owns VEC entries, as in seg_pass); MODE 0 writes the
// distribution (prod_dist), MODE 1 adds it to the session's running sum and reduces the restart's six
// indicator sums exactly as predict_score_kernel does (fixed-order tree over the workgroup's rows).
// ======================================================================================
template <int G, int VEC, int MODE>
__global__ __launch_bounds__(kBlock) void predict_rows_kernel(
    const int32_t *__restrict__ pu, const int32_t *__restrict__ pi, const int32_t *__restrict__ preal,
    RowTab theta, const double *__restrict__ btab, size_t rating_stride, const double *__restrict__ weights,
    double *__restrict__ dist, double *__restrict__ block_out, int64_t n_rows, int n_ratings, int dp,
    int first) {
  constexpr int PER = kBlock / G;
  __shared__ double red[kScoreStats][PER];
  const int gl = threadIdx.x % G, grp = threadIdx.x / G;
  const int64_t m = static_cast<int64_t>(blockIdx.x) * PER + grp;
  double st[kScoreStats];
#pragma unroll
  for (int j = 0; j < kScoreStats; ++j) st[j] = 0.0;
  if (m < n_rows) {  // (whole groups)
    const bool act = gl * VEC < dp;
    const int lane_off = act ? gl * VEC : 0;
    double f[VEC];
    load_vec<VEC>(rowtab_ptr(theta, static_cast<size_t>(pu[m]), lane_off), f);
    if (!act) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) f[v] = 0.0;
    }
    const double *brow = btab + static_cast<size_t>(pi[m]) * dp + lane_off;
    double *srow = dist + m * n_ratings;
    int best = 0;
    double bestv = 0.0, tot = 0.0, pond = 0.0;
    constexpr int RB = 4;  // ratings whose rows are in flight together
    for (int r0 = 0; r0 < n_ratings; r0 += RB) {
      double g[RB][VEC];
#pragma unroll
      for (int b = 0; b < RB; ++b) load_vec<VEC>(brow + static_cast<size_t>(min(r0 + b, n_ratings - 1)) * rating_stride, g[b]);
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const int r = r0 + b;
        if (r < n_ratings) {
          double pt = 0.0;
#pragma unroll
          for (int v = 0; v < VEC; ++v) pt = fma(g[b][v], f[v], pt);
          const double v = group_sum<G>(pt);
          if (MODE == 0) {
            if (gl == 0) srow[r] = v;
          } else {
            if (gl == 0) srow[r] = first ? v : srow[r] + v;
            if (r == 0 || v > bestv) {  // np.argmax: the first maximum
              bestv = v;
              best = r;
            }
            tot += v;
            pond += v * weights[r];
          }
        }
      }
    }
    if (MODE == 1 && gl == 0 && tot != 0.0) {  // src/mmsbm.py:505-510: all-zero rows are dropped
      const int real = preal[m];
      const int dd = abs(best - real);
      st[0] = 1.0;
      st[1] = dd == 0 ? 1.0 : 0.0;
      st[2] = dd <= 1 ? 1.0 : 0.0;
      st[3] = static_cast<double>(dd);
      st[4] = (static_cast<double>(real) == rint(pond)) ? 1.0 : 0.0;  // np.round: half to even
      st[5] = fabs(pond - static_cast<double>(real));
    }
  }
  if (MODE == 0) return;
  if (gl == 0) {
#pragma unroll
    for (int j = 0; j < kScoreStats; ++j) red[j][grp] = st[j];
  }
  __syncthreads();
  for (int h = PER / 2; h > 0; h >>= 1) {
    if (gl == 0 && grp < h) {
#pragma unroll
      for (int j = 0; j < kScoreStats; ++j) red[j][grp] += red[j][grp + h];
    }
    __syncthreads();
  }
  if (threadIdx.x < kScoreStats) block_out[blockIdx.x * kScoreStats + threadIdx.x] = red[threadIdx.x][0];
}

// src/kernels_numpy.py:21-36.  One thread per element; sk/sl = output strides of the
// internal (k, l) indices (they differ from (L, 1) when the sides are swapped).
__global__ __launch_bounds__(kBlock) void omegas_kernel(
    const int32_t *__restrict__ tu, const int32_t *__restrict__ ti,
    const int32_t *__restrict__ tr, RowTab theta,
    const double *__restrict__ eta, const double *__restrict__ p, double *__restrict__ out,
    int64_t n_elems, int k_groups, int l_groups, int kp, int lp, int sk, int sl) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n_elems) return;
  const int kl = k_groups * l_groups;
  const int64_t n = e / kl;
  const int rem = static_cast<int>(e % kl);
  const int k = rem / l_groups, l = rem % l_groups;
  const double v = (*rowtab_ptr(theta, static_cast<size_t>(tu[n]), k) *
                    eta[static_cast<size_t>(ti[n]) * lp + l]) *
                   p[static_cast<size_t>(tr[n]) * kp * lp + k * lp + l];
  out[n * kl + static_cast<int64_t>(k) * sk + static_cast<int64_t>(l) * sl] = v;
}

// ======================================================================================
// host side
// ======================================================================================
int pad_dim(int d) {  // multiples of 4: 32-byte row granules, whole chunks of 4 outputs
  if (d <= 256) return (d + 3) / 4 * 4;
  if (d <= 512) return (d + 7) / 8 * 8;
  return (d + 15) / 16 * 16;
}

// (G, VEC) instantiation for a padded row length: code 0..6
// Four doubles (32 bytes) per lane: fewer lanes per row means more rows per wave instruction,
// i.e. less vector-ALU work (dot product, DPP reduction, division) per triple.
int group_code(int dp) {
  if (dp <= 16) return 0;   // G=4  VEC=4
  if (dp <= 32) return 1;   // G=8  VEC=4
  if (dp <= 64) return 2;   // G=16 VEC=4
  if (dp <= 128) return 3;  // G=32 VEC=4
  if (dp <= 256) return 4;  // G=64 VEC=4
  if (dp <= 512) return 5;  // G=64 VEC=8
  return 6;                 // G=64 VEC=16 (up to 1,024 groups)
}
int group_lanes(int code) {
  static const int g[7] = {4, 8, 16, 32, 64, 64, 64};
  return g[code];
}

#define DISPATCH_GV(code, CALL)                                   \
  switch (code) {                                                 \
    case 0: CALL(4, 4); break;                                    \
    case 1: CALL(8, 4); break;                                    \
    case 2: CALL(16, 4); break;                                   \
    case 3: CALL(32, 4); break;                                   \
    case 4: CALL(64, 4); break;                                   \
    case 5: CALL(64, 8); break;                                   \
    default: CALL(64, 16); break;                                 \
  }

template <class T>
struct DevBuf {
  T *ptr = nullptr;
  size_t count = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  void alloc(size_t n) {
    release();
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ptr), std::max<size_t>(n, 1) * sizeof(T)));
    count = n;  // (only once the memory is there: a failed allocation leaves an empty buffer)
  }
  void upload(const std::vector<T> &h, hipStream_t s) {
    alloc(h.size());
    if (!h.empty())
      HIP_CHECK(hipMemcpyAsync(ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
};

// A per-restart table: `slots` copies, `stride` doubles apart (whole 128-byte lines, so every
// copy keeps the alignment of the first).
struct SlotBuf : DevBuf<double> {
  size_t stride = 0;
  void alloc_slots(size_t per_slot, int slots) {
    stride = (per_slot + 15) / 16 * 16;
    alloc(stride * static_cast<size_t>(slots));
  }
  double *at(int slot) const { return ptr + static_cast<size_t>(slot) * stride; }
};

// Pinned host staging: parameter rows travel as ONE contiguous copy in the device layout
// (packed / unpacked on the host by a few threads) instead of strided 2-D copies from
// pageable memory.
struct PinBuf {
  double *ptr = nullptr;
  size_t cap = 0, used = 0;
  PinBuf() = default;
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() {
    if (ptr) (void)hipHostFree(ptr);
  }
  void reset(size_t need) {
    used = 0;
    if (need <= cap) return;
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    cap = 0;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ptr), need * sizeof(double), hipHostMallocDefault));
    cap = need;
  }
  double *take(size_t n) {
    double *r = ptr + used;
    used += n;
    return r;
  }
};

// fn(first_row, last_row) over [0, rows), on up to 8 host threads when the table is large
template <class F>
void for_row_blocks(int rows, size_t row_doubles, F &&fn) {
  const size_t total = static_cast<size_t>(rows) * row_doubles;
  unsigned nt = total < (size_t(1) << 19) ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (nt <= 1) {
    fn(0, rows);
    return;
  }
  std::vector<std::thread> th;
  const int per = (rows + static_cast<int>(nt) - 1) / static_cast<int>(nt);
  for (unsigned t = 0; t < nt; ++t) {
    const int a = static_cast<int>(t) * per, b = std::min(rows, a + per);
    if (a < b) th.emplace_back([&fn, a, b] { fn(a, b); });
  }
  for (auto &x : th) x.join();
}

enum KernelId { K_SEG = 0, K_DENSE, K_ETAP, K_MATVEC_A, K_COUNT };
// The four launches of an iteration.
const char *const kKernelNames[K_COUNT] = {"seg_pass_kernel", "pair_block_kernel(T+S)",
                                           "eta_p_kernel", "pair_block_kernel(A)"};

}  // namespace

struct mmsbm_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool swapped = false;
  // external dims
  int64_t n_obs = 0;
  int ext_users = 0, ext_items = 0, ext_k = 0, ext_l = 0;
  // internal dims ("item" = the side paired with the rating)
  int n_users = 0, n_items = 0, n_ratings = 0, k = 0, l = 0, kp = 0, lp = 0;
  int n_pairs = 0, n_chunks = 0;
  int code_k = 0, code_l = 0;
  bool direct_out = false;  // pair_block: output rows stored straight from registers (no LDS transpose)
  int ablate = 0;           // tuning aid (mmsbm_hip_time_stage): phases pair_block skips
  bool split_rows = false;  // theta and A kept as 128-byte main lines + tail rows (RowTab)
  int pb_threads_t = kBlock, pb_threads_a = kBlock;  // pair_block workgroup sizes (T+S mode, A mode)
  int pb_kt = 4;  // pair_block S phase: k-rows per register tile (2 when K x L is small)
  int pb_spb = kBlock, pb_nacc = 1, pb_nsub = 1;  // pair_block S phase: threads per slot-grid copy, slots per thread
  size_t lds_t = 0, lds_a = 0;
  bool tl_t = false, tl_a = false;  // rating tile staged in LDS (T+S launch / A launch)
  bool quad_a = false;  // the A launch runs pair_quad_a_kernel (long rows)
  // prod_dist / predict through B[(item, rating), :] = p_r eta_i (predict_rows_kernel)
  DevBuf<int32_t> grid_item;          // item of pair q = r * I + i, every (item, rating) combination
  DevBuf<mmsbm::Chunk> grid_chunks;   // its rating-homogeneous chunks
  int grid_n_chunks = 0, mv_chunk_pairs = mmsbm::kMvChunkPairs;
  DevBuf<double> btab;                // [I * R][kp], of the slot being scored
  bool predict_fast = true;
  bool mfma = false;    // both pair-stage launches run pair_mfma_kernel (tiles beyond the scalar cache, K, L <= 64)
  size_t lds_mt = 0, lds_ma = 0;
  int mfma_threads = kPairBlockMax;  // T+S launch: 512 (eight waves) or 256
  bool mfma_big = false;  // K or L beyond 64: the blocked forms (mfma_rows_kernel + mfma_slab_kernel)
  bool wide = false;    // K, L beyond the LDS stage: wide_matvec / wide_slab kernels (any size)
  bool slot_waves = true;  // several slots: one super-group of lanes walks a segment for all of them
  int ranges_pairs = 1, ranges_users = 1;  // XCD-local work lists: ranges the gathered table is cut into
  int n_cus = 256;
  size_t lds_qa = 0;
  mmsbm::Layout lay;  // host copy (degrees, sizes)
  DevBuf<int32_t> pair_off, pair_user, pair_item, user_off, user_pair, item_off, item_pairs,
      item_deg, mv_chunk_off, orig_u, orig_i, orig_r;
  DevBuf<int32_t> item_grid;  // [n_items][n_ratings] pair ids (-1: none); only for dense (item, rating) grids
  DevBuf<mmsbm::Chunk> mv_chunks;
  DevBuf<mmsbm::Chunk> lik_units;  // 64-pair units for the likelihood kernel (mv_chunks may hold 256)
  int n_lik_units = 0;
  DevBuf<mmsbm::WorkItem> pair_items, user_items;   // only when some segment is long
  DevBuf<mmsbm::SplitSeg> pair_splits, user_splits;
  // Per-restart state, one copy per slot.  A context carries n_slots independent restarts
  // (parameter sets) over the SAME triples; em_iterate advances all of them with one set of
  // launches (blockIdx.y = slot), the single-restart entry points act on slot `sel`.
  int n_slots = 1, sel = 0;
  int base_slot = 0, launch_slots = 1;  // what the next launches cover: [base_slot, base_slot + launch_slots)
  SlotBuf pair_parts, user_parts;
  SlotBuf theta[2], eta[2], p[2], pt[2], atab[2], ctab, ttab, partial, npr;
  DevBuf<double> lik_part;
  DevBuf<double> lg_theta, lg_eta, lg_p;  // logarithm tables of the selected slot (likelihood)
  bool lik_fast = true;                   // option "lik_fast": 0 = the log-per-element kernels
  int lik_g = 0;                          // option "lik_g": lanes per triple (0 = automatic)
  PinBuf pin;  // host staging for set_params / get_params / update_coefficients
  // predict/score session (mmsbm_hip_predict_begin .. finish)
  DevBuf<int32_t> ps_u, ps_i, ps_r;
  DevBuf<double> ps_sum, ps_w, ps_part;
  int64_t ps_rows = -1;  // -1: no session open
  int ps_added = 0;
  int cur = 0;
  std::vector<char> have;  // per slot: set_params has been called
  bool graph_mode = false;  // replay a captured two-iteration hipGraph instead of eager launches
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};  // indexed by `cur` at capture time
  // per-launch profiling
  bool profiling = false;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;

  void drop_graphs() {
    for (auto &g : graph_exec) {
      if (g) (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
  }
  ~mmsbm_hip_ctx() {
    drop_graphs();
    for (auto &pe : prof_events) {
      (void)hipEventDestroy(pe.second.first);
      (void)hipEventDestroy(pe.second.second);
    }
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

struct LaunchScope {  // optional event pair around one launch
  mmsbm_hip_ctx *c;
  int id;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  LaunchScope(mmsbm_hip_ctx *ctx, int kid) : c(ctx), id(kid) {
    if (c->profiling) {
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      HIP_CHECK(hipEventRecord(e0, c->stream));
    }
  }
  void done() {
    HIP_CHECK(hipGetLastError());
    if (c->profiling) {
      HIP_CHECK(hipEventRecord(e1, c->stream));
      c->prof_events.push_back({id, {e0, e1}});
    }
  }
};

void use_device(const mmsbm_hip_ctx *c) { HIP_CHECK(hipSetDevice(c->device)); }

// pair_quad_a -- the A launch for long rows (K, L ~ 50; the tile lives in LDS).  There the lane-per-
// pair mat-vec of pair_block is bound by the LDS pipe: one broadcast ds_read_b128 of the tile per
// two FMAs.  Here a workgroup (8 waves, one per CU: ~130 KB of LDS) stages the transposed input
// rows of ALL four 64-pair units of a chunk, so that every tile value read from LDS feeds four
// pairs per lane (16 FMAs per 4 + 2 LDS reads instead of 4 per 1 + 2), and it is a PERSISTENT
// pipeline: it walks chunks blockIdx.x, + gridDim.x, ...; while it multiplies chunk i from LDS,
// the gathered rows of chunk i+1 are already on their way into registers and the item ids of
// chunk i+2 behind them, so the dependent round trips (ids -> rows) are paid once per workgroup,
// not once per chunk.  Output rows go to memory straight from registers.
// NL = double2 per thread per chunk: 256 pairs x dinp entries / 2 / 512 threads = dinp / 4
template <int NL>
__global__ __launch_bounds__(kPairBlockMax) void pair_quad_a_kernel(PairBlockArgs pa,
                                                                    const double *__restrict__ tiles0,
                                                                    int n_chunks) {
  constexpr int NT = kPairBlockMax;
  const size_t slot = blockIdx.y;
  const double *__restrict__ tiles = tiles0 + slot * pa.bs_tiles;
  const double *__restrict__ in_tab = pa.in_tab + slot * pa.bs_in;
  const int32_t *__restrict__ pair_item = pa.pair_item;
  double *__restrict__ out = pa.out + slot * pa.bs_out;
  double *__restrict__ out_tail = pa.out_tail + slot * pa.bs_out_t;
  const int dinp = pa.dinp, doutp = pa.doutp;
  extern __shared__ double lds[];
  constexpr int CS = kUnitPairs + 1;
  const int ustride = dinp * CS;
  double *cst = lds;                                                 // [4][dinp][CS]
  double *tile_l = cst + static_cast<size_t>(kQuadUnits) * ustride;  // [dinp][doutp]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nch = doutp >> 2;
  const int stride = gridDim.x;
  int ci = blockIdx.x;
  if (ci >= n_chunks) return;

  // this thread's share of a chunk: elements t_j = 2 tid + j * 2 NT of the flat (pair, entry) space
  int pd[NL];  // (pair within the chunk) << 8 | entry   (the pair may be >= the chunk's size: masked by `total`)
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int t = tid * 2 + j * NT * 2;
    const int pr = t / dinp;
    pd[j] = (pr << 8) | (t - pr * dinp);
  }
#define PRJ(j) (pd[j] >> 8)
#define DJ(j) (pd[j] & 255)
  mmsbm::Chunk ch = pa.chunks[ci];
  int ids[NL];
  double2 v[NL];
  {
    const int total = (ch.q_end - ch.q_begin) * dinp;
#pragma unroll
    for (int j = 0; j < NL; ++j)
      ids[j] = (tid * 2 + j * NT * 2 < total) ? pair_item[ch.q_begin + PRJ(j)] : 0;
#pragma unroll
    for (int j = 0; j < NL; ++j)
      v[j] = *reinterpret_cast<const double2 *>(in_tab + static_cast<size_t>(ids[j]) * dinp + DJ(j));
  }
  bool has_next = ci + stride < n_chunks;
  mmsbm::Chunk nx = has_next ? pa.chunks[ci + stride] : ch;
  if (has_next) {
    const int total = (nx.q_end - nx.q_begin) * dinp;
#pragma unroll
    for (int j = 0; j < NL; ++j)
      ids[j] = (tid * 2 + j * NT * 2 < total) ? pair_item[nx.q_begin + PRJ(j)] : 0;
  }
  int tile_rating = -1;
  while (true) {
    const int np_all = ch.q_end - ch.q_begin;
    const int total = np_all * dinp;
    __syncthreads();  // the previous chunk's mat-vec is done with cst and the tile
    if (ch.rating != tile_rating) {
      const double *src = tiles + static_cast<size_t>(ch.rating) * dinp * doutp;
      for (int t = tid * 2; t < dinp * doutp; t += NT * 2)
        *reinterpret_cast<double2 *>(tile_l + t) = *reinterpret_cast<const double2 *>(src + t);
      tile_rating = ch.rating;
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {  // (columns of pairs beyond np_all keep stale data: their outputs are never stored)
      if (tid * 2 + j * NT * 2 < total) {
        double *dst = cst + (PRJ(j) >> 6) * ustride + DJ(j) * CS + (PRJ(j) & 63);
        dst[0] = v[j].x;
        dst[CS] = v[j].y;
      }
    }
    __syncthreads();
    // prefetch: rows of the next chunk (its ids arrived during the previous iteration), then the
    // ids of the chunk after that
    const bool has_next2 = ci + 2 * stride < n_chunks;
    mmsbm::Chunk nn = nx;
    if (has_next) {
#pragma unroll
      for (int j = 0; j < NL; ++j)
        v[j] = *reinterpret_cast<const double2 *>(in_tab + static_cast<size_t>(ids[j]) * dinp + DJ(j));
      if (has_next2) {
        nn = pa.chunks[ci + 2 * stride];
        const int tot2 = (nn.q_end - nn.q_begin) * dinp;
#pragma unroll
        for (int j = 0; j < NL; ++j)
          ids[j] = (tid * 2 + j * NT * 2 < tot2) ? pair_item[nn.q_begin + PRJ(j)] : 0;
      }
    }
    // ---- mat-vec over the four units at once: lane = pair (of each unit), wave = output chunk ----
    for (int c = __builtin_amdgcn_readfirstlane(wave); c < nch; c += NT / 64) {
      double a[kQuadUnits][4];
#pragma unroll
      for (int u = 0; u < kQuadUnits; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[u][j] = 0.0;
      for (int d = 0; d < dinp; d += 2) {
        double2 m0[2], m1[2];
        double x[2][kQuadUnits];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          m0[i] = *reinterpret_cast<const double2 *>(tile_l + (d + i) * doutp + c * 4);
          m1[i] = *reinterpret_cast<const double2 *>(tile_l + (d + i) * doutp + c * 4 + 2);
#pragma unroll
          for (int u = 0; u < kQuadUnits; ++u) x[i][u] = cst[u * ustride + (d + i) * CS + lane];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int u = 0; u < kQuadUnits; ++u) {
            a[u][0] = fma(x[i][u], m0[i].x, a[u][0]);
            a[u][1] = fma(x[i][u], m0[i].y, a[u][1]);
            a[u][2] = fma(x[i][u], m1[i].x, a[u][2]);
            a[u][3] = fma(x[i][u], m1[i].y, a[u][3]);
          }
      }
      const int j0 = c * 4;
#pragma unroll
      for (int u = 0; u < kQuadUnits; ++u) {
        const int pr = u * kUnitPairs + lane;
        if (pr < np_all) {
          const size_t q = static_cast<size_t>(ch.q_begin + pr);
          double *dst = pair_out_ptr(pa, out, out_tail, q, j0);
          double2 w0, w1;
          w0.x = a[u][0]; w0.y = a[u][1]; w1.x = a[u][2]; w1.y = a[u][3];
          *reinterpret_cast<double2 *>(dst) = w0;
          *reinterpret_cast<double2 *>(dst + 2) = w1;
        }
      }
    }
    if (!has_next) break;
    ci += stride;
    ch = nx;
    nx = nn;
    has_next = has_next2;
  }
#undef PRJ
#undef DJ
}

// ======================================================================================
// pair_mfma -- the pair stage on the matrix cores, for rating tiles that no longer fit the scalar cache
// (K x L > 1024 with K, L <= 64: BASELINE's K = L = 50).  There the lane-per-pair form is bound by the
// LDS pipe (one broadcast ds_read_b128 of the tile per two FMAs; profiles/r2_c5: SQ_WAIT_INST_LDS),
// while both products of a 64-pair unit are small dense GEMMs:
//   T[64 x Dout]   = X[64 x Din] . tile[Din x Dout]     (X = C rows; the gathered eta rows in the A launch)
//   S[Din x Dout] += X^T[Din x 64] . E[64 x Dout]       (E = gathered eta rows; T+S launch only)
// v_mfma_f64_16x16x4_f64 takes ONE double per lane and operand (A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15]; D[row = (lane >> 4) + 4 reg][col = lane & 15]): 2,048 flops per KB
// read from LDS, 16 x less LDS traffic per flop than the lane-per-pair form.  (The f64 matrix rate
// equals the f64 vector rate on this chip: the gain is operand delivery, not a higher peak.)
// Staging as in pair_block (cst = X transposed with an odd stride, which serves both products without
// bank conflicts; es = eta rows; the tile once per workgroup).  Four waves:
//   T: wave w owns rows 16w .. 16w+15 of the unit and all (<= 4) column tiles; the results go to
//      memory from the accumulators (16 lanes = 128 contiguous bytes of a row);
//   S: the (<= 16) 16 x 16 tiles of the slab are dealt to the waves, <= 4 each, and stay in the
//      accumulators across the workgroup's units; one slab per workgroup at the end, as before.
// Rows / columns beyond Din / Dout inside a 16-tile are computed on clamped (duplicate) operands and
// never stored.  Association order per output: k (resp. pair) ascending, fused in groups of four.
// ======================================================================================
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
#ifndef MMSBM_MFMA_WPE
#define MMSBM_MFMA_WPE 4  // waves per SIMD the eight-wave form is compiled for (4: two workgroups per CU)
#endif
constexpr int kMfmaMaxDim = 64;       // <= 4 tiles of 16 per side
constexpr int kMfmaChunkPairs = 1024;  // pairs per workgroup at most (their item ids are parked in LDS)
static_assert(kMfmaChunkPairs >= 4 * mmsbm::kMvChunkPairs, "pair_mfma_kernel parks a whole chunk's item ids in LDS");

// NT threads: 256 (four waves as described) or 512 -- eight waves, each with half of the column tiles of its
// T rows and <= 2 slab tiles, so that the accumulators and the prefetched rows of the T+S launch fit
// 128 registers and two workgroups (16 waves) share a CU.
template <bool GATHER, bool DO_S, int NT>
__global__ __launch_bounds__(NT, NT == 512 ? MMSBM_MFMA_WPE : 2) void pair_mfma_kernel(PairBlockArgs pa,
                                                           const double *__restrict__ tiles0) {
  constexpr int NW = NT / 64;                              // waves
  constexpr int NLD = kUnitPairs * kMfmaMaxDim / 2 / NT;   // double2 per thread, unit and table
  constexpr int TC = 16 / NW;                              // column tiles per wave in T (4 or 2)
  constexpr int SA = 16 / NW;                              // slab tiles per wave at most (4 or 2)
  const size_t slot = blockIdx.y;
  const double *__restrict__ tiles = tiles0 + slot * pa.bs_tiles;
  const double *__restrict__ in_tab = pa.in_tab + slot * pa.bs_in;
  const double *__restrict__ e_tab = pa.e_tab + slot * pa.bs_e;
  const int32_t *__restrict__ pair_item = pa.pair_item;
  double *__restrict__ out = pa.out + slot * pa.bs_out;
  double *__restrict__ out_tail = pa.out_tail + slot * pa.bs_out_t;
  double *__restrict__ partial = pa.partial + slot * pa.bs_partial;
  const int dinp = pa.dinp, doutp = pa.doutp;
  extern __shared__ double lds[];
  constexpr int CS = kUnitPairs + 1;
  double *cst = lds;                                            // [dinp][CS]   X transposed
  double *tile_l = cst + static_cast<size_t>(dinp) * CS;        // [dinp][doutp]
  double *es = tile_l + static_cast<size_t>(dinp) * doutp;      // [64][doutp]  (DO_S)
  int *ids_l = reinterpret_cast<int *>(es + (DO_S ? static_cast<size_t>(kUnitPairs) * doutp : 0));  // [256]
  STAMP(0);
  const mmsbm::Chunk ch = pa.chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int nto = (doutp + 15) >> 4, mti = (dinp + 15) >> 4;
  {
    const double *src = tiles + static_cast<size_t>(ch.rating) * dinp * doutp;
    for (int t = tid * 2; t < dinp * doutp; t += NT * 2)
      *reinterpret_cast<double2 *>(tile_l + t) = *reinterpret_cast<const double2 *>(src + t);
    if (GATHER || DO_S)
      for (int t = tid; t < ch.q_end - ch.q_begin; t += NT) ids_l[t] = pair_item[ch.q_begin + t];
  }
  // this thread's share of a unit: elements t = 2 tid + j * 2 NT of the flat (pair, entry) space of X and of E.
  // (pair = t / dinp through a multiply-high with ceil(2^32 / dinp), exact for t < 2^32 / dinp: recomputed at
  // every use, a table of them per thread cost the registers that decide whether two workgroups share a CU)
  const unsigned mx = 0xFFFFFFFFu / static_cast<unsigned>(dinp) + 1u, me = 0xFFFFFFFFu / static_cast<unsigned>(doutp) + 1u;
#define MFMA_PX(j) static_cast<int>(__umulhi(static_cast<unsigned>(tid * 2 + (j) * NT * 2), mx))
#define MFMA_PE(j) static_cast<int>(__umulhi(static_cast<unsigned>(tid * 2 + (j) * NT * 2), me))
  const int trow0 = 16 * (wave & 3), tn0 = TC * (wave >> 2);  // T: this wave's rows and first column tile
  int bcol[TC];  // this lane's column of each of its output tiles
#pragma unroll
  for (int n = 0; n < TC; ++n) bcol[n] = min(16 * (tn0 + n) + li, doutp - 1);
  // S: tile t = wave + NW a of the mti x nto grid
  mfma_d4 acc_s[SA];
  int s_a[SA], s_b[SA];
  bool s_on[SA];
#pragma unroll
  for (int a = 0; a < SA; ++a) {
    const int t = wave + NW * a;
    s_on[a] = DO_S && t < mti * nto;
    const int m = s_on[a] ? t / nto : 0, n = s_on[a] ? t - m * nto : 0;
    s_a[a] = min(16 * m + li, dinp - 1) * CS + lk;         // + 4 s           : X[pair 4s + lk][k]
    s_b[a] = lk * doutp + min(16 * n + li, doutp - 1);     // + 4 s * doutp   : E[pair 4s + lk][l]
    acc_s[a] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();  // ids (and the tile) are in LDS

  // rows of the unit at q0 into registers: every load of the unit is in flight at once
  double2 vx[NLD];
  double vex[NLD], vey[NLD];  // (as scalars: a double2 array stored with ds_write_b128 stayed in scratch memory)
  // (a macro, not a lambda: arrays captured by reference ended up in scratch memory, and every scratch
  // access waits for ALL outstanding loads)
#define MFMA_FETCH(Q0)                                                                                  \
  do {                                                                                                  \
    const int fq = (Q0), fnp = min(kUnitPairs, ch.q_end - fq), fbase = fq - ch.q_begin;                  \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                   \
      /* unconditional: shares beyond the unit's pairs re-read its last row and are dropped */          \
      const int fp0 = MFMA_PX(j), fpr = min(fp0, fnp - 1);                                              \
      const size_t frow = GATHER ? static_cast<size_t>(ids_l[fbase + fpr]) : static_cast<size_t>(fq + fpr); \
      vx[j] = *reinterpret_cast<const double2 *>(in_tab + frow * dinp + (tid * 2 + j * NT * 2 - fp0 * dinp)); \
      if (DO_S) {                                                                                       \
        const int fe0 = MFMA_PE(j);                                                                     \
        const size_t ferow = static_cast<size_t>(ids_l[fbase + min(fe0, fnp - 1)]);                     \
        const double2 fe = *reinterpret_cast<const double2 *>(e_tab + ferow * doutp + (tid * 2 + j * NT * 2 - fe0 * doutp)); \
        vex[j] = fe.x;                                                                                  \
        vey[j] = fe.y;                                                                                  \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
  if (ch.q_begin < ch.q_end) MFMA_FETCH(ch.q_begin);  // (an empty chunk -- padding of the unit list -- only writes its zero slab)

  for (int q0 = ch.q_begin; q0 < ch.q_end; q0 += kUnitPairs) {
    const int np = min(kUnitPairs, ch.q_end - q0);
    STAMP(1);
    if (q0 != ch.q_begin) __syncthreads();  // previous unit fully consumed
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int pr = MFMA_PX(j);
      if (pr < np) {
        double *dst = cst + (tid * 2 + j * NT * 2 - pr * dinp) * CS + pr;
        dst[0] = vx[j].x;
        dst[CS] = vx[j].y;
      }
      if (DO_S && MFMA_PE(j) < np) {
        double2 e2;
        e2.x = vex[j]; e2.y = vey[j];
        *reinterpret_cast<double2 *>(es + (tid * 2 + j * NT * 2)) = e2;
      }
    }
    if (np < kUnitPairs) {  // ragged tail of a rating: the missing pairs are zero columns of X, zero rows of E
      for (int t = tid; t < (kUnitPairs - np) * dinp; t += NT)
        cst[(t / (kUnitPairs - np)) * CS + np + t % (kUnitPairs - np)] = 0.0;
      if (DO_S)
        for (int t = np * doutp + tid; t < kUnitPairs * doutp; t += NT) es[t] = 0.0;
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    if (q0 + kUnitPairs < ch.q_end) MFMA_FETCH(q0 + kUnitPairs);  // the next unit's rows travel during the products
    STAMP(4);
    if (DO_S) {  // S += X^T E : the 64 pairs are the summed index, four per instruction
#pragma unroll 4
      for (int s = 0; s < kUnitPairs / 4; ++s) {
#pragma unroll
        for (int a = 0; a < SA; ++a)  // (unconditional: a tile beyond the grid repeats tile 0 and is never stored)
          acc_s[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(cst[s_a[a] + 4 * s], es[s_b[a] + 4 * s * doutp],
                                                          acc_s[a], 0, 0, 0);
      }
    }
    STAMP(5);
    // T = X tile : this wave's 16 rows, its column tiles
    mfma_d4 acc_t[TC];
#pragma unroll
    for (int n = 0; n < TC; ++n) acc_t[n] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    if (tn0 < nto) {
#pragma unroll 2
      for (int s = 0; s < dinp / 4; ++s) {
        const double x = cst[(4 * s + lk) * CS + trow0 + li];
        const double *trow = tile_l + (4 * s + lk) * doutp;
#pragma unroll
        for (int n = 0; n < TC; ++n)  // (a column tile beyond Dout repeats the last column and is never stored)
          acc_t[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, trow[bcol[n]], acc_t[n], 0, 0, 0);
      }
    }
    STAMP(6);
#pragma unroll
    for (int n = 0; n < TC; ++n) {
      const int col = 16 * (tn0 + n) + li;
      if (tn0 + n < nto && col < doutp) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r = trow0 + lk + 4 * g;
          if (r < np) *pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + r), col) = acc_t[n][g];
        }
      }
    }
  }
  STAMP(7);
  if (DO_S) {
    double *dst = partial + static_cast<size_t>(blockIdx.x) * dinp * doutp;
#pragma unroll
    for (int a = 0; a < SA; ++a) {
      if (!s_on[a]) continue;
      const int t = wave + NW * a, m = t / nto, n = t - m * nto;
      const int col = 16 * n + li;
      if (col < doutp) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k = 16 * m + lk + 4 * g;
          if (k < dinp) dst[static_cast<size_t>(k) * doutp + col] = acc_s[a][g];
        }
      }
    }
  }
#ifdef MMSBM_STAMPS
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(8);
#endif
}
#undef MFMA_FETCH
#undef MFMA_PX
#undef MFMA_PE
// ======================================================================================
// The same two products for K or L beyond 64, in 64 x 64 blocks (round 2).  Before, these shapes ran the
// lane-per-pair stage with the tile through scalar loads (K, L up to ~150) or the plain wide-row kernels
// (beyond): at K = L = 100 the pair stage took 61 % of the iteration, at K = L = 200 75 %.  Blocked, T and
// S no longer share a workgroup (T sums over ALL of Din for a block of outputs; S keeps a Din x Dout block
// in the accumulators over ALL pairs of a chunk), so the T+S stage is two launches:
//   mfma_rows_kernel<GATHER>  workgroup = (64-pair unit, block of <= 64 output columns): loops over the
//       64-blocks of Din -- X block transposed + tile block in LDS, the next blocks in flight -- with the
//       output tiles in the accumulators throughout (wave = 16 rows x 2 column tiles);
//   mfma_slab_kernel          workgroup = (chunk, Din block, Dout block): loops over the chunk's units --
//       X block transposed + E block in LDS -- with its <= 16 slab tiles dealt to the 8 waves.
// Operand layouts, clamping of partial tiles and association order as in pair_mfma_kernel.  Each table is
// re-read once per block of the other side (from L2 / the Infinity Cache: blocks of one unit are
// neighbours in the grid).
// ======================================================================================
constexpr int kMfmaBlk = 64;

// pair = t / w for t < 2^32 / w through a multiply-high (see pair_mfma_kernel)
__device__ __forceinline__ unsigned mfma_magic(int w) { return 0xFFFFFFFFu / static_cast<unsigned>(w) + 1u; }

template <bool GATHER>
__global__ __launch_bounds__(kPairBlockMax, 4) void mfma_rows_kernel(PairBlockArgs pa, const double *__restrict__ tiles0,
                                                                    int subs_per_chunk, int n_lb) {
  constexpr int NT = kPairBlockMax, CS = kUnitPairs + 1, NLD = kUnitPairs * kMfmaBlk / 2 / NT;  // 4 double2 per table
  const size_t slot = blockIdx.y;
  const double *__restrict__ tiles = tiles0 + slot * pa.bs_tiles;
  const double *__restrict__ in_tab = pa.in_tab + slot * pa.bs_in;
  double *__restrict__ out = pa.out + slot * pa.bs_out;
  double *__restrict__ out_tail = pa.out_tail + slot * pa.bs_out_t;
  const int dinp = pa.dinp, doutp = pa.doutp;
  const int lb = static_cast<int>(blockIdx.x % n_lb), usub = static_cast<int>(blockIdx.x / n_lb);
  const mmsbm::Chunk ch = pa.chunks[usub / subs_per_chunk];
  const int q0 = ch.q_begin + (usub % subs_per_chunk) * kUnitPairs;
  if (q0 >= ch.q_end) return;
  const int np = min(kUnitPairs, ch.q_end - q0);
  const int lb0 = lb * kMfmaBlk, lbw = min(kMfmaBlk, doutp - lb0);
  extern __shared__ double lds[];
  double *cst = lds;                        // [64 k'][CS]   X block, transposed
  double *tile_b = cst + kMfmaBlk * CS;     // [64 k'][64]   tile block
  int *ids_l = reinterpret_cast<int *>(tile_b + kMfmaBlk * kMfmaBlk);  // [64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  if (GATHER) {
    if (tid < kUnitPairs) ids_l[tid] = pa.pair_item[q0 + min(tid, np - 1)];
    __syncthreads();
  }
  const double *__restrict__ tile_r = tiles + static_cast<size_t>(ch.rating) * dinp * doutp + lb0;
  const int trow0 = 16 * (wave & 3), tn0 = 2 * (wave >> 2);
  int bcol[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) bcol[n] = min(16 * (tn0 + n) + li, lbw - 1);
  mfma_d4 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) acc[n] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  const unsigned ml = mfma_magic(lbw);
  double2 vx[NLD];
  double tx[NLD], ty[NLD];
  // shares of a block: X element t = (pair, k') with pair = t / kbw; tile element t = (k', j) with k' = t / lbw
#define ROWS_FETCH(KB0)                                                                                        \
  do {                                                                                                         \
    const int fk0 = (KB0), fkw = min(kMfmaBlk, dinp - fk0);                                                    \
    const unsigned fmx = mfma_magic(fkw);                                                                      \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                          \
      const int t = tid * 2 + j * NT * 2;                                                                      \
      const int fp0 = static_cast<int>(__umulhi(static_cast<unsigned>(t), fmx)), fpr = min(fp0, np - 1);       \
      const int fd = min(t - fp0 * fkw, fkw - 2);                                                              \
      const size_t frow = GATHER ? static_cast<size_t>(ids_l[fpr]) : static_cast<size_t>(q0 + fpr);            \
      vx[j] = *reinterpret_cast<const double2 *>(in_tab + frow * dinp + fk0 + fd);                             \
      const int fr0 = static_cast<int>(__umulhi(static_cast<unsigned>(t), ml)), frr = min(fr0, fkw - 1);       \
      const double2 ft = *reinterpret_cast<const double2 *>(tile_r + static_cast<size_t>(fk0 + frr) * doutp +  \
                                                            min(t - fr0 * lbw, lbw - 2));                      \
      tx[j] = ft.x;                                                                                            \
      ty[j] = ft.y;                                                                                            \
    }                                                                                                          \
  } while (0)
  ROWS_FETCH(0);
  for (int kb0 = 0; kb0 < dinp; kb0 += kMfmaBlk) {
    const int kbw = min(kMfmaBlk, dinp - kb0);
    const unsigned mx = mfma_magic(kbw);
    if (kb0 != 0) __syncthreads();  // previous block fully consumed
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int t = tid * 2 + j * NT * 2;
      const int pr = static_cast<int>(__umulhi(static_cast<unsigned>(t), mx));
      if (pr < kUnitPairs) {  // (pairs >= np hold a copy of the last row: their output rows are never stored)
        double *dst = cst + (t - pr * kbw) * CS + pr;
        dst[0] = vx[j].x;
        dst[CS] = vx[j].y;
      }
      const int kr = static_cast<int>(__umulhi(static_cast<unsigned>(t), ml));
      if (kr < kbw) {
        double2 t2;
        t2.x = tx[j]; t2.y = ty[j];
        *reinterpret_cast<double2 *>(tile_b + kr * kMfmaBlk + (t - kr * lbw)) = t2;
      }
    }
    __syncthreads();
    if (kb0 + kMfmaBlk < dinp) ROWS_FETCH(kb0 + kMfmaBlk);
#pragma unroll 2
    for (int s = 0; s < kbw / 4; ++s) {
      const double x = cst[(4 * s + lk) * CS + trow0 + li];
      const double *trow = tile_b + (4 * s + lk) * kMfmaBlk;
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, trow[bcol[n]], acc[n], 0, 0, 0);
    }
  }
#undef ROWS_FETCH
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = 16 * (tn0 + n) + li;
    if (col < lbw) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = trow0 + lk + 4 * g;
        if (r < np) *pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + r), lb0 + col) = acc[n][g];
      }
    }
  }
}

__global__ __launch_bounds__(kPairBlockMax, 4) void mfma_slab_kernel(PairBlockArgs pa, int n_kb, int n_lb) {
  constexpr int NT = kPairBlockMax, NW = NT / 64, CS = kUnitPairs + 1, NLD = kUnitPairs * kMfmaBlk / 2 / NT;
  const size_t slot = blockIdx.y;
  const double *__restrict__ in_tab = pa.in_tab + slot * pa.bs_in;
  const double *__restrict__ e_tab = pa.e_tab + slot * pa.bs_e;
  double *__restrict__ partial = pa.partial + slot * pa.bs_partial;
  const int dinp = pa.dinp, doutp = pa.doutp;
  const int blk = static_cast<int>(blockIdx.x % (n_kb * n_lb)), chunk = static_cast<int>(blockIdx.x / (n_kb * n_lb));
  const int kb0 = (blk / n_lb) * kMfmaBlk, lb0 = (blk % n_lb) * kMfmaBlk;
  const int kbw = min(kMfmaBlk, dinp - kb0), lbw = min(kMfmaBlk, doutp - lb0);
  const mmsbm::Chunk ch = pa.chunks[chunk];
  extern __shared__ double lds[];
  double *cst = lds;                      // [64 k'][CS]  X block, transposed
  double *es = cst + kMfmaBlk * CS;       // [64 pairs][64]  E block
  int *ids_l = reinterpret_cast<int *>(es + kUnitPairs * kMfmaBlk);  // [kMfmaChunkPairs]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  for (int t = tid; t < ch.q_end - ch.q_begin; t += NT) ids_l[t] = pa.pair_item[ch.q_begin + t];
  const int mt = (kbw + 15) >> 4, nt = (lbw + 15) >> 4;
  mfma_d4 acc[2];
  int s_a[2], s_b[2];
  bool s_on[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int t = wave + NW * a;
    s_on[a] = t < mt * nt;
    const int m = s_on[a] ? t / nt : 0, n = s_on[a] ? t - m * nt : 0;
    s_a[a] = min(16 * m + li, kbw - 1) * CS + lk;
    s_b[a] = lk * kMfmaBlk + min(16 * n + li, lbw - 1);
    acc[a] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  }
  const unsigned mx = mfma_magic(kbw), me = mfma_magic(lbw);
  __syncthreads();  // ids
  double2 vx[NLD];
  double ex[NLD], ey[NLD];
#define SLAB_FETCH(Q0)                                                                                         \
  do {                                                                                                         \
    const int fq = (Q0), fnp = min(kUnitPairs, ch.q_end - fq), fbase = fq - ch.q_begin;                         \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                          \
      const int t = tid * 2 + j * NT * 2;                                                                      \
      const int fp0 = static_cast<int>(__umulhi(static_cast<unsigned>(t), mx)), fpr = min(fp0, fnp - 1);       \
      vx[j] = *reinterpret_cast<const double2 *>(in_tab + static_cast<size_t>(fq + fpr) * dinp + kb0 +         \
                                                 min(t - fp0 * kbw, kbw - 2));                                 \
      const int fe0 = static_cast<int>(__umulhi(static_cast<unsigned>(t), me));                                \
      const size_t ferow = static_cast<size_t>(ids_l[fbase + min(fe0, fnp - 1)]);                              \
      const double2 fe = *reinterpret_cast<const double2 *>(e_tab + ferow * doutp + lb0 + min(t - fe0 * lbw, lbw - 2)); \
      ex[j] = fe.x;                                                                                            \
      ey[j] = fe.y;                                                                                            \
    }                                                                                                          \
  } while (0)
  if (ch.q_begin < ch.q_end) SLAB_FETCH(ch.q_begin);  // (an empty chunk only writes its zero block)
  for (int q0 = ch.q_begin; q0 < ch.q_end; q0 += kUnitPairs) {
    const int np = min(kUnitPairs, ch.q_end - q0);
    if (q0 != ch.q_begin) __syncthreads();
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int t = tid * 2 + j * NT * 2;
      const int pr = static_cast<int>(__umulhi(static_cast<unsigned>(t), mx));
      if (pr < kUnitPairs) {  // pairs >= np: zero columns of X (so that whatever E holds there adds nothing)
        double *dst = cst + (t - pr * kbw) * CS + pr;
        dst[0] = pr < np ? vx[j].x : 0.0;
        dst[CS] = pr < np ? vx[j].y : 0.0;
      }
      const int pe = static_cast<int>(__umulhi(static_cast<unsigned>(t), me));
      if (pe < kUnitPairs) {  // (pairs >= np: a copy of the last row, finite)
        double2 e2;
        e2.x = ex[j]; e2.y = ey[j];
        *reinterpret_cast<double2 *>(es + pe * kMfmaBlk + (t - pe * lbw)) = e2;
      }
    }
    __syncthreads();
    if (q0 + kUnitPairs < ch.q_end) SLAB_FETCH(q0 + kUnitPairs);
#pragma unroll 4
    for (int s = 0; s < kUnitPairs / 4; ++s) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
        acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(cst[s_a[a] + 4 * s], es[s_b[a] + 4 * s * kMfmaBlk], acc[a], 0, 0, 0);
    }
  }
#undef SLAB_FETCH
  double *dst = partial + static_cast<size_t>(chunk) * dinp * doutp;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    if (!s_on[a]) continue;
    const int t = wave + NW * a, m = t / nt, n = t - m * nt;
    const int col = 16 * n + li;
    if (col < lbw) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = 16 * m + lk + 4 * g;
        if (k < kbw) dst[static_cast<size_t>(kb0 + k) * doutp + lb0 + col] = acc[a][g];
      }
    }
  }
}
constexpr size_t kMfmaRowsLds = (kMfmaBlk * (kUnitPairs + 1) + kMfmaBlk * kMfmaBlk) * sizeof(double) + kUnitPairs * sizeof(int);
constexpr size_t kMfmaSlabLds = (kMfmaBlk * (kUnitPairs + 1) + kUnitPairs * kMfmaBlk) * sizeof(double) + kMfmaChunkPairs * sizeof(int);

size_t pair_mfma_lds(int dinp, int doutp, bool with_s) {
  return (static_cast<size_t>(dinp) * (kUnitPairs + 1) + static_cast<size_t>(dinp) * doutp +
          (with_s ? static_cast<size_t>(kUnitPairs) * doutp : 0)) * sizeof(double) + kMfmaChunkPairs * sizeof(int);
}

// ======================================================================================
// wide rows (K, L beyond what the 64-pair LDS stage holds: roughly K + L > 300): the pair stage in its
// plain form, so that every (K, L) the reference accepts runs (src/kernels_numpy.py:21-79 has no size
// limit).  Same tables, same chunk list (chunks of up to kWideChunkPairs pairs of one rating), same
// eta_p launch behind it; only the two mat-vecs and the slab sums are done differently:
//   wide_matvec : a workgroup takes 8 pairs, parks their input rows in LDS and walks the outputs
//                 j = tid, tid + 256, ...: out[q, j] = sum_d in[q, d] tile[d, j] (tile rows read
//                 coalesced from global memory / L2, each value feeding 8 pairs);
//   wide_slab   : thread = one l for 8 consecutive k of one chunk: S[k, l] = sum_q C[q, k] eta[i_q, l].
// Per output the association order is the LDS stage's (d ascending, one accumulator).
// ======================================================================================
constexpr int kWidePairs = 8, kWideChunkPairs = 1024;

template <bool GATHER>
__global__ __launch_bounds__(kBlock) void wide_matvec_kernel(PairBlockArgs pa, int subs_per_chunk) {
  extern __shared__ double lds[];  // [kWidePairs][dinp]
  const size_t slot = blockIdx.y;
  const mmsbm::Chunk ch = pa.chunks[blockIdx.x / subs_per_chunk];
  const int q0 = ch.q_begin + static_cast<int>(blockIdx.x % subs_per_chunk) * kWidePairs;
  if (q0 >= ch.q_end) return;
  const int np = min(kWidePairs, ch.q_end - q0);
  const int dinp = pa.dinp, doutp = pa.doutp, tid = threadIdx.x;
  const double *__restrict__ in_tab = pa.in_tab + slot * pa.bs_in;
  const double *__restrict__ tile = pa.tiles + slot * pa.bs_tiles + static_cast<size_t>(ch.rating) * dinp * doutp;
  double *__restrict__ out = pa.out + slot * pa.bs_out;
  double *__restrict__ out_tail = pa.out_tail + slot * pa.bs_out_t;
  for (int t = tid; t < kWidePairs * dinp; t += kBlock) {
    const int pr = t / dinp, d = t - pr * dinp;
    double v = 0.0;
    if (pr < np) {
      const size_t row = GATHER ? static_cast<size_t>(pa.pair_item[q0 + pr]) : static_cast<size_t>(q0 + pr);
      v = in_tab[row * dinp + d];
    }
    lds[t] = v;
  }
  __syncthreads();
  for (int j = tid; j < doutp; j += kBlock) {
    double acc[kWidePairs];
#pragma unroll
    for (int pr = 0; pr < kWidePairs; ++pr) acc[pr] = 0.0;
    for (int d = 0; d < dinp; ++d) {
      const double m = tile[static_cast<size_t>(d) * doutp + j];
#pragma unroll
      for (int pr = 0; pr < kWidePairs; ++pr) acc[pr] = fma(lds[pr * dinp + d], m, acc[pr]);
    }
#pragma unroll
    for (int pr = 0; pr < kWidePairs; ++pr) {
      if (pr < np) {
        const size_t q = static_cast<size_t>(q0 + pr);
        *pair_out_ptr(pa, out, out_tail, q, j) = acc[pr];
      }
    }
  }
}

// thread = one l of a block of 256, for kWideKG consecutive k: every eta value read feeds kWideKG sums
constexpr int kWideKG = 8;  // (16: the C values no longer fit the scalar registers, 921 vs 477 us)
__global__ __launch_bounds__(kBlock) void wide_slab_kernel(PairBlockArgs pa, int k_groups, int l_blocks) {
  const size_t slot = blockIdx.y;
  const int per_chunk = k_groups * l_blocks;
  const int chunk = blockIdx.x / per_chunk, rem = blockIdx.x - chunk * per_chunk;
  const int k0 = (rem / l_blocks) * kWideKG, l = (rem % l_blocks) * kBlock + static_cast<int>(threadIdx.x);
  const int kp = pa.dinp, lp = pa.doutp, kl = kp * lp;
  if (l >= lp) return;
  const mmsbm::Chunk ch = pa.chunks[chunk];
  const double *__restrict__ ctab = pa.in_tab + slot * pa.bs_in;
  const double *__restrict__ eta = pa.e_tab + slot * pa.bs_e;
  double acc[kWideKG];
#pragma unroll
  for (int j = 0; j < kWideKG; ++j) acc[j] = 0.0;
  constexpr int UB = 4;  // pairs per round: their ids, eta values and C values are in flight together
  for (int q0 = ch.q_begin; q0 < ch.q_end; q0 += UB) {
    int id[UB];
    double ev[UB], cv[UB][kWideKG];
#pragma unroll
    for (int b = 0; b < UB; ++b) id[b] = pa.pair_item[min(q0 + b, ch.q_end - 1)];
#pragma unroll
    for (int b = 0; b < UB; ++b) {
      ev[b] = eta[static_cast<size_t>(id[b]) * lp + l];
      const double *crow = ctab + static_cast<size_t>(min(q0 + b, ch.q_end - 1)) * kp + k0;  // (kp is a multiple of 4)
#pragma unroll
      for (int j = 0; j < kWideKG; ++j) cv[b][j] = (k0 + j < kp) ? crow[j] : 0.0;
    }
#pragma unroll
    for (int b = 0; b < UB; ++b) {
      if (q0 + b < ch.q_end) {  // (per (k, l): pairs in ascending order, one accumulator)
#pragma unroll
        for (int j = 0; j < kWideKG; ++j) acc[j] = fma(cv[b][j], ev[b], acc[j]);
      }
    }
  }
  double *dst = pa.partial + slot * pa.bs_partial + static_cast<size_t>(chunk) * kl;
#pragma unroll
  for (int j = 0; j < kWideKG; ++j)
    if (k0 + j < kp) dst[static_cast<size_t>(k0 + j) * lp + l] = acc[j];
}

constexpr size_t kLdsBudget = 64 * 1024;  // dynamic LDS a launch may use without hipFuncSetAttribute

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

SegArgs seg_pairs_args(const mmsbm_hip_ctx *c) {  // C = sum over a pair's triples
  const bool it = !c->lay.pair_work.items.empty();
  return SegArgs{a_tab(c, c->cur), theta_tab(c, c->cur), c->pair_off.ptr, c->pair_user.ptr,
                 plain_tab(c->ctab.at(c->base_slot), c->kp, c->ctab.stride),
                 it ? static_cast<int32_t>(c->lay.pair_work.items.size()) : c->n_pairs, 0,
                 it ? c->pair_items.ptr : nullptr, c->pair_parts.at(c->base_slot), c->pair_parts.stride};
}
SegArgs seg_users_args(const mmsbm_hip_ctx *c, bool commit, int seg_end) {  // theta_new
  const bool it = !c->lay.user_work.items.empty();
  return SegArgs{theta_tab(c, c->cur),     a_tab(c, c->cur), c->user_off.ptr, c->user_pair.ptr,
                 theta_tab(c, c->cur ^ 1),
                 it ? static_cast<int32_t>(c->lay.user_work.items.size()) : seg_end,
                 commit ? 1 : 2,
                 it ? c->user_items.ptr : nullptr, c->user_parts.at(c->base_slot), c->user_parts.stride};
}
PairBlockArgs pair_block_t_args(const mmsbm_hip_ctx *c) {
  const int s = c->base_slot;
  PairBlockArgs pa{};
  pa.tiles = c->p[c->cur].at(s); pa.in_tab = c->ctab.at(s); pa.e_tab = c->eta[c->cur].at(s);
  pa.pair_item = c->pair_item.ptr; pa.chunks = c->mv_chunks.ptr;
  pa.out = c->ttab.at(s); pa.partial = c->partial.at(s);
  pa.din = c->k; pa.dinp = c->kp; pa.doutp = c->lp; pa.spb = c->pb_spb; pa.nsub = c->pb_nsub; pa.abl = c->ablate;
  pa.out_mw = c->lp; pa.out_rs_m = c->lp; pa.out_rs_t = 0; pa.out_tail = c->ttab.at(s);
  pa.bs_tiles = c->p[0].stride; pa.bs_in = c->ctab.stride; pa.bs_e = c->eta[0].stride;
  pa.bs_out = c->ttab.stride; pa.bs_out_t = 0; pa.bs_partial = c->partial.stride;
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
  pa.out_mw = at.mw; pa.out_rs_m = at.rs_m; pa.out_rs_t = at.rs_t; pa.out_tail = at.tail;
  pa.bs_tiles = c->pt[0].stride; pa.bs_in = c->eta[0].stride; pa.bs_e = 0;
  pa.bs_out = at.so_m; pa.bs_out_t = at.so_t; pa.bs_partial = 0;
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
  LaunchScope ls(c, K_SEG);
  const SegArgs sp = seg_pairs_args(c), su = seg_users_args(c, commit, c->n_users);
  const int per = kBlock / group_lanes(c->code_k);
  // several restart slots: a super-group of SW x G lanes per segment (seg_pass_slots_kernel)
  int sw = 1;
  if (c->launch_slots > 1 && c->slot_waves) {
    const int room = 64 / group_lanes(c->code_k);
    while (sw * 2 <= room && sw < c->launch_slots) sw *= 2;
  }
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
#define CALL(G, V) \
  seg_pass_kernel<G, V, 4><<<slot_grid(c, bp + bu), kBlock, 0, st>>>(sp, su, bp, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  }
  // long segments were processed in pieces: add the pieces up (fixed order) and finish them
  // (splits with few pieces come first in the lists: one group of lanes each; the rest: a workgroup each)
  const mmsbm::WorkList &wp = c->lay.pair_work, &wu = c->lay.user_work;
  const int nsp_s = with_pairs ? wp.n_small : 0;
  const int nsp_b = with_pairs ? static_cast<int>(wp.splits.size()) - wp.n_small : 0;
  const int nsu_s = with_users ? wu.n_small : 0;
  const int nsu_b = with_users ? static_cast<int>(wu.splits.size()) - wu.n_small : 0;
  if (nsp_s + nsu_s > 0) {
    const CombineArgs cp{c->pair_splits.ptr, sp.parts, c->pair_off.ptr, sp.fixed, sp.out, nsp_s,
                         sp.mode, sp.bs_parts};
    const CombineArgs cu{c->user_splits.ptr, su.parts, c->user_off.ptr, su.fixed, su.out, nsu_s,
                         su.mode, su.bs_parts};
    const int ba = (nsp_s + per - 1) / per, bb = (nsu_s + per - 1) / per;
#define CALL(G, V) \
  seg_combine_small_kernel<G, V><<<slot_grid(c, ba + bb), kBlock, 0, st>>>(cp, cu, ba, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  }
  if (nsp_b + nsu_b > 0) {
    const CombineArgs cp{c->pair_splits.ptr + wp.n_small, sp.parts, c->pair_off.ptr, sp.fixed, sp.out, nsp_b,
                         sp.mode, sp.bs_parts};
    const CombineArgs cu{c->user_splits.ptr + wu.n_small, su.parts, c->user_off.ptr, su.fixed, su.out, nsu_b,
                         su.mode, su.bs_parts};
    const size_t lds = static_cast<size_t>(per) * c->kp * sizeof(double);
#define CALL(G, V) \
  seg_combine_kernel<G, V><<<slot_grid(c, nsp_b + nsu_b), kBlock, lds, st>>>(cp, cu, nsp_b, c->kp)
    DISPATCH_GV(c->code_k, CALL);
#undef CALL
  }
  ls.done();
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
    const int subs = c->mv_chunk_pairs / kUnitPairs;
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
    LaunchScope ls(c, K_DENSE);
    const int nb = static_cast<int>(c->lay.mv_chunks.size());
    const PairBlockArgs pa = pair_block_t_args(c);
    if (c->mfma_threads == kBlock) {
      allow_big_lds(pair_mfma_kernel<false, true, kBlock>, c->lds_mt);
      pair_mfma_kernel<false, true, kBlock><<<slot_grid(c, nb), kBlock, c->lds_mt, c->stream>>>(pa, pa.tiles);
    } else {
      allow_big_lds(pair_mfma_kernel<false, true, kPairBlockMax>, c->lds_mt);
      pair_mfma_kernel<false, true, kPairBlockMax><<<slot_grid(c, nb), kPairBlockMax, c->lds_mt, c->stream>>>(pa, pa.tiles);
    }
    ls.done();
    return;
  }
  {
    LaunchScope ls(c, K_DENSE);
    const int nb = static_cast<int>(c->lay.mv_chunks.size());
    const PairBlockArgs pa = pair_block_t_args(c);
#define PB_D(N, TL, NT, KT, D)                                                              \
  do {                                                                                      \
    allow_big_lds(pair_block_kernel<false, true, N, TL, NT, KT, D>, c->lds_t);              \
    pair_block_kernel<false, true, N, TL, NT, KT, D><<<slot_grid(c, nb), NT, c->lds_t, c->stream>>>(pa, pa.tiles); \
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
  LaunchScope ls(c, K_ETAP);
  const EtaPArgs a = eta_p_args(c, commit, kRedCols);
  const int per = kRedThreads / group_lanes(c->code_l);
  const int nb_i = (c->n_items + per - 1) / per;
#define CALL(G, V) eta_p_kernel<G, V><<<slot_grid(c, a.nb_p + nb_i), kRedThreads, 0, c->stream>>>(a)
  DISPATCH_GV(c->code_l, CALL);
#undef CALL
  ls.done();
}

// A[q,:] from (eta, pT) of parameter slot `slot` into atab[a_slot] -- or, with `grid` set, the same
// mat-vec over every (item, rating) combination into the plain table btab (prod_dist / predict)
void stage_matvec_a(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid = false) {
  const int nb = grid ? c->grid_n_chunks : static_cast<int>(c->lay.mv_chunks.size());
  if (nb == 0) return;
  LaunchScope ls(c, K_MATVEC_A);
  PairBlockArgs pa = pair_block_a_args(c, slot, a_slot);
  if (grid) {
    pa.pair_item = c->grid_item.ptr; pa.chunks = c->grid_chunks.ptr;
    pa.out = c->btab.ptr; pa.out_tail = nullptr;
    pa.out_mw = pa.doutp; pa.out_rs_m = pa.doutp; pa.out_rs_t = 0; pa.bs_out = 0; pa.bs_out_t = 0;
  }
  if (c->mfma_big) {
    const int subs = c->mv_chunk_pairs / kUnitPairs, n_kb = (c->kp + kMfmaBlk - 1) / kMfmaBlk;  // (outputs: K columns)
    allow_big_lds(mfma_rows_kernel<true>, kMfmaRowsLds);
    mfma_rows_kernel<true><<<slot_grid(c, nb * subs * n_kb), kPairBlockMax, kMfmaRowsLds, c->stream>>>(pa, pa.tiles, subs, n_kb);
  } else if (c->wide) {
    const int subs = kWideChunkPairs / kWidePairs;
    const size_t lds = static_cast<size_t>(kWidePairs) * c->lp * sizeof(double);
    allow_big_lds(wide_matvec_kernel<true>, lds);
    wide_matvec_kernel<true><<<slot_grid(c, nb * subs), kBlock, lds, c->stream>>>(pa, subs);
  } else if (c->mfma) {
    allow_big_lds(pair_mfma_kernel<true, false, kBlock>, c->lds_ma);
    pair_mfma_kernel<true, false, kBlock><<<slot_grid(c, nb), kBlock, c->lds_ma, c->stream>>>(pa, pa.tiles);
  } else if (c->quad_a) {
    const dim3 grid = slot_grid(c, std::min(nb, c->n_cus));
#define QA(NL)                                                                                    \
  do {                                                                                            \
    allow_big_lds(pair_quad_a_kernel<NL>, c->lds_qa);                                             \
    pair_quad_a_kernel<NL><<<grid, kPairBlockMax, c->lds_qa, c->stream>>>(pa, pa.tiles, nb);      \
  } while (0)
    const int nl = (c->lp + 3) / 4;  // dinp of the A launch = lp
    if (nl <= 8) QA(8); else if (nl <= 10) QA(10); else if (nl <= 12) QA(12);
    else if (nl <= 13) QA(13); else if (nl <= 14) QA(14); else QA(16);
#undef QA
  } else {
#define PA_D(TL, NT, D)                                                                     \
  do {                                                                                      \
    allow_big_lds(pair_block_kernel<true, false, 1, TL, NT, 4, D>, c->lds_a);               \
    pair_block_kernel<true, false, 1, TL, NT, 4, D><<<slot_grid(c, nb), NT, c->lds_a, c->stream>>>(pa, pa.tiles); \
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

void launch_iteration(mmsbm_hip_ctx *c, bool commit) {
  stage_seg(c, commit, true, true, c->stream);
  stage_dense(c);
  stage_eta_p(c, commit);
  if (commit) {
    stage_matvec_a(c, c->cur ^ 1, c->cur ^ 1);
    c->cur ^= 1;
  }
}

// n committed iterations: graph replays of two iterations each when enabled, the rest eager
void run_iterations(mmsbm_hip_ctx *c, int n) {
  if (c->graph_mode && !c->profiling) {
    while (n >= 2) {
      const int slot = c->cur;
      if (!c->graph_exec[slot]) {
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
  c->ctab.alloc_slots(static_cast<size_t>(c->n_pairs) * c->kp, slots);
  c->ttab.alloc_slots(static_cast<size_t>(c->n_pairs) * c->lp, slots);
  c->partial.alloc_slots(c->lay.mv_chunks.size() * c->kp * c->lp, slots);
  c->npr.alloc_slots(klr, slots);
  c->pair_parts.alloc_slots(static_cast<size_t>(c->lay.pair_work.n_parts) * c->kp, slots);
  c->user_parts.alloc_slots(static_cast<size_t>(c->lay.user_work.n_parts) * c->kp, slots);
  HIP_CHECK(hipStreamSynchronize(s));
  c->n_slots = slots;
  c->sel = 0;
  c->base_slot = 0;
  c->launch_slots = slots;
  c->cur = 0;
  c->have.assign(static_cast<size_t>(slots), 0);
}

// host (rows, d) row-major  <->  device RowTab (rows, dp) zero-padded, main + tail parts, staged
// through pinned memory in the one-slot layout (main rows, then tail rows): one contiguous copy when
// the context has one slot, a strided (2-D) copy per part when the slots' rows are interleaved
bool tab_is_packed(const RowTab &t, int rows) {
  return t.rs_m == t.mw && t.rs_t == t.tw && (t.tw == 0 || t.tail == t.main + static_cast<size_t>(rows) * t.mw);
}
void copy_rows(mmsbm_hip_ctx *c, const RowTab &t, double *stage, int rows, bool to_device) {
  const size_t e = sizeof(double);
  if (rows == 0) return;
  if (tab_is_packed(t, rows)) {
    if (to_device)
      HIP_CHECK(hipMemcpyAsync(t.main, stage, e * rows * (t.mw + t.tw), hipMemcpyHostToDevice, c->stream));
    else
      HIP_CHECK(hipMemcpyAsync(stage, t.main, e * rows * (t.mw + t.tw), hipMemcpyDeviceToHost, c->stream));
    return;
  }
  double *stage_t = stage + static_cast<size_t>(rows) * t.mw;
  if (to_device) {
    HIP_CHECK(hipMemcpy2DAsync(t.main, e * t.rs_m, stage, e * t.mw, e * t.mw, rows, hipMemcpyHostToDevice, c->stream));
    if (t.tw > 0)
      HIP_CHECK(hipMemcpy2DAsync(t.tail, e * t.rs_t, stage_t, e * t.tw, e * t.tw, rows, hipMemcpyHostToDevice, c->stream));
  } else {
    HIP_CHECK(hipMemcpy2DAsync(stage, e * t.mw, t.main, e * t.rs_m, e * t.mw, rows, hipMemcpyDeviceToHost, c->stream));
    if (t.tw > 0)
      HIP_CHECK(hipMemcpy2DAsync(stage_t, e * t.tw, t.tail, e * t.rs_t, e * t.tw, rows, hipMemcpyDeviceToHost, c->stream));
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
  HIP_CHECK(hipStreamSynchronize(c->stream));
  c->pin.reset(rows_doubles(c));
  const size_t klr = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  const double *st = theta ? download_rows(c, tt, c->n_users) : nullptr;
  const double *se = eta ? download_rows(c, et, c->n_items) : nullptr;
  double *sp = nullptr;
  if (pr) {
    sp = c->pin.take(klr);
    HIP_CHECK(hipMemcpyAsync(sp, p_dev, sizeof(double) * klr, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_CHECK(hipStreamSynchronize(c->stream));
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

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" {

int mmsbm_hip_abi_version(void) { return MMSBM_HIP_ABI_VERSION; }

const char *mmsbm_hip_last_error(void) { return g_last_error.c_str(); }

int mmsbm_hip_device_count(int *count) {
  return guarded([&] {
    if (!count) throw std::invalid_argument("null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
      *count = 0;
      throw ApiError(MMSBM_E_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
  });
}

int mmsbm_hip_device_info(int device, char *name, int name_len, int *compute_units,
                          int64_t *global_mem_bytes) {
  return guarded([&] {
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
      std::snprintf(name, static_cast<size_t>(name_len), "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (global_mem_bytes) *global_mem_bytes = static_cast<int64_t>(prop.totalGlobalMem);
  });
}

int mmsbm_hip_device_mem(int device, int64_t *free_bytes, int64_t *total_bytes) {
  return guarded([&] {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw ApiError(MMSBM_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    int prev = 0;
    HIP_CHECK(hipGetDevice(&prev));
    HIP_CHECK(hipSetDevice(device));
    size_t f = 0, t = 0;
    const hipError_t e = hipMemGetInfo(&f, &t);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) throw ApiError(MMSBM_E_HIP, std::string("hipMemGetInfo: ") + hipGetErrorString(e));
    if (free_bytes) *free_bytes = static_cast<int64_t>(f);
    if (total_bytes) *total_bytes = static_cast<int64_t>(t);
  });
}

int mmsbm_hip_create(int device, int64_t n_obs, int32_t n_users, int32_t n_items,
                     int32_t n_ratings, int32_t k_groups, int32_t l_groups,
                     const int32_t *user, const int32_t *item, const int32_t *rating,
                     int swap_sides, mmsbm_hip_ctx **out) {
  return guarded([&] {
    if (!out) throw std::invalid_argument("null out");
    *out = nullptr;
    if (k_groups <= 0 || l_groups <= 0) throw std::invalid_argument("K and L must be positive");
    if (k_groups > 1024 || l_groups > 1024)
      throw ApiError(MMSBM_E_UNSUPPORTED, "K and L are limited to 1024 groups (64 lanes x 16 doubles per row)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw ApiError(MMSBM_E_NODEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");

    // MMSBM_HIP_TIMING=1: where the time of building a context goes (stderr)
    const bool timing = std::getenv("MMSBM_HIP_TIMING") != nullptr;
    auto clk = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[mmsbm_hip_create] %-28s %8.2f ms\n", what,
                   std::chrono::duration<double, std::milli>(now - clk).count());
      clk = now;
    };
    std::unique_ptr<mmsbm_hip_ctx> c(new mmsbm_hip_ctx());
    c->device = device;
    c->swapped = swap_sides > 0 || (swap_sides < 0 && n_users < n_items);
    c->n_obs = n_obs;
    c->ext_users = n_users; c->ext_items = n_items; c->ext_k = k_groups; c->ext_l = l_groups;
    c->n_ratings = n_ratings;
    const int32_t *iu = user, *ii = item;
    if (c->swapped) {
      c->n_users = n_items; c->n_items = n_users; c->k = l_groups; c->l = k_groups;
      iu = item; ii = user;
    } else {
      c->n_users = n_users; c->n_items = n_items; c->k = k_groups; c->l = l_groups;
    }
    c->kp = pad_dim(c->k); c->lp = pad_dim(c->l);
    c->code_k = group_code(c->kp); c->code_l = group_code(c->lp);
    {
      // four waves share the chunks of 4 outputs of a short row; long rows get up to 8 waves
      // (measured: 320 threads do not beat 256 at L = 20, 512 beat 256 by 15 % at L = 50)
      auto threads_for = [](int nch) { return nch <= 6 ? kBlock : kPairBlockMax; };
      c->pb_threads_t = threads_for(c->lp / 4);
      c->pb_threads_a = threads_for(c->kp / 4);
      const int nthr = c->pb_threads_t;
      c->pb_kt = ((c->kp / 2) * (c->lp / 4) <= kBlock / 2) ? 2 : 4;
      const int nslot = (c->kp / c->pb_kt) * (c->lp / 4);
      if (nslot <= nthr / 2) {
        c->pb_spb = nslot; c->pb_nacc = 1;
        const int room = (c->kp * (kUnitPairs + 1) + kUnitPairs * c->lp) /
                         (nslot * 4 * c->pb_kt);  // hand-over area
        c->pb_nsub = std::max(1, std::min(std::min(nthr / nslot, 8), 1 + room));
      }
      else { c->pb_spb = nthr; int n = 1; while (n * nthr < nslot) n *= 2; c->pb_nacc = n; }
    }
    // the rating tile sits in LDS when it is too big for the scalar cache -- unless that does not
    // fit beside the rows, then it is read through scalar loads after all (slower, but it runs)
    c->tl_t = tile_in_lds(c->kp, c->lp);
    c->tl_a = tile_in_lds(c->lp, c->kp);
    c->lds_t = pair_block_lds(c->kp, c->lp, c->tl_t);
    c->lds_a = pair_block_lds(c->lp, c->kp, c->tl_a);
    if (c->lds_t > kLdsMax) { c->tl_t = false; c->lds_t = pair_block_lds(c->kp, c->lp, false); }
    if (c->lds_a > kLdsMax) { c->tl_a = false; c->lds_a = pair_block_lds(c->lp, c->kp, false); }
    // still too large for the 64-pair LDS stage (roughly K + L > 300): the plain wide-row kernels
    c->wide = c->lds_t > kLdsMax || c->lds_a > kLdsMax || c->pb_nacc > 4 ||
              std::getenv("MMSBM_HIP_FORCE_WIDE") != nullptr;
    c->split_rows = true;
    if (n_ratings > 65535)
      throw ApiError(MMSBM_E_UNSUPPORTED, "more than 65535 distinct ratings are not supported");

    lap("checks");
    // The sorts: on the host for small inputs (14 ms at 1M ratings), on the device beyond
    // kGpuLayoutMin triples (layout_gpu.hpp; MMSBM_HIP_GPU_LAYOUT=0/1 forces either).  The id columns
    // are uploaded first in both cases (the element-wise kernels keep them in the original order).
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    bool gpu_layout = n_obs >= kGpuLayoutMin;
    if (const char *g = std::getenv("MMSBM_HIP_GPU_LAYOUT")) gpu_layout = std::atoi(g) != 0;
    // (the device sort packs (rating, item) into 31 bits; sparser key spaces stay on the host)
    if (static_cast<uint64_t>(n_ratings) * static_cast<uint64_t>(c->n_items) >= (uint64_t(1) << 31)) gpu_layout = false;
    mmsbm::gpu_layout::DeviceArrays dev_idx;
    if (gpu_layout) {
      mmsbm::validate_triples(n_obs, c->n_users, c->n_items, n_ratings, iu, ii, rating);
      lap("id checks");
    }
    {
      const size_t bytes = sizeof(int32_t) * static_cast<size_t>(n_obs);
      c->orig_u.alloc(n_obs); c->orig_i.alloc(n_obs); c->orig_r.alloc(n_obs);
      if (n_obs > 0 && gpu_layout) {  // (host layout: uploaded after its own id checks, below)
        HIP_CHECK(hipMemcpyAsync(c->orig_u.ptr, iu, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->orig_i.ptr, ii, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->orig_r.ptr, rating, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));  // the caller's buffers are free again
        lap("id upload");
      }
    }
    if (gpu_layout) {
      try {
        mmsbm::gpu_layout::sort_stage(c->stream, n_obs, c->n_users, c->n_items, n_ratings, c->orig_u.ptr,
                                      c->orig_i.ptr, c->orig_r.ptr, c->lay, dev_idx);
      } catch (const std::invalid_argument &) {
        throw;
      } catch (const std::exception &e) {
        throw ApiError(MMSBM_E_HIP, std::string("device layout: ") + e.what());
      }
      c->pair_user.ptr = dev_idx.pair_user; c->pair_user.count = static_cast<size_t>(n_obs);
      c->user_pair.ptr = dev_idx.user_pair; c->user_pair.count = static_cast<size_t>(n_obs);
      mmsbm::finish_layout(c->lay, 512);
      lap("device layout (sorts)");
    } else {
      mmsbm::build_layout(n_obs, c->n_users, c->n_items, n_ratings, iu, ii, rating, 512, c->lay);
      lap("host layout (sorts)");
    }
    // big K x L tiles: four 64-pair units per pair_block workgroup (4x fewer slabs to write + add)
    const std::vector<mmsbm::Chunk> units64 = c->lay.mv_chunks;  // likelihood_units_kernel: <= 64 pairs
    c->n_lik_units = static_cast<int>(units64.size());
    // ... and both launches on the matrix cores where the tile has left the scalar cache (pair_mfma_kernel),
    // with eight units per workgroup while that still leaves every CU a few rounds of workgroups (C5: T+S
    // 358 -> 342 us, half the slabs for eta_p: 123 -> 111 us; 768 or 1,024 pairs per workgroup are slower)
    c->lds_mt = pair_mfma_lds(c->kp, c->lp, true);
    c->lds_ma = pair_mfma_lds(c->lp, c->kp, false);
    c->mfma = mfma_possible(c.get()) && c->kp * c->lp > 1024 && std::getenv("MMSBM_HIP_NO_MFMA") == nullptr;
    // K or L beyond 64: the blocked matrix-core kernels take over from the lane-per-pair stage with its tile in
    // scalar loads and from the wide-row kernels (skinny shapes -- a side below 16 -- keep those: one
    // 16 x 16 tile would be mostly padding)
    c->mfma_big = !c->mfma && c->kp * c->lp > 1024 && std::min(c->kp, c->lp) >= 16 &&
                  std::getenv("MMSBM_HIP_NO_MFMA") == nullptr;
    int big_chunk = 4 * mmsbm::kMvChunkPairs;
    if (c->mfma && c->lay.n_pairs >= 2 * big_chunk * 4 * c->n_cus) big_chunk *= 2;
    if (const char *e = std::getenv("MMSBM_HIP_MFMA_CHUNK")) big_chunk = std::min(std::max(std::atoi(e) / 64 * 64, 64), kMfmaChunkPairs);  // (tuning)
    if (c->wide) mmsbm::build_mv_chunks(c->lay, kWideChunkPairs);
    else if (c->kp * c->lp > 1024) mmsbm::build_mv_chunks(c->lay, big_chunk);
    c->mv_chunk_pairs = c->wide ? kWideChunkPairs : (c->kp * c->lp > 1024 ? big_chunk : mmsbm::kMvChunkPairs);
    // long rows: the mat-vec's outputs go to memory straight from registers (C5: -6 % on both
    // pair_block launches); short rows are cheaper transposed through LDS and copied out flat
    // (C3: direct stores cost +1.1 / +1.7 us)
    c->direct_out = c->kp * c->lp > 1024;
    // ... and the A launch as a persistent four-unit pipeline where the tile sits in LDS and
    // everything fits (C5: 312 -> 259 us)
    c->lds_qa = (static_cast<size_t>(kQuadUnits) * c->lp * (kUnitPairs + 1) + static_cast<size_t>(c->lp) * c->kp) *
                sizeof(double);
    c->quad_a = !c->wide && c->kp * c->lp > 1024 && c->tl_a && c->pb_threads_a == kPairBlockMax &&
                c->lds_qa <= kLdsMax - 2048 && c->lp <= 64 && big_chunk == 4 * mmsbm::kMvChunkPairs;
    c->n_pairs = c->lay.n_pairs;
    c->n_chunks = static_cast<int>(c->lay.mv_chunks.size());
    // dense data: XCD-local work lists (layout.hpp) -- every segment cut at fixed borders of the
    // gathered index, each range's work on one XCD, whose L2 then holds that slice of the table
    if (std::getenv("MMSBM_HIP_NO_RANGES") == nullptr) {
      const int per = kBlock / group_lanes(c->code_k);
      const size_t row_bytes = static_cast<size_t>(c->kp) * sizeof(double);
      const int64_t mean_p = c->n_pairs > 0 ? n_obs / c->n_pairs : 0, mean_u = n_obs / std::max(c->n_users, 1);
      int rp = mmsbm::range_count(static_cast<size_t>(c->n_users) * row_bytes, mean_p);   // pair pass gathers theta
      int ru = mmsbm::range_count(static_cast<size_t>(c->n_pairs) * row_bytes, mean_u);   // user pass gathers A
      // ... unless the table's hot rows (what one L2 keeps by itself) already take most of the gathers:
      // theta rows are gathered once per triple of that user, A rows once per triple of that pair
      // Where the rows one L2 keeps by itself already take half of the gathers (heavy-tailed gather
      // counts, or a table only a few times an L2) there is little left to win: measured +16 % (50M
      // ratings, log-normal item popularity, 8.5 MB table) and +24 % (Zipf(1.2) degrees) if cut anyway.
      const int64_t fit = static_cast<int64_t>(mmsbm::kRangeSliceBytes / row_bytes);
      if (rp > 1 && mmsbm::hot_fraction(c->lay.user_off, fit) > 0.5) rp = 1;
      if (ru > 1 && mmsbm::hot_fraction(c->lay.pair_off, fit) > 0.5) ru = 1;
      if (const char *f = std::getenv("MMSBM_HIP_RANGES")) {  // tuning: "pairs,users" forced range counts
        int a = 0, b = 0;
        if (std::sscanf(f, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 1 && a <= 512 && b <= 512) { rp = a; ru = b; }
      }
      // (device layout: the index arrays live on the device, so the borders are found there and only the
      // cut positions -- segments x (ranges + 1) integers -- come back)
      std::vector<int32_t> cuts_p, cuts_u;
      if (gpu_layout && rp > 1) cuts_p = mmsbm::gpu_layout::range_cuts(c->stream, c->lay.pair_off, c->pair_user.ptr, c->n_users, rp);
      if (gpu_layout && ru > 1) cuts_u = mmsbm::gpu_layout::range_cuts(c->stream, c->lay.user_off, c->user_pair.ptr, c->n_pairs, ru);
      if (rp > 1)
        mmsbm::build_worklist_ranges(c->lay.pair_off, c->lay.pair_user.data(), c->n_users, rp,
                                     mmsbm::item_length(n_obs, c->n_pairs), per, c->lay.pair_work,
                                     gpu_layout ? cuts_p.data() : nullptr);
      if (ru > 1)
        mmsbm::build_worklist_ranges(c->lay.user_off, c->lay.user_pair.data(), c->n_pairs, ru,
                                     mmsbm::item_length(n_obs, c->n_users), per, c->lay.user_work,
                                     gpu_layout ? cuts_u.data() : nullptr);
      c->ranges_pairs = rp;
      c->ranges_users = ru;
    }
    lap("xcd-local work lists");

    {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        c->n_cus = cus;
    }
    hipStream_t s = c->stream;
    c->pair_off.upload(c->lay.pair_off, s);
    c->pair_item.upload(c->lay.pair_item, s);
    c->user_off.upload(c->lay.user_off, s);
    if (!gpu_layout) {  // (the device layout left these two where they were built)
      c->pair_user.upload(c->lay.pair_user, s);
      c->user_pair.upload(c->lay.user_pair, s);
    }
    c->item_off.upload(c->lay.item_off, s);
    c->item_pairs.upload(c->lay.item_pairs, s);
    c->item_deg.upload(c->lay.item_deg, s);
    // at least half of all (item, rating) combinations occur and R is small: fixed-width grid of pair ids
    if (n_ratings <= 16 && static_cast<int64_t>(c->n_pairs) * 2 >= static_cast<int64_t>(c->n_items) * n_ratings &&
        std::getenv("MMSBM_HIP_NO_ITEMGRID") == nullptr) {
      std::vector<int32_t> grid(static_cast<size_t>(c->n_items) * n_ratings, -1);
      for (int r = 0; r < n_ratings; ++r)
        for (int32_t q = c->lay.rating_off[static_cast<size_t>(r)]; q < c->lay.rating_off[static_cast<size_t>(r) + 1]; ++q)
          grid[static_cast<size_t>(c->lay.pair_item[static_cast<size_t>(q)]) * n_ratings + r] = q;
      c->item_grid.upload(grid, s);
      HIP_CHECK(hipStreamSynchronize(s));  // `grid` is a local
    }
    c->mv_chunks.upload(c->lay.mv_chunks, s);
    c->lik_units.upload(units64, s);
    c->mv_chunk_off.upload(c->lay.mv_chunk_off, s);
    c->pair_items.upload(c->lay.pair_work.items, s);
    c->user_items.upload(c->lay.user_work.items, s);
    c->pair_splits.upload(c->lay.pair_work.splits, s);
    c->user_splits.upload(c->lay.user_work.splits, s);
    if (!gpu_layout && n_obs > 0) {
      const size_t bytes = sizeof(int32_t) * static_cast<size_t>(n_obs);
      HIP_CHECK(hipMemcpyAsync(c->orig_u.ptr, iu, bytes, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(c->orig_i.ptr, ii, bytes, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(c->orig_r.ptr, rating, bytes, hipMemcpyHostToDevice, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));  // nothing of the caller's (or this function's) host memory is still being read
    lap("index uploads");
    alloc_state(c.get(), 1);
    c->lik_part.alloc(4096);
    HIP_CHECK(hipStreamSynchronize(s));
    lap("state allocation");
    *out = c.release();
  });
}

int mmsbm_hip_destroy(mmsbm_hip_ctx *ctx) {
  return guarded([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    delete ctx;
  });
}

int mmsbm_hip_dims(const mmsbm_hip_ctx *ctx, int64_t dims[8]) {
  return guarded([&] {
    if (!ctx || !dims) throw std::invalid_argument("null argument");
    dims[0] = ctx->n_obs; dims[1] = ctx->ext_users; dims[2] = ctx->ext_items;
    dims[3] = ctx->n_ratings; dims[4] = ctx->ext_k; dims[5] = ctx->ext_l;
    dims[6] = ctx->n_pairs; dims[7] = ctx->swapped ? 1 : 0;
  });
}

int mmsbm_hip_degrees(const mmsbm_hip_ctx *ctx, int64_t *d_user, int64_t *d_item) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    const mmsbm::Layout &L = ctx->lay;
    int64_t *du = ctx->swapped ? d_item : d_user;  // internal users
    int64_t *di = ctx->swapped ? d_user : d_item;  // internal items
    if (du)
      for (int u = 0; u < L.n_users; ++u)
        du[u] = std::max<int64_t>(L.user_off[u + 1] - L.user_off[u], 1);
    if (di)
      for (int i = 0; i < L.n_items; ++i) di[i] = std::max<int64_t>(L.item_deg[i], 1);
  });
}

int mmsbm_hip_set_params(mmsbm_hip_ctx *ctx, const double *theta, const double *eta,
                         const double *pr) {
  return guarded([&] {
    if (!ctx || !theta || !eta || !pr) throw std::invalid_argument("null argument");
    use_device(ctx);
    OneSlot one(ctx);
    const double *it = ctx->swapped ? eta : theta;  // internal theta rows = internal users
    const double *ie = ctx->swapped ? theta : eta;
    const int cur = ctx->cur, sl = ctx->sel;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // nothing in flight still reads the staging area
    ctx->pin.reset(rows_doubles(ctx));
    upload_rows(ctx, theta_tab(ctx, cur), it, ctx->n_users, ctx->k);
    upload_rows(ctx, plain_tab(ctx->eta[cur].at(sl), ctx->lp), ie, ctx->n_items, ctx->l);
    const size_t klr = static_cast<size_t>(ctx->n_ratings) * ctx->kp * ctx->lp;
    double *p = ctx->pin.take(klr), *pt = ctx->pin.take(klr);
    p_host_to_dev(ctx, pr, p, pt);
    HIP_CHECK(hipMemcpyAsync(ctx->p[cur].at(sl), p, sizeof(double) * klr, hipMemcpyHostToDevice,
                             ctx->stream));
    HIP_CHECK(hipMemcpyAsync(ctx->pt[cur].at(sl), pt, sizeof(double) * klr, hipMemcpyHostToDevice,
                             ctx->stream));
    stage_matvec_a(ctx, cur, cur);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the staging area is free again
    ctx->have[sl] = 1;
  });
}

int mmsbm_hip_init_params(mmsbm_hip_ctx *ctx, const uint64_t pcg64_state[4], const double *pr) {
  return guarded([&] {
    if (!ctx || !pcg64_state || !pr) throw std::invalid_argument("null argument");
    use_device(ctx);
    OneSlot one(ctx);
    const int cur = ctx->cur, sl = ctx->sel;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const size_t klr = static_cast<size_t>(ctx->n_ratings) * ctx->kp * ctx->lp;
    ctx->pin.reset(2 * klr);
    double *p = ctx->pin.take(klr), *pt = ctx->pin.take(klr);
    p_host_to_dev(ctx, pr, p, pt);
    HIP_CHECK(hipMemcpyAsync(ctx->p[cur].at(sl), p, sizeof(double) * klr, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(ctx->pt[cur].at(sl), pt, sizeof(double) * klr, hipMemcpyHostToDevice, ctx->stream));
    // the reference draws theta (external users x K) first, then eta; internally the two sides
    // may be swapped, the stream offsets are not
    const uint64_t n_theta_ext = static_cast<uint64_t>(ctx->ext_users) * ctx->ext_k;
    const uint64_t off_users = ctx->swapped ? n_theta_ext : 0;  // internal users' table
    const uint64_t off_items = ctx->swapped ? 0 : n_theta_ext;
    const RowTab tt = theta_tab(ctx, cur), et = plain_tab(ctx->eta[cur].at(sl), ctx->lp);
    zero_rows(ctx, tt, ctx->n_users);  // padding columns
    zero_rows(ctx, et, ctx->n_items);
    auto blocks = [](uint64_t total) {
      return static_cast<unsigned>((total + uint64_t(kBlock) * kDrawsPerThread - 1) / (uint64_t(kBlock) * kDrawsPerThread));
    };
    const uint64_t nu = static_cast<uint64_t>(ctx->n_users) * ctx->k, ni = static_cast<uint64_t>(ctx->n_items) * ctx->l;
    if (nu > 0)
      init_rows_kernel<<<blocks(nu), kBlock, 0, ctx->stream>>>(tt, ctx->user_off.ptr, nullptr, ctx->n_users, ctx->k,
                                                               pcg64_state[0], pcg64_state[1], pcg64_state[2],
                                                               pcg64_state[3], off_users);
    if (ni > 0)
      init_rows_kernel<<<blocks(ni), kBlock, 0, ctx->stream>>>(et, nullptr, ctx->item_deg.ptr, ctx->n_items, ctx->l,
                                                               pcg64_state[0], pcg64_state[1], pcg64_state[2],
                                                               pcg64_state[3], off_items);
    HIP_CHECK(hipGetLastError());
    stage_matvec_a(ctx, cur, cur);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->have[sl] = 1;
  });
}

int mmsbm_hip_pcg64_doubles(const uint64_t pcg64_state[4], uint64_t offset, int64_t n, double *out) {
  return guarded([&] {
    if (!pcg64_state || (n > 0 && !out) || n < 0) throw std::invalid_argument("bad argument");
    pcg64::Stream g{pcg64::make128(pcg64_state[0], pcg64_state[1]), pcg64::make128(pcg64_state[2], pcg64_state[3])};
    pcg64::advance(g, offset);
    for (int64_t j = 0; j < n; ++j) out[j] = pcg64::next_double(g);
  });
}

int mmsbm_hip_get_params(mmsbm_hip_ctx *ctx, double *theta, double *eta, double *pr) {
  return guarded([&] {
    require_params(ctx);
    use_device(ctx);
    OneSlot one(ctx);
    double *it = ctx->swapped ? eta : theta;
    double *ie = ctx->swapped ? theta : eta;
    const int cur = ctx->cur, sl = ctx->sel;
    fetch_params(ctx, theta_tab(ctx, cur), plain_tab(ctx->eta[cur].at(sl), ctx->lp),
                 ctx->p[cur].at(sl), it, ie, pr);
  });
}

int mmsbm_hip_set_slots(mmsbm_hip_ctx *ctx, int n_slots) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (n_slots < 1 || n_slots > 65535) throw std::invalid_argument("n_slots must be in [1, 65535]");
    use_device(ctx);
    if (n_slots == ctx->n_slots) {
      ctx->have.assign(static_cast<size_t>(n_slots), 0);
      ctx->sel = 0;
      return;
    }
    try {
      alloc_state(ctx, n_slots);
    } catch (...) {  // most likely out of device memory: leave a consistent one-slot context
      (void)hipGetLastError();
      try { alloc_state(ctx, 1); } catch (...) {}
      throw;
    }
  });
}

int mmsbm_hip_select_slot(mmsbm_hip_ctx *ctx, int slot) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (slot < 0 || slot >= ctx->n_slots)
      throw std::invalid_argument("slot " + std::to_string(slot) + " out of range (the context has " +
                                  std::to_string(ctx->n_slots) + ")");
    ctx->sel = slot;
  });
}

int mmsbm_hip_slots(const mmsbm_hip_ctx *ctx, int *n_slots, int *selected,
                    int64_t *bytes_per_slot) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (n_slots) *n_slots = ctx->n_slots;
    if (selected) *selected = ctx->sel;
    if (bytes_per_slot) {
      size_t d = ctx->ctab.stride + ctx->ttab.stride + ctx->partial.stride + ctx->npr.stride +
                 ctx->pair_parts.stride + ctx->user_parts.stride;
      for (int b = 0; b < 2; ++b)
        d += ctx->theta[b].stride + ctx->eta[b].stride + ctx->p[b].stride + ctx->pt[b].stride +
             ctx->atab[b].stride;
      *bytes_per_slot = static_cast<int64_t>(d * sizeof(double));
    }
  });
}

int mmsbm_hip_em_iterate(mmsbm_hip_ctx *ctx, int n_iters) {
  return guarded([&] {
    require_all_params(ctx);
    if (n_iters < 0) throw std::invalid_argument("n_iters must be >= 0");
    use_device(ctx);
    run_iterations(ctx, n_iters);
  });
}

int mmsbm_hip_synchronize(mmsbm_hip_ctx *ctx) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    use_device(ctx);
    // Short waits are polled (hipStreamQuery returns within a microsecond or two of the last kernel;
    // a blocking wait is woken by an interrupt tens of microseconds later, which is several percent
    // of a 2 ms run of 20 iterations); after 50 ms the thread blocks like any other waiter.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(ctx->stream);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) HIP_CHECK(e);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}

int mmsbm_hip_update_coefficients(mmsbm_hip_ctx *ctx, double *n_theta, double *n_eta,
                                  double *n_pr) {
  return guarded([&] {
    require_params(ctx);
    use_device(ctx);
    OneSlot one(ctx);
    launch_iteration(ctx, false);
    const int nxt = ctx->cur ^ 1, sl = ctx->sel;
    double *it = ctx->swapped ? n_eta : n_theta;
    double *ie = ctx->swapped ? n_theta : n_eta;
    fetch_params(ctx, theta_tab(ctx, nxt), plain_tab(ctx->eta[nxt].at(sl), ctx->lp),
                 ctx->npr.at(sl), it, ie, n_pr);
  });
}

namespace {
// likelihood_fast_kernel: tile + its logarithms in LDS?  (always with several lanes per triple)
bool lik_fast_tile_lds(const mmsbm_hip_ctx *c) {
  return c->lp > 20 || c->lik_g > 1 ||
         2 * static_cast<size_t>(c->kp) * c->lp * sizeof(double) > kScalarTileBytes;
}
bool lik_fast_usable(const mmsbm_hip_ctx *c) {
  const size_t lds = lik_fast_tile_lds(c) ? 2 * static_cast<size_t>(c->kp) * c->lp * sizeof(double) : 0;
  return c->lik_fast && c->lp <= 160 && c->n_lik_units > 0 && lds <= kLdsMax - 4096;
}
// likelihood of the selected slot through the logarithm tables; returns the number of partial sums
int likelihood_fast(mmsbm_hip_ctx *c) {
  const int cur = c->cur, sl = c->sel;
  const size_t nt = static_cast<size_t>(c->n_users) * c->kp, ne = static_cast<size_t>(c->n_items) * c->lp;
  const size_t np = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  if (c->lg_theta.count < nt) c->lg_theta.alloc(nt);
  if (c->lg_eta.count < ne) c->lg_eta.alloc(ne);
  if (c->lg_p.count < np) c->lg_p.alloc(np);
  auto logs = [&](const double *in, double *out, size_t n) {
    if (n == 0) return;
    log_table_kernel<<<static_cast<unsigned>((n + kBlock - 1) / kBlock), kBlock, 0, c->stream>>>(in, out, n);
  };
  const RowTab th = theta_tab(c, cur);
  const RowTab lth{c->lg_theta.ptr, c->lg_theta.ptr + static_cast<size_t>(c->n_users) * th.mw, th.mw, th.tw,
                   th.mw, th.tw, 0, 0};  // main + tail like theta, one slot
  if (nt > 0)
    log_rows_kernel<<<static_cast<unsigned>((nt + kBlock - 1) / kBlock), kBlock, 0, c->stream>>>(
        th, lth, static_cast<size_t>(c->n_users), c->kp);
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
  likelihood_fast_kernel<LW_, G_, TL_><<<nb, kLikThreads, lds, c->stream>>>(                      \
      c->lik_units.ptr, c->pair_off.ptr, c->pair_user.ptr, c->pair_item.ptr, th, lth,            \
      c->eta[cur].at(sl), c->lg_eta.ptr, c->p[cur].at(sl), c->lg_p.ptr, c->lik_part.ptr, c->k,    \
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
}  // namespace

int mmsbm_hip_likelihood(mmsbm_hip_ctx *ctx, double *out) {
  return guarded([&] {
    require_params(ctx);
    if (!out) throw std::invalid_argument("null out");
    use_device(ctx);
    OneSlot one(ctx);
    const int cur = ctx->cur, sl = ctx->sel;
    int nb;
    if (lik_fast_usable(ctx)) {
      nb = likelihood_fast(ctx);
    } else {
    const size_t lik_lds = static_cast<size_t>(ctx->kp + ctx->lp) * kLikThreads * sizeof(double);
    if (lik_lds <= kLdsMax - 2048 && ctx->n_lik_units > 0) {
      nb = ctx->n_lik_units;
      allow_big_lds(likelihood_units_kernel, lik_lds);
      if (ctx->lik_part.count < static_cast<size_t>(nb)) ctx->lik_part.alloc(nb);
      likelihood_units_kernel<<<nb, kLikThreads, lik_lds, ctx->stream>>>(
          ctx->lik_units.ptr, ctx->pair_off.ptr, ctx->pair_user.ptr, ctx->pair_item.ptr,
          theta_tab(ctx, cur), ctx->eta[cur].at(sl), ctx->p[cur].at(sl), ctx->lik_part.ptr, ctx->k,
          ctx->l, ctx->kp, ctx->lp);
    } else {
      nb = static_cast<int>(std::min<int64_t>((ctx->n_obs + kBlock - 1) / kBlock, 4096));
      nb = std::max(nb, 1);
      likelihood_kernel<<<nb, kBlock, 0, ctx->stream>>>(
          ctx->orig_u.ptr, ctx->orig_i.ptr, ctx->orig_r.ptr, theta_tab(ctx, cur),
          ctx->eta[cur].at(sl), ctx->p[cur].at(sl), ctx->lik_part.ptr, ctx->n_obs, ctx->k, ctx->l,
          ctx->kp, ctx->lp);
    }
    }
    HIP_CHECK(hipGetLastError());
    std::vector<double> part(nb);
    HIP_CHECK(hipMemcpyAsync(part.data(), ctx->lik_part.ptr, sizeof(double) * nb,
                             hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (double v : part) tot += v;
    *out = tot;
  });
}

int mmsbm_hip_compute_omegas(mmsbm_hip_ctx *ctx, double *out, int64_t capacity_elems) {
  return guarded([&] {
    require_params(ctx);
    if (!out) throw std::invalid_argument("null out");
    use_device(ctx);
    const int64_t kl = static_cast<int64_t>(ctx->k) * ctx->l;
    const int64_t n_elems = ctx->n_obs * kl;
    if (n_elems > capacity_elems)
      throw ApiError(MMSBM_E_TOOLARGE, "omega tensor larger than the caller's buffer");
    if (n_elems > (int64_t(1) << 31))  // 16 GiB: the factorised path exists so nobody needs this
      throw ApiError(MMSBM_E_TOOLARGE, "omega tensor above the 2^31-element cap");
    if (n_elems == 0) return;
    DevBuf<double> dev;
    dev.alloc(static_cast<size_t>(n_elems));
    OneSlot one(ctx);
    const int cur = ctx->cur, sl = ctx->sel;
    // internal (k,l) -> external position: not swapped [k][l] strides (L,1); swapped the
    // external tensor is [l_int][k_int] so strides are (1, K_int)
    const int sk = ctx->swapped ? 1 : ctx->l;
    const int sl_stride = ctx->swapped ? ctx->k : 1;
    const int64_t nb = (n_elems + kBlock - 1) / kBlock;
    omegas_kernel<<<static_cast<unsigned>(nb), kBlock, 0, ctx->stream>>>(
        ctx->orig_u.ptr, ctx->orig_i.ptr, ctx->orig_r.ptr, theta_tab(ctx, cur),
        ctx->eta[cur].at(sl), ctx->p[cur].at(sl), dev.ptr, n_elems, ctx->k, ctx->l, ctx->kp,
        ctx->lp, sk, sl_stride);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out, dev.ptr, sizeof(double) * n_elems, hipMemcpyDeviceToHost,
                             ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}

namespace {
// ---- prod_dist / predict through B = p_r eta_i over every (item, rating) combination ---------------
// Worth it when the rows to score are not far fewer than the items (B costs I R K L multiply-adds, a row
// then R K instead of R K L) and B fits comfortably: otherwise the per-row kernels run.
bool rows_fast_ok(const mmsbm_hip_ctx *c, int64_t n_rows) {
  if (!c->predict_fast || n_rows <= 0) return false;
  const uint64_t combos = static_cast<uint64_t>(c->n_items) * static_cast<uint64_t>(c->n_ratings);
  if (combos >= (uint64_t(1) << 31) - 2048 || n_rows * 4 < c->n_items) return false;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  return combos * static_cast<uint64_t>(c->kp) * sizeof(double) <= (free_b + c->btab.count * sizeof(double)) / 2;
}
void ensure_pair_grid(mmsbm_hip_ctx *c) {  // q = r * I + i, chunks of the pair stage's size per rating
  const size_t combos = static_cast<size_t>(c->n_items) * c->n_ratings;
  if (c->grid_item.count != combos || c->grid_n_chunks == 0) {
    std::vector<int32_t> item(combos);
    std::vector<mmsbm::Chunk> chunks;
    for (int r = 0; r < c->n_ratings; ++r) {
      const int32_t base = static_cast<int32_t>(static_cast<size_t>(r) * c->n_items);
      for (int i = 0; i < c->n_items; ++i) item[static_cast<size_t>(base) + i] = i;
      for (int i = 0; i < c->n_items; i += c->mv_chunk_pairs)
        chunks.push_back(mmsbm::Chunk{r, base + i, base + std::min(i + c->mv_chunk_pairs, c->n_items), 0});
    }
    c->grid_item.upload(item, c->stream);
    c->grid_chunks.upload(chunks, c->stream);
    HIP_CHECK(hipStreamSynchronize(c->stream));  // (host vectors are locals)
    c->grid_n_chunks = static_cast<int>(chunks.size());
  }
  if (c->btab.count < combos * c->kp) c->btab.alloc(combos * c->kp);
}
// eligibility + the table's memory; false (and no error left behind) sends the caller to the per-row kernels
bool rows_fast_prepare(mmsbm_hip_ctx *c, int64_t n_rows) {
  if (!rows_fast_ok(c, n_rows)) return false;
  try {
    ensure_pair_grid(c);
  } catch (const ApiError &) {  // out of memory after all (another context took it): not an error of this call
    (void)hipGetLastError();
    c->btab.release();
    return false;
  }
  return true;
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
      predict_rows_kernel<G, V, 0><<<nb, kBlock, 0, c->stream>>>(pu, pi, preal, theta_tab(c, c->cur), c->btab.ptr, \
          rstride, weights, dist, block_out, n_rows, c->n_ratings, c->kp, first);                            \
    else                                                                                                     \
      predict_rows_kernel<G, V, 1><<<nb, kBlock, 0, c->stream>>>(pu, pi, preal, theta_tab(c, c->cur), c->btab.ptr, \
          rstride, weights, dist, block_out, n_rows, c->n_ratings, c->kp, first);                            \
  } while (0)
  DISPATCH_GV(c->code_k, CALL);
#undef CALL
  HIP_CHECK(hipGetLastError());
  return nb;
}

}  // namespace

int mmsbm_hip_prod_dist(mmsbm_hip_ctx *ctx, int64_t n_pairs, const int32_t *user,
                        const int32_t *item, double *out) {
  return guarded([&] {
    require_params(ctx);
    if (n_pairs < 0) throw std::invalid_argument("negative n_pairs");
    if (n_pairs == 0) return;
    if (!user || !item || !out) throw std::invalid_argument("null argument");
    for (int64_t m = 0; m < n_pairs; ++m)
      if (user[m] < 0 || user[m] >= ctx->ext_users || item[m] < 0 || item[m] >= ctx->ext_items)
        throw std::invalid_argument("prod_dist: id out of range at row " + std::to_string(m));
    use_device(ctx);
    const int64_t n_elems = n_pairs * ctx->n_ratings;
    if ((n_elems + kBlock - 1) / kBlock > (int64_t(1) << 31) - 1)
      throw ApiError(MMSBM_E_TOOLARGE, "prod_dist: too many pairs for one launch");
    DevBuf<int32_t> du, di;
    DevBuf<double> dout;
    du.alloc(n_pairs); di.alloc(n_pairs); dout.alloc(n_elems);
    const int32_t *iu = ctx->swapped ? item : user;
    const int32_t *ii = ctx->swapped ? user : item;
    HIP_CHECK(hipMemcpyAsync(du.ptr, iu, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(di.ptr, ii, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    OneSlot one(ctx);
    const int cur = ctx->cur, sl = ctx->sel;
    if (rows_fast_prepare(ctx, n_pairs)) {
      rows_launch(ctx, 0, du.ptr, di.ptr, nullptr, nullptr, dout.ptr, nullptr, n_pairs, 1);
    } else {
      const int64_t nb = (n_elems + kBlock - 1) / kBlock;
      prod_dist_kernel<<<static_cast<unsigned>(nb), kBlock, 0, ctx->stream>>>(
          du.ptr, di.ptr, theta_tab(ctx, cur), ctx->eta[cur].at(sl), ctx->p[cur].at(sl), dout.ptr,
          n_pairs, ctx->n_ratings, ctx->k, ctx->l, ctx->kp, ctx->lp);
      HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipMemcpyAsync(out, dout.ptr, sizeof(double) * n_elems, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->btab.release();  // (the B table is scratch: restart slots are sized from the free memory)
  });
}

namespace {
void score_launch(mmsbm_hip_ctx *ctx, bool finish, double *stats) {
  const bool fast = !finish && rows_fast_prepare(ctx, ctx->ps_rows);
  const int per_block = fast ? kBlock / group_lanes(ctx->code_k) : kBlock;
  const int64_t nb64 = (ctx->ps_rows + per_block - 1) / per_block;
  const int nb = static_cast<int>(nb64);
  if (ctx->ps_part.count < static_cast<size_t>(nb) * kScoreStats) ctx->ps_part.alloc(static_cast<size_t>(nb) * kScoreStats);
  const int cur = ctx->cur, sl = ctx->sel;
  if (nb > 0 && fast) {
    rows_launch(ctx, 1, ctx->ps_u.ptr, ctx->ps_i.ptr, ctx->ps_r.ptr, ctx->ps_w.ptr, ctx->ps_sum.ptr, ctx->ps_part.ptr,
                ctx->ps_rows, ctx->ps_added == 0 ? 1 : 0);
  } else if (nb > 0) {
    if (finish)
      predict_score_kernel<true><<<nb, kBlock, 0, ctx->stream>>>(
          ctx->ps_u.ptr, ctx->ps_i.ptr, ctx->ps_r.ptr, theta_tab(ctx, cur), ctx->eta[cur].at(sl),
          ctx->p[cur].at(sl), ctx->ps_w.ptr, ctx->ps_sum.ptr, ctx->ps_part.ptr, ctx->ps_rows,
          ctx->n_ratings, ctx->k, ctx->l, ctx->kp, ctx->lp, 0, static_cast<double>(ctx->ps_added));
    else
      predict_score_kernel<false><<<nb, kBlock, 0, ctx->stream>>>(
          ctx->ps_u.ptr, ctx->ps_i.ptr, ctx->ps_r.ptr, theta_tab(ctx, cur), ctx->eta[cur].at(sl),
          ctx->p[cur].at(sl), ctx->ps_w.ptr, ctx->ps_sum.ptr, ctx->ps_part.ptr, ctx->ps_rows,
          ctx->n_ratings, ctx->k, ctx->l, ctx->kp, ctx->lp, ctx->ps_added == 0 ? 1 : 0, 1.0);
    HIP_CHECK(hipGetLastError());
  }
  std::vector<double> part(static_cast<size_t>(nb) * kScoreStats);
  if (nb > 0)
    HIP_CHECK(hipMemcpyAsync(part.data(), ctx->ps_part.ptr, sizeof(double) * part.size(),
                             hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (int j = 0; j < kScoreStats; ++j) stats[j] = 0.0;
  for (int b = 0; b < nb; ++b)
    for (int j = 0; j < kScoreStats; ++j) stats[j] += part[static_cast<size_t>(b) * kScoreStats + j];
}
}  // namespace

int mmsbm_hip_predict_begin(mmsbm_hip_ctx *ctx, int64_t n_rows, const int32_t *user,
                            const int32_t *item, const int32_t *rating,
                            const double *rating_weights) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (n_rows < 0) throw std::invalid_argument("negative n_rows");
    if (!rating_weights || (n_rows > 0 && (!user || !item || !rating)))
      throw std::invalid_argument("null argument");
    if (n_rows > (int64_t(1) << 31) - kBlock)
      throw ApiError(MMSBM_E_TOOLARGE, "predict: too many rows for one launch");
    for (int64_t m = 0; m < n_rows; ++m)
      if (user[m] < 0 || user[m] >= ctx->ext_users || item[m] < 0 || item[m] >= ctx->ext_items ||
          rating[m] < 0 || rating[m] >= ctx->n_ratings)
        throw std::invalid_argument("predict: id out of range at row " + std::to_string(m));
    use_device(ctx);  // (arguments are fine: from here on the previous session is gone)
    ctx->ps_rows = -1;
    const int32_t *iu = ctx->swapped ? item : user;
    const int32_t *ii = ctx->swapped ? user : item;
    hipStream_t s = ctx->stream;
    ctx->ps_u.alloc(n_rows); ctx->ps_i.alloc(n_rows); ctx->ps_r.alloc(n_rows);
    ctx->ps_sum.alloc(static_cast<size_t>(n_rows) * ctx->n_ratings);
    ctx->ps_w.alloc(ctx->n_ratings);
    if (n_rows > 0) {
      HIP_CHECK(hipMemcpyAsync(ctx->ps_u.ptr, iu, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(ctx->ps_i.ptr, ii, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(ctx->ps_r.ptr, rating, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
    }
    HIP_CHECK(hipMemcpyAsync(ctx->ps_w.ptr, rating_weights, sizeof(double) * ctx->n_ratings,
                             hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));  // the caller's buffers are free again
    ctx->ps_rows = n_rows;
    ctx->ps_added = 0;
  });
}

int mmsbm_hip_predict_add(mmsbm_hip_ctx *ctx, double stats[6]) {
  return guarded([&] {
    require_params(ctx);
    if (!stats) throw std::invalid_argument("null stats");
    if (ctx->ps_rows < 0) throw std::invalid_argument("predict_begin has not been called");
    use_device(ctx);
    OneSlot one(ctx);
    score_launch(ctx, false, stats);
    ctx->ps_added++;
  });
}

int mmsbm_hip_predict_finish(mmsbm_hip_ctx *ctx, double *mean_dist, double stats[6]) {
  return guarded([&] {
    if (!ctx || !stats) throw std::invalid_argument("null argument");
    if (ctx->ps_rows < 0) throw std::invalid_argument("predict_begin has not been called");
    if (ctx->ps_added < 1) throw std::invalid_argument("predict_finish before any predict_add");
    use_device(ctx);
    OneSlot one(ctx);
    const int64_t rows = ctx->ps_rows;
    score_launch(ctx, true, stats);
    ctx->ps_rows = -1;  // the session is over whatever happens next
    ctx->btab.release();
    if (mean_dist && rows > 0) {
      HIP_CHECK(hipMemcpyAsync(mean_dist, ctx->ps_sum.ptr, sizeof(double) * rows * ctx->n_ratings,
                               hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
  });
}

int mmsbm_hip_time_iterations(mmsbm_hip_ctx *ctx, int n_iters, float *elapsed_ms) {
  return guarded([&] {
    require_all_params(ctx);
    if (!elapsed_ms || n_iters < 0) throw std::invalid_argument("bad argument");
    use_device(ctx);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, ctx->stream));
    run_iterations(ctx, n_iters);
    HIP_CHECK(hipEventRecord(e1, ctx->stream));
    HIP_CHECK(hipEventSynchronize(e1));
    HIP_CHECK(hipEventElapsedTime(elapsed_ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

int mmsbm_hip_kernel_count(void) { return K_COUNT; }

const char *mmsbm_hip_kernel_name(int index) {
  return (index >= 0 && index < K_COUNT) ? kKernelNames[index] : "";
}

int mmsbm_hip_profile_iterations(mmsbm_hip_ctx *ctx, int n_iters, float *mean_us,
                                 int *launches_per_iter) {
  return guarded([&] {
    require_all_params(ctx);
    if (!mean_us || n_iters <= 0) throw std::invalid_argument("bad argument");
    use_device(ctx);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->profiling = true;
    try {
      for (int it = 0; it < n_iters; ++it) launch_iteration(ctx, true);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } catch (...) {
      ctx->profiling = false;
      throw;
    }
    ctx->profiling = false;
    collect_profile(ctx, mean_us, launches_per_iter, n_iters);
  });
}

int mmsbm_hip_kernel_bytes(const mmsbm_hip_ctx *ctx, int index, int64_t *bytes_read,
                           int64_t *bytes_written) {
  return guarded([&] {
    if (!ctx || !bytes_read || !bytes_written) throw std::invalid_argument("null argument");
    const int64_t N = ctx->n_obs, U = ctx->n_users, I = ctx->n_items, R = ctx->n_ratings;
    const int64_t K = ctx->k, L = ctx->l, Q = ctx->n_pairs;
    const int64_t C = static_cast<int64_t>(ctx->lay.mv_chunks.size());
    int64_t rd = 0, wr = 0;
    switch (index) {
      case K_SEG:  // two passes: index + one gathered K-row per triple; fixed rows and offsets once
        rd = 2 * N * (4 + 8 * K) + (U + Q) * (8 * K + 4);
        wr = (U + Q) * 8 * K;
        break;
      case K_DENSE:  // C rows + gathered eta rows + item ids + one tile per block; T rows + slabs
        rd = Q * (8 * K + 8 * L + 4) + C * 8 * K * L;
        wr = Q * 8 * L + C * 8 * K * L;
        break;
      case K_ETAP:  // slabs + p ; T rows through the item CSR + eta
        rd = C * 8 * K * L + R * 8 * K * L + Q * (8 * L + 4) + I * (8 * L + 8);
        wr = 3 * R * 8 * K * L + I * 8 * L;
        break;
      case K_MATVEC_A: rd = Q * (8 * L + 4) + C * 8 * K * L; wr = Q * 8 * K; break;
      default: throw std::invalid_argument("kernel index out of range");
    }
    *bytes_read = rd * ctx->n_slots;  // one launch covers every restart slot
    *bytes_written = wr * ctx->n_slots;
  });
}

int mmsbm_hip_time_stage(mmsbm_hip_ctx *ctx, int stage, int reps, float *mean_us) {
  return guarded([&] {
    require_all_params(ctx);
    ctx->ablate = stage >> 8;  // bits 8.. : phases to skip (timing experiments only)
    stage &= 0xff;
    struct Reset { mmsbm_hip_ctx *c; ~Reset() { c->ablate = 0; } } reset{ctx};
    if (!mean_us || reps <= 0 || stage < 0 || stage >= K_COUNT)
      throw std::invalid_argument("bad argument");
    use_device(ctx);
    auto one = [&] {
      switch (stage) {
        case K_SEG: stage_seg(ctx, true, true, true, ctx->stream); break;
        case K_DENSE: stage_dense(ctx); break;
        case K_ETAP: stage_eta_p(ctx, true); break;
        default: stage_matvec_a(ctx, ctx->cur, ctx->cur ^ 1); break;
      }
    };
    for (int w = 0; w < 3; ++w) one();
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; ++r) one();
    HIP_CHECK(hipEventRecord(e1, ctx->stream));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *mean_us = ms * 1000.f / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

int mmsbm_hip_set_option(mmsbm_hip_ctx *ctx, const char *name, double value) {
  return guarded([&] {
    if (!ctx || !name) throw std::invalid_argument("null argument");
    const std::string key(name);
    if (key == "graph") {
      ctx->graph_mode = value != 0.0;

    } else if (key == "slot_waves") {  // 0: restart slots as separate workgroups (blockIdx.y) in the triple passes
      ctx->slot_waves = value != 0.0;
    } else if (key == "lik_fast") {
      ctx->lik_fast = value != 0.0;
    } else if (key == "lik_g") {
      const int g = static_cast<int>(value);
      if (g != 0 && g != 1 && g != 2 && g != 4 && g != 8) throw std::invalid_argument("lik_g: 0, 1, 2, 4 or 8");
      ctx->lik_g = g;
    } else if (key == "direct") {
      ctx->direct_out = value != 0.0;
    } else if (key == "quad") {  // 0: the A launch through pair_block like every other shape
      ctx->quad_a = value != 0.0 && !ctx->wide && ctx->kp * ctx->lp > 1024 && ctx->tl_a &&
                    ctx->pb_threads_a == kPairBlockMax && ctx->lds_qa <= kLdsMax - 2048 && ctx->lp <= 64;
    } else if (key == "predict_fast") {  // 0: prod_dist / predict through the per-row kernels (R K L multiply-adds per row)
      ctx->predict_fast = value != 0.0;
    } else if (key == "mfma_threads") {
      if (value != kBlock && value != kPairBlockMax) throw std::invalid_argument("mfma_threads: 256 or 512");
      ctx->mfma_threads = static_cast<int>(value);
    } else if (key == "mfma") {  // the pair stage on the matrix cores: 0 off, 1 on (one-block form if K, L <= 64,
                                 // else the blocked form), 2 the blocked form whatever the shape
      ctx->mfma = value == 1.0 && mfma_possible(ctx);
      ctx->mfma_big = value != 0.0 && !ctx->mfma && ctx->mv_chunk_pairs <= kMfmaChunkPairs;
    } else {
      throw std::invalid_argument("unknown option: " + key);
    }
    ctx->drop_graphs();
  });
}

int mmsbm_hip_get_option(const mmsbm_hip_ctx *ctx, const char *name, double *value) {
  return guarded([&] {
    if (!ctx || !name || !value) throw std::invalid_argument("null argument");
    const std::string key(name);
    if (key == "graph") *value = ctx->graph_mode;
    else if (key == "direct") *value = ctx->direct_out;
    else if (key == "quad") *value = ctx->quad_a;
    else if (key == "mfma") *value = ctx->mfma ? 1.0 : (ctx->mfma_big ? 2.0 : 0.0);
    else if (key == "mfma_threads") *value = ctx->mfma_threads;
    else if (key == "predict_fast") *value = ctx->predict_fast;
    else if (key == "wide") *value = ctx->wide;
    else if (key == "slot_waves") *value = ctx->slot_waves;
    else if (key == "lik_fast") *value = ctx->lik_fast;
    else if (key == "lik_g") *value = ctx->lik_g;
    else if (key == "ranges_pairs") *value = ctx->ranges_pairs;   // read-only: XCD-local work lists,
    else if (key == "ranges_users") *value = ctx->ranges_users;   // ranges per pass (1 = off)
    else if (key == "items_pairs") *value = static_cast<double>(ctx->lay.pair_work.items.size());
    else if (key == "items_users") *value = static_cast<double>(ctx->lay.user_work.items.size());
    else throw std::invalid_argument("unknown option: " + key);
  });
}

int mmsbm_hip_set_graph_mode(mmsbm_hip_ctx *ctx, int enabled) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    ctx->graph_mode = enabled != 0;
    ctx->drop_graphs();
  });
}

#ifdef MMSBM_STAMPS
// diagnostic build: copy the phase stamps of the last pair-stage launch (blocks x 16 words)
int mmsbm_hip_debug_stamps(unsigned long long *out, int n_words) {
  return guarded([&] {
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n_words));
  });
}
#endif

// ---- host-only layout helpers (no device needed; used by the CPU tests) --------------------
struct mmsbm_hip_layout {
  mmsbm::Layout lay;
};

int mmsbm_hip_layout_build(int64_t n_obs, int32_t n_users, int32_t n_items, int32_t n_ratings,
                           const int32_t *user, const int32_t *item, const int32_t *rating,
                           int32_t target_chunks, mmsbm_hip_layout **out) {
  return guarded([&] {
    if (!out) throw std::invalid_argument("null out");
    *out = nullptr;
    std::unique_ptr<mmsbm_hip_layout> h(new mmsbm_hip_layout());
    mmsbm::build_layout(n_obs, n_users, n_items, n_ratings, user, item, rating, target_chunks,
                        h->lay);
    *out = h.release();
  });
}

int mmsbm_hip_layout_free(mmsbm_hip_layout *h) {
  delete h;
  return MMSBM_OK;
}

// which: 0 pair_off 1 pair_user 2 pair_item 3 rating_off 4 user_off 5 user_pair 6 item_off
//        7 item_pairs 8 item_deg 9 chunk_off; 4-int records: 10 chunks 11 mv_chunks
//        12 pair work items 13 user work items 14 pair splits 15 user splits
int mmsbm_hip_layout_array(const mmsbm_hip_layout *h, int which, int32_t *out, int64_t capacity,
                           int64_t *count) {
  return guarded([&] {
    if (!h || !count) throw std::invalid_argument("null argument");
    const mmsbm::Layout &L = h->lay;
    const std::vector<int32_t> *v = nullptr;
    switch (which) {
      case 0: v = &L.pair_off; break;
      case 1: v = &L.pair_user; break;
      case 2: v = &L.pair_item; break;
      case 3: v = &L.rating_off; break;
      case 4: v = &L.user_off; break;
      case 5: v = &L.user_pair; break;
      case 6: v = &L.item_off; break;
      case 7: v = &L.item_pairs; break;
      case 8: v = &L.item_deg; break;
      case 9: v = &L.chunk_off; break;
      case 10: case 11: case 12: case 13: case 14: case 15: break;
      default: throw std::invalid_argument("unknown layout array");
    }
    if (which >= 10) {  // arrays of 4-int records
      const void *src = nullptr;
      size_t n = 0;
      switch (which) {
        case 10: src = L.chunks.data(); n = L.chunks.size(); break;
        case 11: src = L.mv_chunks.data(); n = L.mv_chunks.size(); break;
        case 12: src = L.pair_work.items.data(); n = L.pair_work.items.size(); break;
        case 13: src = L.user_work.items.data(); n = L.user_work.items.size(); break;
        case 14: src = L.pair_work.splits.data(); n = L.pair_work.splits.size(); break;
        default: src = L.user_work.splits.data(); n = L.user_work.splits.size(); break;
      }
      *count = static_cast<int64_t>(n) * 4;
      if (out && n) {
        if (capacity < *count) throw ApiError(MMSBM_E_TOOLARGE, "buffer too small");
        std::memcpy(out, src, sizeof(int32_t) * *count);
      }
      return;
    }
    *count = static_cast<int64_t>(v->size());
    if (out) {
      if (capacity < *count) throw ApiError(MMSBM_E_TOOLARGE, "buffer too small");
      std::memcpy(out, v->data(), sizeof(int32_t) * v->size());
    }
  });
}

}  // extern "C"

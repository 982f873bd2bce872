// pair_mfma.hpp -- the pair stage on the matrix cores: pair_mfma_kernel (K, L <= 64), mfma_rows_kernel + mfma_slab_kernel (blocked)
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
#pragma once

namespace {

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

// NT threads: 256 (four waves as described) or 512 -- eight waves, each with half of the column tiles of its
// T rows and <= 2 slab tiles, so that the accumulators and the prefetched rows of the T+S launch fit
// 128 registers and two workgroups (16 waves) share a CU.
// BLK (the A launch): an output remainder of 4 or 8 groups as 4 x 4 block instructions (the launch's geometry may still
// hold none).
template <bool GATHER, bool DO_S, int NT, bool BLK>
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
  // ---- geometry (uniform).  A side of 16 f + 4 r groups: f full 16-tiles; a remainder of 4 or 8 (r = 1, 2) runs as
  // 4 x 4 blocks on v_mfma_f64_4x4x4_4b_f64 -- four independent 4 x 4 x 4 products per instruction at the rate of the
  // 16 x 16 x 4 form (16-18 cycles against 64: microbench/mfma4_micro.hip), nothing of it padding; a remainder of 12
  // stays a padded tile.  K = L = 50 (52 = 3 * 16 + 4): 9 tiles + 7 block instructions per step of S instead of 16
  // tiles, 3 tiles + 1 per row tile of T instead of 4.  Lane map of the 4 x 4 x 4 form (probed, same file): lane =
  // 16 kk + 4 blk + i holds A_blk[i][kk], lane = 16 kk + 4 blk + j holds B_blk[kk][j], lane = 16 i + 4 blk + j gets
  // D_blk[i][j] -- the A operand of a row tile's four 4-row blocks is the register the 16 x 16 x 4 form takes.
  const unsigned mg0 = static_cast<unsigned>(pa.mg0), mg1 = static_cast<unsigned>(pa.mg1);
#define MG_RO static_cast<int>(mg0 & 3u)
#define MG_NTO static_cast<int>((mg0 >> 4) & 7u)
#define MG_MTI static_cast<int>((mg0 >> 7) & 7u)
#define MG_COLS_T static_cast<int>((mg1 >> 7) & 127u)
  // (without the blocks: the plain padded geometry, worked out here as before round 4 -- the kernel's registers are counted)
  const int nto = BLK ? MG_NTO : (doutp + 15) >> 4, mti = BLK ? MG_MTI : (dinp + 15) >> 4, ro = BLK ? MG_RO : 0;
  // T: wave = 16 rows x its share of the column work.  Eight waves: the two halves split the tiles, the second half
  // takes the blocks (it is the lighter one: it also gets more of S)
  const int trow0 = 16 * (wave & 3);
  const int half = NW == 8 ? (wave >> 2) : 0, nfh = NW == 8 ? (nto + 1) >> 1 : nto;
  // this wave's column tiles [tlo, tlo + tcnt) (without the blocks: TC per half, as before)
  const int tlo = BLK ? (half == 0 ? 0 : nfh) : TC * half, tcnt = BLK ? (half == 0 ? nfh : nto - nfh) : max(min(nto - TC * half, TC), 0);
  const int tq = (NW == 4 || half == 1) ? ro : 0;                               // ... and column blocks (tcnt < TC then)
  int bcol[TC];  // this lane's column of each of its output tiles
#pragma unroll
  for (int n = 0; n < TC; ++n) bcol[n] = min(16 * (tlo + n) + li, doutp - 1);
  const int bq = MG_COLS_T + (lane & 3);                    // + 4 c: this lane's column of block instruction c
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
    // (the products at a raised wave priority: a wave that has its operands should not queue behind the other
    // workgroup's loads and stores -- C5's T + S launch 356 -> 351 us, variants side by side on one box, round 4)
    __builtin_amdgcn_s_setprio(1);
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
    mfma_d4 acc_t[TC];  // (the last one: the block accumulators of a wave with column blocks)
#pragma unroll
    for (int n = 0; n < TC; ++n) acc_t[n] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    if (tcnt > 0 && tq == 0) {
      // (no unroll pragma: `#pragma unroll 2` on this run-time trip count was never honoured -- -Wpass-failed --
      // and two steps per trip written out by hand are SLOWER: C5's T+S launch 366 -> 434 us, round 3)
      for (int s = 0; s < dinp / 4; ++s) {
        const double x = cst[(4 * s + lk) * CS + trow0 + li];
        const double *trow = tile_l + (4 * s + lk) * doutp;
#pragma unroll
        for (int n = 0; n < TC; ++n)  // (a column tile beyond Dout repeats the last column and is never stored)
          acc_t[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, trow[bcol[n]], acc_t[n], 0, 0, 0);
      }
    } else if (BLK && tq > 0) {  // this wave's tiles (one fewer accumulator than a tile-only wave may hold) and its column blocks
      for (int s = 0; s < dinp / 4; ++s) {
        const double x = cst[(4 * s + lk) * CS + trow0 + li];
        const double *trow = tile_l + (4 * s + lk) * doutp;
#pragma unroll
        for (int n = 0; n < TC - 1; ++n)
          acc_t[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, trow[bcol[n]], acc_t[n], 0, 0, 0);
        acc_t[TC - 1][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, trow[bq], acc_t[TC - 1][0], 0, 0, 0);
        if (tq > 1) acc_t[TC - 1][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, trow[bq + 4], acc_t[TC - 1][1], 0, 0, 0);
      }
    }
    STAMP(6);
    __builtin_amdgcn_s_setprio(0);
    if (BLK && tq > 0) {  // block results: lane = 16 i + 4 blk + j holds row 4 blk + i, column j of its block
      const int r = trow0 + 4 * ((lane >> 2) & 3) + lk;
      if (r < np) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (c < tq) {
            double *dst = pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + r), bq + 4 * c);
            if (GATHER && (pa.nt & 2)) __builtin_nontemporal_store(acc_t[TC - 1][c], dst);
            else *dst = acc_t[TC - 1][c];
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < TC; ++n) {
      const int col = 16 * (tlo + n) + li;
      if (n < tcnt && col < doutp) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r = trow0 + lk + 4 * g;
          if (r < np) {
            double *dst = pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + r), col);
            if (GATHER && (pa.nt & 2)) __builtin_nontemporal_store(acc_t[n][g], dst);  // (A rows: stages.hpp, nt_on)
            else *dst = acc_t[n][g];
          }
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
#undef MG_RO
#undef MG_NTO
#undef MG_MTI
#undef MG_COLS_T
// ======================================================================================
// The same two products for K or L beyond 64, in 64 x 64 blocks (round 2).  Before, these shapes ran the
// lane-per-pair stage with the tile through scalar loads (K, L up to ~150) or the plain wide-row kernels
// (beyond): at K = L = 100 the pair stage took 61 % of the iteration, at K = L = 200 75 %.  Blocked, T and
// S no longer share a workgroup (T sums over ALL of Din for a block of outputs; S keeps a Din x Dout block
// in the accumulators over ALL pairs of a chunk), so the T+S stage is two launches:
//   mfma_rows_kernel<GATHER>  workgroup = (two 64-pair units, block of <= 64 output columns): loops over the
//       64-blocks of Din -- the tile block once, then each unit's X block transposed, in LDS; the next step's
//       blocks in flight -- with the units' output tiles in the accumulators throughout (wave = 16 rows x 2
//       column tiles per unit);
//   mfma_slab_kernel          workgroup = (chunk, Din block, Dout block): loops over the chunk's units --
//       X block transposed + E block in LDS -- with its <= 16 slab tiles dealt to the 8 waves.
// Operand layouts, clamping of partial tiles and association order as in pair_mfma_kernel.  Each table is
// re-read once per block of the other side (from L2 / the Infinity Cache: blocks of one unit are
// neighbours in the grid).
// ======================================================================================

// pair = t / w for t < 2^32 / w through a multiply-high (see pair_mfma_kernel)
__device__ __forceinline__ unsigned mfma_magic(int w) { return 0xFFFFFFFFu / static_cast<unsigned>(w) + 1u; }


template <bool GATHER>
__global__ __launch_bounds__(kPairBlockMax, 4) void mfma_rows_kernel(PairBlockArgs pa, const double *__restrict__ tiles0,
                                                                    int groups_per_chunk, int n_lb) {
  constexpr int NT = kPairBlockMax, CS = kUnitPairs + 1, NLD = kUnitPairs * kMfmaBlk / 2 / NT;  // 4 double2 per table
  constexpr int UPW = kRowsUnits;
  const size_t slot = blockIdx.y;
  const double *__restrict__ tiles = tiles0 + slot * pa.bs_tiles;
  const double *__restrict__ in_tab = pa.in_tab + slot * pa.bs_in;
  double *__restrict__ out = pa.out + slot * pa.bs_out;
  double *__restrict__ out_tail = pa.out_tail + slot * pa.bs_out_t;
  const int dinp = pa.dinp, doutp = pa.doutp;
  const int lb = static_cast<int>(blockIdx.x % n_lb), ug = static_cast<int>(blockIdx.x / n_lb);
  const mmsbm::Chunk ch = pa.chunks[ug / groups_per_chunk];
  const int g0 = ch.q_begin + (ug % groups_per_chunk) * (UPW * kUnitPairs);  // first pair of this group of units
  if (g0 >= ch.q_end) return;
  const int gpairs = min(UPW * kUnitPairs, ch.q_end - g0);
  const int n_units = (gpairs + kUnitPairs - 1) / kUnitPairs;
  const int lb0 = lb * kMfmaBlk, lbw = min(kMfmaBlk, doutp - lb0);
  extern __shared__ double lds[];
  double *cst = lds;                        // [64 k'][CS]   X block of one unit, transposed
  double *tile_b = cst + kMfmaBlk * CS;     // [64 k'][64]   tile block
  int *ids_l = reinterpret_cast<int *>(tile_b + kMfmaBlk * kMfmaBlk);  // [UPW * 64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  if (GATHER) {
    if (tid < UPW * kUnitPairs) ids_l[tid] = pa.pair_item[g0 + min(tid, gpairs - 1)];
    __syncthreads();
  }
  const double *__restrict__ tile_r = tiles + static_cast<size_t>(ch.rating) * dinp * doutp + lb0;
  const int trow0 = 16 * (wave & 3), tn0 = 2 * (wave >> 2);
  int bcol[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) bcol[n] = min(16 * (tn0 + n) + li, lbw - 1);
  mfma_d4 acc[UPW][2];
#pragma unroll
  for (int u = 0; u < UPW; ++u)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[u][n] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  const unsigned ml = mfma_magic(lbw);
  double2 vx[NLD];
  double tx[NLD], ty[NLD];
  // Steps: for every 64-block of Din, the tile block once and then the X block of each of the group's units.
  // shares of a block: X element t = (pair, k') with pair = t / kbw; tile element t = (k', j) with k' = t / lbw
#define ROWS_FETCH_X(KB0, U)                                                                                   \
  do {                                                                                                         \
    const int fk0 = (KB0), fkw = min(kMfmaBlk, dinp - fk0), fu0 = (U) * kUnitPairs;                            \
    const int fnp = min(kUnitPairs, gpairs - fu0);                                                             \
    const unsigned fmx = mfma_magic(fkw);                                                                      \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                          \
      const int t = tid * 2 + j * NT * 2;                                                                      \
      const int fp0 = static_cast<int>(__umulhi(static_cast<unsigned>(t), fmx)), fpr = fu0 + min(fp0, fnp - 1); \
      const size_t frow = GATHER ? static_cast<size_t>(ids_l[fpr]) : static_cast<size_t>(g0 + fpr);            \
      vx[j] = *reinterpret_cast<const double2 *>(in_tab + frow * dinp + fk0 + min(t - fp0 * fkw, fkw - 2));    \
    }                                                                                                          \
  } while (0)
#define ROWS_FETCH_T(KB0)                                                                                      \
  do {                                                                                                         \
    const int fk0 = (KB0), fkw = min(kMfmaBlk, dinp - fk0);                                                    \
    _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                          \
      const int t = tid * 2 + j * NT * 2;                                                                      \
      const int fr0 = static_cast<int>(__umulhi(static_cast<unsigned>(t), ml)), frr = min(fr0, fkw - 1);       \
      const double2 ft = *reinterpret_cast<const double2 *>(tile_r + static_cast<size_t>(fk0 + frr) * doutp +  \
                                                            min(t - fr0 * lbw, lbw - 2));                      \
      tx[j] = ft.x;                                                                                            \
      ty[j] = ft.y;                                                                                            \
    }                                                                                                          \
  } while (0)
  ROWS_FETCH_T(0);
  ROWS_FETCH_X(0, 0);
  for (int kb0 = 0; kb0 < dinp; kb0 += kMfmaBlk) {
    const int kbw = min(kMfmaBlk, dinp - kb0);
    const unsigned mx = mfma_magic(kbw);
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
      if (u < n_units) {
        if (kb0 != 0 || u != 0) __syncthreads();  // previous step fully consumed
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
          const int t = tid * 2 + j * NT * 2;
          const int pr = static_cast<int>(__umulhi(static_cast<unsigned>(t), mx));
          if (pr < kUnitPairs) {  // (pairs beyond the unit hold a copy of its last row: their outputs are never stored)
            double *dst = cst + (t - pr * kbw) * CS + pr;
            dst[0] = vx[j].x;
            dst[CS] = vx[j].y;
          }
          if (u == 0) {
            const int kr = static_cast<int>(__umulhi(static_cast<unsigned>(t), ml));
            if (kr < kbw) {
              double2 t2;
              t2.x = tx[j]; t2.y = ty[j];
              *reinterpret_cast<double2 *>(tile_b + kr * kMfmaBlk + (t - kr * lbw)) = t2;
            }
          }
        }
        __syncthreads();
        // the next step's blocks travel during the products
        if (u + 1 < n_units) {
          ROWS_FETCH_X(kb0, u + 1);
        } else if (kb0 + kMfmaBlk < dinp) {
          ROWS_FETCH_T(kb0 + kMfmaBlk);
          ROWS_FETCH_X(kb0 + kMfmaBlk, 0);
        }
        for (int s = 0; s < kbw / 4; ++s) {  // (no unroll pragma: see pair_mfma_kernel)
          const double x = cst[(4 * s + lk) * CS + trow0 + li];
          const double *trow = tile_b + (4 * s + lk) * kMfmaBlk;
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[u][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, trow[bcol[n]], acc[u][n], 0, 0, 0);
        }
      }
    }
  }
#undef ROWS_FETCH_X
#undef ROWS_FETCH_T
#pragma unroll
  for (int u = 0; u < UPW; ++u) {
    if (u < n_units) {
      const int q0 = g0 + u * kUnitPairs, np = min(kUnitPairs, gpairs - u * kUnitPairs);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int col = 16 * (tn0 + n) + li;
        if (col < lbw) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int r = trow0 + lk + 4 * g;
            if (r < np) *pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + r), lb0 + col) = acc[u][n][g];
          }
        }
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

}  // namespace

// pair_block.hpp -- the pair stage on the vector ALUs: pair_block_kernel (T + S launch, A launch)
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
#pragma once

namespace {

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
struct PairBlockArgs {
  const double *tiles; const double *in_tab; const double *e_tab; const int32_t *pair_item;
  const mmsbm::Chunk *chunks; double *out; double *partial;
  int din, dinp, doutp, spb, nsub, nt;  // nt: output rows as non-temporal stores (bit 0: here, bit 1: pair_mfma_kernel's A rows)
#ifdef MMSBM_ABLATE
  int abl;  // diagnostic build only: phases to skip (mmsbm_hip_time_stage, stage >> 8)
#endif
  // output rows: `out` is a plain [rows][doutp] table (T: out_mw == doutp, out_rs == doutp) or the
  // main part of a RowTab whose tail part starts at out_tail (A)
  int out_mw, out_rs_m, out_rs_t;
  double *out_tail;
  size_t bs_tiles, bs_in, bs_e, bs_out, bs_out_t, bs_partial;  // restart slots (blockIdx.y): offsets
  int mg0, mg1;  // pair_mfma_kernel: tiles and 4 x 4 blocks of this launch's product (pair_big.hpp: mfma_geometry)
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
  const int dinp = pa.dinp, doutp = pa.doutp, spb = pa.spb;  // (rows >= din are zero)
  const int nsub = pa.nsub;
  // abl: a compile-time 0 in the product.  The diagnostic build (-DMMSBM_ABLATE) takes it from the arguments -- bit0
  // rows, bit1 eta rows, bit2 S, bit3 mat-vec, bit4 output copy, bit5 slab store are skipped when set
#ifdef MMSBM_ABLATE
  const int abl = pa.abl;
#else
  constexpr int abl = 0;
#endif
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
          store_out2(dst + t, *reinterpret_cast<const double2 *>(tout + t), (pa.nt & 1) != 0);
      } else {  // RowTab output (A): row by row, main part and tail part
        for (int t = tid * 2; t < total; t += nthr * 2) {
          const int pr = t / doutp, j = t - pr * doutp;
          store_out2(pair_out_ptr(pa, out, out_tail, static_cast<size_t>(q0 + pr), j), *reinterpret_cast<const double2 *>(tout + t), (pa.nt & 1) != 0);
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

// ---- host side: the argument blocks of these kernels from the context ----
PairBlockArgs pair_block_t_args(const mmsbm_hip_ctx *c) {
  const int s = c->base_slot;
  PairBlockArgs pa{};
  pa.tiles = c->p[c->cur].at(s); pa.in_tab = c->ctab.at(s); pa.e_tab = c->eta[c->cur].at(s);
  pa.pair_item = c->pair_item.ptr; pa.chunks = c->mv_chunks.ptr;
  pa.out = c->ttab.at(s); pa.partial = c->partial.at(s);
  pa.din = c->k; pa.dinp = c->kp; pa.doutp = c->lp; pa.spb = c->pb_spb; pa.nsub = c->pb_nsub;
#ifdef MMSBM_ABLATE
  pa.abl = c->ablate;
#endif
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
  pa.din = c->l; pa.dinp = c->lp; pa.doutp = c->kp; pa.spb = kBlock; pa.nsub = 1;
#ifdef MMSBM_ABLATE
  pa.abl = c->ablate;
#endif
  pa.nt = (nt_on(c) & 1) | (((c->nt_out & 1) && c->launch_slots == 1 && c->lay.pair_work.items.empty() && c->lay.user_work.items.empty()) ? 2 : 0);  // bit 1: the matrix-core A rows (C5 2,236 -> 2,215 us; T rows there: nothing)
  pa.out_mw = at.mw; pa.out_rs_m = at.rs_m; pa.out_rs_t = at.rs_t; pa.out_tail = at.tail;
  pa.bs_tiles = c->pt[0].stride; pa.bs_in = c->eta[0].stride; pa.bs_e = 0;
  pa.bs_out = at.so_m; pa.bs_out_t = at.so_t; pa.bs_partial = 0;
  mfma_geometry(pa.dinp, pa.doutp, &pa.mg0, &pa.mg1);
  return pa;
}
// The A launch's arguments and workgroup count: A[q,:] from (eta, pT) of parameter buffer `slot` into atab[a_slot] -- or,
// with `grid` set, the same mat-vec over every (item, rating) combination into the plain table btab (prod_dist / predict)
PairBlockArgs matvec_a_args(const mmsbm_hip_ctx *c, int slot, int a_slot, bool grid, int *n_blocks) {
  int nb = grid ? c->grid_n_chunks : static_cast<int>(c->lay.mv_chunks.size());
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
  *n_blocks = nb;
  return pa;
}

}  // namespace

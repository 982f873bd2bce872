// pair_quad.hpp -- the pair stage for big rating tiles on the vector ALUs: pair_quad_a_kernel (long rows, the tile in LDS)
// and the wide-row kernels (wide_matvec_kernel, wide_slab_kernel: any K, L)
// Included by the translation units that launch these kernels (see prelude.hpp for the order); not a stand-alone header.
#pragma once

namespace {

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

}  // namespace

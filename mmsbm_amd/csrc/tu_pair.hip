// tu_pair.hip -- translation unit of the pair stage on the vector ALUs: pair_block_kernel (T + S launch, A launch;
// pair_block.hpp), pair_quad_a_kernel and the wide-row kernels (pair_quad.hpp), with the host code that chooses among them
#include "prelude.hpp"
#include "pair_block.hpp"
#include "pair_quad.hpp"

namespace mmsbm_hip_impl {

// pair_block_kernel<GATHER, DO_S, NACC, TLDS, NT, KT, DIRECT>: create() ties three of the switches to ONE fact about the
// shape -- the padded tile holds more than 1,024 entries (8 KB, the scalar cache's share):
//   small tile: KT = 2, rows through LDS (DIRECT off), tile in scalar registers (TLDS off), NACC = 1
//   big tile:   KT = 4, rows straight from registers (DIRECT on), tile in LDS unless it does not fit beside the rows
// so only those combinations are instantiated.
void stage_dense_valu(mmsbm_hip_ctx *c) {  // T = P^T C  and the K x L slabs for p
  const int nb = static_cast<int>(c->lay.mv_chunks.size());
  const PairBlockArgs pa = pair_block_t_args(c);
  if (c->wide) {
    LaunchScope ls(c, K_DENSE);
    const int subs = kWideChunkPairs / kWidePairs;
    const int kgs = (c->kp + kWideKG - 1) / kWideKG, lbs = (c->lp + kBlock - 1) / kBlock;
    const size_t lds = static_cast<size_t>(kWidePairs) * c->kp * sizeof(double);
    allow_big_lds(wide_matvec_kernel<false>, lds);
    LAUNCH((wide_matvec_kernel<false>), slot_grid(c, nb * subs), kBlock, lds, c->stream, pa, subs);
    LAUNCH(wide_slab_kernel, slot_grid(c, nb * kgs * lbs), kBlock, 0, c->stream, pa, kgs, lbs);
    ls.done();
    return;
  }
  LaunchScope ls(c, K_DENSE, true);
  const bool big_tile = c->direct_out, big_wg = c->pb_threads_t > kBlock;
  if ((c->pb_kt == 4) != big_tile || (c->tl_t && !big_tile) || (!big_tile && c->pb_nacc != 1))
    throw ApiError(MMSBM_E_INTERNAL, "pair_block (T + S): inconsistent launch shape");
#define PB_GO(N, TL, NT, KT, D)                                                             \
  do {                                                                                      \
    allow_big_lds(pair_block_kernel<false, true, N, TL, NT, KT, D>, c->lds_t);              \
    LAUNCH_IN(ls, (pair_block_kernel<false, true, N, TL, NT, KT, D>), slot_grid(c, nb), NT, c->lds_t, c->stream, pa, pa.tiles); \
  } while (0)
#define PB_BIG(N)                                                                           \
  do {                                                                                      \
    if (c->tl_t && big_wg) PB_GO(N, true, kPairBlockMax, 4, true);                          \
    else if (c->tl_t) PB_GO(N, true, kBlock, 4, true);                                      \
    else if (big_wg) PB_GO(N, false, kPairBlockMax, 4, true);                               \
    else PB_GO(N, false, kBlock, 4, true);                                                  \
  } while (0)
  if (!big_tile) {
    if (big_wg) PB_GO(1, false, kPairBlockMax, 2, false); else PB_GO(1, false, kBlock, 2, false);
  } else {
    // (four slots per thread: only where 512 threads are too few for the tile's slots, (K/4)(L/4) > 1,024 -- such a tile
    // does not fit the LDS beside the rows, and a 256-thread launch (L <= 24) runs out of LDS for its K rows first)
    switch (c->pb_nacc) {
      case 1: PB_BIG(1); break;
      case 2: PB_BIG(2); break;
      default:
        if (c->tl_t || !big_wg) throw ApiError(MMSBM_E_INTERNAL, "pair_block (T + S): four slots per thread with the tile in LDS or 256 threads");
        PB_GO(4, false, kPairBlockMax, 4, true);
        break;
    }
  }
#undef PB_BIG
#undef PB_GO
  ls.done();
}

void stage_matvec_a_valu(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid) {
  int nb = 0;
  const PairBlockArgs pa = matvec_a_args(c, slot, a_slot, grid, &nb);
  if (nb == 0) return;
  LaunchScope ls(c, K_MATVEC_A, true);
  if (c->wide) {
    const int subs = kWideChunkPairs / kWidePairs;
    const size_t lds = static_cast<size_t>(kWidePairs) * c->lp * sizeof(double);
    allow_big_lds(wide_matvec_kernel<true>, lds);
    LAUNCH_IN(ls, (wide_matvec_kernel<true>), slot_grid(c, nb * subs), kBlock, lds, c->stream, pa, subs);
  } else if (c->quad_a) {
    const dim3 grid_q = slot_grid(c, std::min(nb, c->n_cus));
#define QA(NL)                                                                                    \
  do {                                                                                            \
    allow_big_lds(pair_quad_a_kernel<NL>, c->lds_qa);                                             \
    LAUNCH_IN(ls, (pair_quad_a_kernel<NL>), grid_q, kPairBlockMax, c->lds_qa, c->stream, pa, pa.tiles, nb); \
  } while (0)
    const int nl = (c->lp + 3) / 4;  // dinp of the A launch = lp
    if (nl <= 8) QA(8); else if (nl <= 10) QA(10); else if (nl <= 12) QA(12);
    else if (nl <= 13) QA(13); else QA(14);   // (kQuadMaxL: 15 or 16 double2 per thread spill at 256 registers)
#undef QA
  } else {
    const bool big_tile = c->direct_out, big_wg = c->pb_threads_a > kBlock;
    if (c->tl_a && !big_tile) throw ApiError(MMSBM_E_INTERNAL, "pair_block (A): inconsistent launch shape");
#define PA_GO(TL, NT, D)                                                                    \
  do {                                                                                      \
    allow_big_lds(pair_block_kernel<true, false, 1, TL, NT, 4, D>, c->lds_a);               \
    LAUNCH_IN(ls, (pair_block_kernel<true, false, 1, TL, NT, 4, D>), slot_grid(c, nb), NT, c->lds_a, c->stream, pa, pa.tiles); \
  } while (0)
    if (!big_tile) { if (big_wg) PA_GO(false, kPairBlockMax, false); else PA_GO(false, kBlock, false); }
    else if (c->tl_a) { if (big_wg) PA_GO(true, kPairBlockMax, true); else PA_GO(true, kBlock, true); }
    else { if (big_wg) PA_GO(false, kPairBlockMax, true); else PA_GO(false, kBlock, true); }
#undef PA_GO
  }
  ls.done();
}

}  // namespace mmsbm_hip_impl

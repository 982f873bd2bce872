// tu_mfma.hip -- translation unit of the pair stage on the matrix cores (pair_mfma.hpp): pair_mfma_kernel for tiles of
// K, L <= 64 beyond the scalar cache, mfma_rows_kernel + mfma_slab_kernel (blocked) beyond that and for skinny tiles
#include "prelude.hpp"
#include "pair_block.hpp"
#include "pair_mfma.hpp"

namespace mmsbm_hip_impl {

void stage_dense_mfma(mmsbm_hip_ctx *c) {  // T = P^T C  and the K x L slabs for p
  const int nb = static_cast<int>(c->lay.mv_chunks.size());
  const PairBlockArgs pa = pair_block_t_args(c);
  if (c->mfma_big) {
    LaunchScope ls(c, K_DENSE);
    const int subs = (c->mv_chunk_pairs + kRowsUnits * kUnitPairs - 1) / (kRowsUnits * kUnitPairs);  // groups of units per chunk
    const int n_kb = (c->kp + kMfmaBlk - 1) / kMfmaBlk, n_lb = (c->lp + kMfmaBlk - 1) / kMfmaBlk;
    allow_big_lds(mfma_rows_kernel<false>, kMfmaRowsLds);
    allow_big_lds(mfma_slab_kernel, kMfmaSlabLds);
    LAUNCH((mfma_rows_kernel<false>), slot_grid(c, nb * subs * n_lb), kPairBlockMax, kMfmaRowsLds, c->stream, pa, pa.tiles, subs, n_lb);
    LAUNCH(mfma_slab_kernel, slot_grid(c, nb * n_kb * n_lb), kPairBlockMax, kMfmaSlabLds, c->stream, pa, n_kb, n_lb);
    ls.done();
    return;
  }
  LaunchScope ls(c, K_DENSE, true);
  // (padded 16-tiles, eight waves: the 4 x 4 blocks of the A launch do not fit this launch's 128 registers -- measured,
  // EXPERIMENTS.md -- and four waves leave one workgroup per CU)
  allow_big_lds(pair_mfma_kernel<false, true, kPairBlockMax, false>, c->lds_mt);
  LAUNCH_IN(ls, (pair_mfma_kernel<false, true, kPairBlockMax, false>), slot_grid(c, nb), kPairBlockMax, c->lds_mt, c->stream, pa, pa.tiles);
  ls.done();
}

void stage_matvec_a_mfma(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid) {
  int nb = 0;
  const PairBlockArgs pa = matvec_a_args(c, slot, a_slot, grid, &nb);
  if (nb == 0) return;
  LaunchScope ls(c, K_MATVEC_A, true);
  if (c->mfma_big) {
    const int subs = (c->mv_chunk_pairs + kRowsUnits * kUnitPairs - 1) / (kRowsUnits * kUnitPairs);
    const int n_kb = (c->kp + kMfmaBlk - 1) / kMfmaBlk;  // (outputs: K columns)
    allow_big_lds(mfma_rows_kernel<true>, kMfmaRowsLds);
    LAUNCH_IN(ls, (mfma_rows_kernel<true>), slot_grid(c, nb * subs * n_kb), kPairBlockMax, kMfmaRowsLds, c->stream, pa, pa.tiles, subs, n_kb);
  } else {
    // (eight waves like the T + S launch: 84 registers against the four-wave form's 148, so that two to three
    // workgroups = 16 to 24 waves share a CU instead of 12 -- C5 150 -> 139-146 us, 4M ratings at K = L = 50 unchanged)
    allow_big_lds(pair_mfma_kernel<true, false, kPairBlockMax, true>, c->lds_ma);
    LAUNCH_IN(ls, (pair_mfma_kernel<true, false, kPairBlockMax, true>), slot_grid(c, nb), kPairBlockMax, c->lds_ma, c->stream, pa, pa.tiles);
  }
  ls.done();
}

// workgroups of the matrix-core A launch a CU holds (create() sizes that launch's runs of units from it)
int mfma_a_blocks_per_cu(const mmsbm_hip_ctx *c) {
  int per_cu = 0;
  allow_big_lds(pair_mfma_kernel<true, false, kPairBlockMax, true>, c->lds_ma);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pair_mfma_kernel<true, false, kPairBlockMax, true>, kPairBlockMax, c->lds_ma) != hipSuccess || per_cu < 1) per_cu = 1;
  return per_cu;
}

}  // namespace mmsbm_hip_impl

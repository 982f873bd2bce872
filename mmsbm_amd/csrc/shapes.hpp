// shapes.hpp -- sizes and limits every translation unit of the library agrees on: workgroup sizes, unit lengths and the
// LDS footprints the kernels are launched with (host constants only; no kernels)
#pragma once

namespace {

constexpr int kUnitPairs = 64;       // pairs of one rating a pair-stage workgroup multiplies at a time
constexpr int kPairBlockMax = 512;   // threads of the pair stage's big workgroups

constexpr int kQuadUnits = 4;  // 64-pair units a workgroup of pair_quad_a_kernel multiplies jointly
constexpr int kQuadMaxL = 56;  // its input rows at most (14 double2 per thread and chunk; beyond: pair_block like every other shape)

constexpr size_t kLdsBudget = 64 * 1024;  // dynamic LDS a launch may use without hipFuncSetAttribute
constexpr size_t kLdsMax = 160 * 1024;    // with hipFuncAttributeMaxDynamicSharedMemorySize

// dynamic LDS of pair_block: transposed rows + output rows (shared with the eta rows)
constexpr size_t kScalarTileBytes = 8 * 1024;  // larger tiles thrash the scalar cache: stage in LDS
inline bool tile_in_lds(int dinp, int doutp) {
  return static_cast<size_t>(dinp) * doutp * sizeof(double) > kScalarTileBytes;
}
inline size_t pair_block_lds(int dinp, int doutp, bool tile_lds) {
  // transposed input rows + one region shared by the eta rows and the output rows [+ the tile]
  const size_t d = static_cast<size_t>(dinp) * (kUnitPairs + 1) + static_cast<size_t>(kUnitPairs) * doutp +
                   (tile_lds ? static_cast<size_t>(dinp) * doutp : 0);
  return d * sizeof(double);  // the S hand-over area reuses it (create() bounds the copies by it)
}

// ---- the pair stage on the matrix cores (pair_mfma.hpp) ----
constexpr int kMfmaMaxDim = 64;       // <= 4 tiles of 16 per side
constexpr int kMfmaChunkPairs = 1024;  // pairs per workgroup at most (their item ids are parked in LDS)
static_assert(kMfmaChunkPairs >= 4 * mmsbm::kMvChunkPairs, "pair_mfma_kernel parks a whole chunk's item ids in LDS");
constexpr int kMfmaBlk = 64;          // block of the blocked forms (mfma_rows_kernel, mfma_slab_kernel)
constexpr int kRowsUnits = 2;  // 64-pair units per workgroup of mfma_rows_kernel: one staged tile block serves them all
constexpr size_t kMfmaRowsLds = (kMfmaBlk * (kUnitPairs + 1) + kMfmaBlk * kMfmaBlk) * sizeof(double) + kRowsUnits * kUnitPairs * sizeof(int);
constexpr size_t kMfmaSlabLds = (kMfmaBlk * (kUnitPairs + 1) + kUnitPairs * kMfmaBlk) * sizeof(double) + kMfmaChunkPairs * sizeof(int);
inline size_t pair_mfma_lds(int dinp, int doutp, bool with_s) {
  return (static_cast<size_t>(dinp) * (kUnitPairs + 1) + static_cast<size_t>(dinp) * doutp +
          (with_s ? static_cast<size_t>(kUnitPairs) * doutp : 0)) * sizeof(double) + kMfmaChunkPairs * sizeof(int);
}

// The 4 x 4 blocks of the A launch (the T + S launch keeps padded tiles: its 128 registers do not hold them; measured,
// EXPERIMENTS.md).  An output side of 16 f + 4 r groups: f full 16-tiles; a remainder of 4 or 8 (r = 1, 2) runs as 4 x 4
// blocks on v_mfma_f64_4x4x4_4b_f64, a remainder of 12 stays a padded tile.  Packed for PairBlockArgs::mg0 / mg1.
inline void mfma_geometry(int dinp, int doutp, int *mg0, int *mg1) {
  const int fo = doutp >> 4, ro0 = (doutp & 15) >> 2;
  const int ro = (ro0 == 1 || ro0 == 2) ? ro0 : 0;
  const int nto = fo + ((ro0 && !ro) ? 1 : 0), mti = (dinp + 15) >> 4;      // (padded) 16-tiles per side
  const int rows_t = std::min(16 * mti, dinp), cols_t = std::min(16 * nto, doutp);   // rows / columns the tiles cover
  *mg0 = ro | (nto << 4) | (mti << 7);
  *mg1 = rows_t | (cols_t << 7);
}

// ---- wide rows (pair_quad.hpp: wide_matvec_kernel, wide_slab_kernel) ----
constexpr int kWidePairs = 8, kWideChunkPairs = 1024;
constexpr int kWideKG = 8;  // (16: the C values no longer fit the scalar registers, 921 vs 477 us)

// ---- small problems: the two-launch iteration (fused_small.hpp) ----
inline size_t pairs_fused_lds(int kp, int lp, int split_parts = 0) {  // split_parts: partial rows of a unit's split pairs
  return (static_cast<size_t>(lp) * (kUnitPairs + 1) + static_cast<size_t>(kUnitPairs) * lp +
          static_cast<size_t>(kUnitPairs) * kp + static_cast<size_t>(kp) * (kUnitPairs + 1) +
          static_cast<size_t>(split_parts) * kp) * sizeof(double);
}
constexpr size_t kFusedSplitLds = 96 * 1024;  // partial rows of a workgroup's split user segments, at most

}  // namespace

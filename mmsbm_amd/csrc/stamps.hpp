// stamps.hpp -- phase stamps of the diagnostic build (-DMMSBM_STAMPS); empty macros in the product
#pragma once

namespace {

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
// where the workgroup runs: HW_ID (wave, SIMD, CU, SH, SE ...) in the low word, XCC_ID in the high one
#define STAMP_WHERE(i)                                                                       \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < kStampBlocks && blockIdx.y == 0)                    \
      g_stamps[blockIdx.x * kStampSlots + (i)] =                                             \
          static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(4 | (31 << 11))) |      \
          (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(20 | (31 << 11))) << 32); \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMP_WHERE(i) do {} while (0)
#endif

}  // namespace

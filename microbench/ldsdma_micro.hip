// ldsdma_micro.hip -- can LDS-DMA (global_load_lds_dwordx4, gfx950) feed the matrix-core pair stage?
// Part A: where do the bytes land?  One wave, 16 bytes per lane, some lanes masked, a wave-uniform LDS base and an
//         instruction offset: prints the LDS chunk every lane's data arrived in.
// Part B: the T + S launch's data movement per 64-pair unit at BASELINE's config 5 (K = L = 50, rows padded to 52
//         doubles = 416 bytes: 64 C rows, contiguous, and 64 gathered eta rows -- 53 KB per unit, 15,616 units) with
//         the kernel's matrix work beside it (58 v_mfma_f64_16x16x4 per wave and unit, operands from LDS):
//           REG  rows -> registers (next unit in flight during the products) -> ds_write -> barrier   (today's form)
//           DMA  rows -> the OTHER LDS buffer by global_load_lds_dwordx4 during the products; no registers, no ds_write
//         at the dynamic-LDS sizes that decide how many workgroups share a CU.  Prints us per launch and TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const void __attribute__((address_space(1))) *)(p))
#define LPTR(p) ((void __attribute__((address_space(3))) *)(p))

__global__ void probe(const double *g, double *out, int mode) {
  __shared__ double lds[512];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = -1.0;
  __syncthreads();
  if (lane % 3 != 0) {
    if (mode == 0) __builtin_amdgcn_global_load_lds(GPTR(g + 2 * (100 + lane)), LPTR(lds + 64), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds(GPTR(g + 2 * (100 + lane)), LPTR(lds + 64), 16, 256, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

constexpr int kW = 52, kRowChunks = kW / 2;   // a row: 26 chunks of 16 bytes
template <bool DMA, int ROWS, int NT>
__global__ __launch_bounds__(NT) void stream(const double *__restrict__ xtab, const double *__restrict__ etab,
                                             const int *__restrict__ ids, double *__restrict__ out, int n_units, int mf,
                                             double *__restrict__ tout) {
  extern __shared__ double lds[];
  constexpr int TAB = ROWS * kW, BUF = 2 * TAB, CH = ROWS * kRowChunks, NLD = (CH + NT - 1) / NT, NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  double vxx[NLD], vxy[NLD], vex[NLD], vey[NLD];   // (scalars: a double2 array that feeds ds_write_b128 stays in scratch memory)
  const int u0 = blockIdx.x, ustep = gridDim.x;
  // (a macro, not a lambda: register arrays captured by reference end up in scratch memory)
#define FETCH_REGS(U)                                                                                      \
  _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                        \
    const int c = min(tid + j * NT, CH - 1), r = c / kRowChunks, cc = c - r * kRowChunks;                 \
    const double2 fx = *reinterpret_cast<const double2 *>(xtab + (size_t(U) * ROWS + r) * kW + 2 * cc);    \
    const double2 fe = *reinterpret_cast<const double2 *>(etab + size_t(ids[size_t(U) * ROWS + r]) * kW + 2 * cc); \
    vxx[j] = fx.x; vxy[j] = fx.y; vex[j] = fe.x; vey[j] = fe.y;                                           \
  }
  auto fetch_dma = [xtab, etab, ids, wave, lane](int u, double *buf) {   // wave-instruction i covers chunks 64 i .. 64 i + 63 of a table
    for (int i = wave; i * 64 < CH; i += NW) {
      const int c = i * 64 + lane;
      if (c < CH) {
        const int r = c / kRowChunks, cc = c - r * kRowChunks;
        __builtin_amdgcn_global_load_lds(GPTR(xtab + (size_t(u) * ROWS + r) * kW + 2 * cc), LPTR(buf + i * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GPTR(etab + size_t(ids[size_t(u) * ROWS + r]) * kW + 2 * cc), LPTR(buf + TAB + i * 128), 16, 0, 0);
      }
    }
  };
  int cur = 0;
  if (u0 < n_units) { if (DMA) fetch_dma(u0, lds); else { FETCH_REGS(u0) } }
  for (int u = u0; u < n_units; u += ustep) {
    double *buf = lds + (DMA ? cur * BUF : 0);
    if (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                     // this unit's rows are in `buf`; the other buffer is free
      if (u + ustep < n_units) fetch_dma(u + ustep, lds + (cur ^ 1) * BUF);
    } else {
      __syncthreads();                                     // the previous unit has been consumed
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        const int c = tid + j * NT;
        if (c < CH) {
          double2 sx, se;
          sx.x = vxx[j]; sx.y = vxy[j]; se.x = vex[j]; se.y = vey[j];
          *reinterpret_cast<double2 *>(buf + 2 * c) = sx;
          *reinterpret_cast<double2 *>(buf + TAB + 2 * c) = se;
        }
      }
      __syncthreads();
      if (u + ustep < n_units) { FETCH_REGS(u + ustep) }
    }
    // the products' share of the matrix pipe and of the LDS pipe: mf instructions, two operands each
    const int li = lane & 15, lk = lane >> 4;
    for (int s = 0; s < mf; s += 2) {
      const int row = (4 * (s >> 1) + lk) % ROWS;
      const double a = buf[row * kW + li + 16 * (wave & 1)], b = buf[TAB + row * kW + li + 16 * ((wave >> 1) & 1)];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
    }
    if (tout) {   // the unit's T rows, from the accumulators: 8-byte stores, 16 lanes = 128 contiguous bytes
      for (int e = tid; e < ROWS * kW; e += NT) tout[size_t(u) * ROWS * kW + e] = acc0[e & 3];
    }
    cur ^= 1;
  }
  out[size_t(blockIdx.x) * NT + tid] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

template <bool DMA, int ROWS, int NT>
void run(const char *name, const double *x, const double *e, const int *ids, double *out, int n_units, bool mfma, size_t lds, int grid,
         double *tout, int mfma_pct = 100) {
  const int mf = mfma ? 464 * ROWS / 64 / (NT / 64) * mfma_pct / 100 / 2 * 2 : 0;   // the T + S launch: 464 matrix instructions per 64 pairs, dealt to the waves
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream<DMA, ROWS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stream<DMA, ROWS, NT>, NT, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) stream<DMA, ROWS, NT><<<grid, NT, lds>>>(x, e, ids, out, n_units, mf, tout);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) stream<DMA, ROWS, NT><<<grid, NT, lds>>>(x, e, ids, out, n_units, mf, tout);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000 / reps, bytes = double(n_units) * ROWS * kW * 8 * (tout ? 3 : 2);
  printf("%-28s %2d rows/unit, %3d threads, %3zu KB LDS (%d per CU), grid %5d, %3d mfma per wave and unit, T rows %s: %7.1f us  %5.2f TB/s\n",
         name, ROWS, NT, lds / 1024, per_cu, grid, mf, tout ? "stored" : "-     ", us, bytes / us / 1e6);
}


// A ring of NB LDS buffers filled by LDS-DMA DIST units ahead, one workgroup per CU: a unit's rows were asked for DIST
// units ago, so the wait in front of a unit is a COUNTED s_waitcnt (everything but the younger DMA and T-store
// instructions) and normally finds them there; the T rows of a unit are stored without anyone waiting for them.
template <int ROWS, int NT, int NB>
__global__ __launch_bounds__(NT) void ring(const double *__restrict__ xtab, const double *__restrict__ etab,
                                           const int *__restrict__ ids, double *__restrict__ out, int n_units, int mf,
                                           double *__restrict__ tout) {
  extern __shared__ double lds[];
  constexpr int TAB = ROWS * kW, BUF = 2 * TAB, CH = ROWS * kRowChunks, NW = NT / 64, DIST = NB - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const int u0 = blockIdx.x, ustep = gridDim.x;
  auto fetch_dma = [xtab, etab, ids, wave, lane](int u, double *buf) {
    for (int i = wave; i * 64 < CH; i += NW) {
      const int c = i * 64 + lane;
      const int r = c / kRowChunks, cc = c - r * kRowChunks;
      __builtin_amdgcn_global_load_lds(GPTR(xtab + (size_t(u) * ROWS + r) * kW + 2 * cc), LPTR(buf + i * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(etab + size_t((size_t(u) * ROWS + r) % 100000) * kW + 2 * cc   /* (the real kernel has its item ids in LDS: no vector-memory load in front of the DMA) */), LPTR(buf + TAB + i * 128), 16, 0, 0);
    }
  };
  // vector-memory instructions this wave issues per unit: D loads (LDS-DMA) and S stores (T rows)
  const int D = 2 * ((CH / 64 - wave + NW - 1) / NW), S = tout ? (ROWS * kW - wave * 64 + NT - 1) / NT : 0;
  const int younger = (DIST - 1) * D + DIST * S;   // issued after the DMA of the unit about to be used
  static_assert(CH % 64 == 0, "whole wave-instructions");
  for (int d = 0; d < DIST; ++d)
    if (u0 + d * ustep < n_units) fetch_dma(u0 + d * ustep, lds + d * BUF);
  int k = 0;
  for (int u = u0; u < n_units; u += ustep, ++k) {
    double *buf = lds + (k % NB) * BUF;
    if (k >= DIST && u + DIST * ustep < n_units) {   // steady state: a counted wait (the immediate must be a constant)
      switch (younger) {
#define WCASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        WCASE(0) WCASE(1) WCASE(2) WCASE(3) WCASE(4) WCASE(5) WCASE(6) WCASE(7) WCASE(8) WCASE(9) WCASE(10) WCASE(11) WCASE(12)
        WCASE(13) WCASE(14) WCASE(15) WCASE(16) WCASE(17) WCASE(18) WCASE(19) WCASE(20) WCASE(21) WCASE(22) WCASE(23) WCASE(24)
#undef WCASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a workgroup's first and last units: the count of younger instructions is another
    }
    // every wave's share of this unit has arrived; the buffer of the unit before is free.  (A bare s_barrier:
    // __syncthreads() carries a fence for which the compiler waits out EVERY vector-memory instruction -- the younger
    // DMA too -- which is exactly the serialisation the ring is there to avoid.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (u + DIST * ustep < n_units) fetch_dma(u + DIST * ustep, lds + ((k + DIST) % NB) * BUF);
    const int li = lane & 15, lk = lane >> 4;
    for (int s = 0; s < mf; s += 2) {
      const int row = (4 * (s >> 1) + lk) % ROWS;
      const double a = buf[row * kW + li + 16 * (wave & 1)], b = buf[TAB + row * kW + li + 16 * ((wave >> 1) & 1)];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
    }
    if (tout) {
      for (int e = tid; e < ROWS * kW; e += NT) tout[size_t(u) * ROWS * kW + e] = acc0[e & 3];
    }
  }
  out[size_t(blockIdx.x) * NT + tid] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

template <int ROWS, int NT, int NB>
void run_ring(const char *name, const double *x, const double *e, const int *ids, double *out, int n_units, int mfma_pct, size_t lds, int grid,
              double *tout) {
  const int mf = 464 * ROWS / 64 / (NT / 64) * mfma_pct / 100 / 2 * 2;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ring<ROWS, NT, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ring<ROWS, NT, NB>, NT, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) ring<ROWS, NT, NB><<<grid, NT, lds>>>(x, e, ids, out, n_units, mf, tout);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) ring<ROWS, NT, NB><<<grid, NT, lds>>>(x, e, ids, out, n_units, mf, tout);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000 / reps, bytes = double(n_units) * ROWS * kW * 8 * (tout ? 3 : 2);
  printf("%-28s %2d rows/unit, %3d threads, %3zu KB LDS (%d per CU), grid %5d, %3d mfma per wave and unit, T rows %s: %7.1f us  %5.2f TB/s\n",
         name, ROWS, NT, lds / 1024, per_cu, grid, mf, tout ? "stored" : "-     ", us, bytes / us / 1e6);
}

int main() {
  // ---- A
  {
    std::vector<double> g(1024); for (int i = 0; i < 1024; ++i) g[i] = i;
    double *dg, *dout; CK(hipMalloc((void **)&dg, 8192)); CK(hipMalloc((void **)&dout, 4096));
    CK(hipMemcpy(dg, g.data(), 8192, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode) {
      probe<<<1, 64>>>(dg, dout, mode);
      std::vector<double> o(512); CK(hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost));
      int ok = 1, seen = 0;
      const int base = 32 + (mode ? 16 : 0);   // chunk index of the LDS base (+ the instruction offset of 256 bytes)
      for (int c = 0; c < 256; ++c) {
        const int lane = c - base;
        const bool expect = lane >= 0 && lane < 64 && lane % 3 != 0;
        const int shift = mode ? 32 : 0;   // (measured: the instruction offset is added to the GLOBAL address as well)
        const double want0 = expect ? 2 * (100 + lane) + shift : -1.0, want1 = expect ? 2 * (100 + lane) + 1 + shift : -1.0;
        if (o[2 * c] != want0 || o[2 * c + 1] != want1) { ok = 0; printf("  chunk %d holds (%g, %g), expected (%g, %g)\n", c, o[2 * c], o[2 * c + 1], want0, want1); }
        if (o[2 * c] >= 0) ++seen;
      }
      printf("layout probe, instruction offset %d (added to the LDS AND the global address): lane l -> LDS base + offset + 16 l, masked lanes leave their chunk alone: %s (%d chunks written)\n",
             mode ? 256 : 0, ok ? "CONFIRMED" : "NOT as assumed", seen);
    }
  }
  // ---- B
  int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int n_pairs = 999424, n_items = 100000, n_units64 = n_pairs / 64;
  double *x, *e, *out; int *ids;
  CK(hipMalloc((void **)&x, size_t(n_pairs) * kW * 8)); CK(hipMalloc((void **)&e, size_t(n_items) * kW * 8));
  CK(hipMalloc((void **)&out, size_t(8192) * 512 * 8)); CK(hipMalloc((void **)&ids, size_t(n_pairs) * 4));
  CK(hipMemset(x, 0, size_t(n_pairs) * kW * 8)); CK(hipMemset(e, 0, size_t(n_items) * kW * 8));
  { std::vector<int> h(n_pairs); for (int q = 0; q < n_pairs; ++q) h[q] = q % n_items; CK(hipMemcpy(ids, h.data(), size_t(n_pairs) * 4, hipMemcpyHostToDevice)); }
  double *tout; CK(hipMalloc((void **)&tout, size_t(n_pairs) * kW * 8));
  const size_t b64 = 2 * 64 * kW * 8, b32 = 2 * 32 * kW * 8, tile = 22 * 1024, idsb = 4096;
  for (int mode = 0; mode < 3; ++mode) {
    const bool mfma = mode > 0;
    double *t = mode == 2 ? tout : nullptr;
    printf("-- %s\n", mode == 0 ? "rows only" : (mode == 1 ? "rows + the matrix work" : "rows + the matrix work + T rows stored"));
    run<false, 64, 512>("registers (today's form)", x, e, ids, out, n_units64, mfma, b64 + tile + idsb, 3904, t);
    run<false, 64, 512>("registers", x, e, ids, out, n_units64, mfma, b64 + tile + idsb, 2 * cus, t);
    run<false, 64, 256>("registers", x, e, ids, out, n_units64, mfma, b64 + tile + idsb, 2 * cus, t);
    run<false, 32, 256>("registers", x, e, ids, out, 2 * n_units64, mfma, b32 + tile + idsb, 3 * cus, t);
    run<false, 32, 256>("registers", x, e, ids, out, 2 * n_units64, mfma, b32 + tile + idsb, 3904, t);
    run<false, 32, 512>("registers", x, e, ids, out, 2 * n_units64, mfma, b32 + tile + idsb, 3 * cus, t);
    if (mfma) {   // ... with the matrix work of unpadded tiles (77 %)
      run<false, 64, 512>("registers, unpadded tiles", x, e, ids, out, n_units64, mfma, b64 + tile + idsb, 2 * cus, t, 77);
      run<false, 32, 256>("registers, unpadded tiles", x, e, ids, out, 2 * n_units64, mfma, b32 + tile + idsb, 3 * cus, t, 77);
      run<false, 32, 256>("registers, unpadded tiles", x, e, ids, out, 2 * n_units64, mfma, b32 + tile + idsb, 3904, t, 77);
      run<false, 32, 512>("registers, unpadded tiles", x, e, ids, out, 2 * n_units64, mfma, b32 + tile + idsb, 3 * cus, t, 77);
    }
    run<true, 64, 512>("LDS-DMA, two buffers", x, e, ids, out, n_units64, mfma, 2 * b64 + tile, cus, t);
    run<true, 32, 512>("LDS-DMA, two buffers", x, e, ids, out, 2 * n_units64, mfma, 2 * b32 + tile, 2 * cus, t);
    run<true, 32, 256>("LDS-DMA, two buffers", x, e, ids, out, 2 * n_units64, mfma, 2 * b32 + tile, 2 * cus, t);
    run<true, 32, 512>("LDS-DMA, two, no tile", x, e, ids, out, 2 * n_units64, mfma, 2 * b32, 3 * cus, t);
    run<true, 32, 256>("LDS-DMA, two, no tile", x, e, ids, out, 2 * n_units64, mfma, 2 * b32, 3 * cus, t);
    // rings: one workgroup per CU, rows asked for two (three) units ahead, counted waits
    run_ring<32, 512, 3>("LDS-DMA ring of 3", x, e, ids, out, 2 * n_units64, mfma ? 100 : 0, 3 * b32 + tile, cus, t);
    run_ring<32, 512, 4>("LDS-DMA ring of 4", x, e, ids, out, 2 * n_units64, mfma ? 100 : 0, 4 * b32 + tile, cus, t);
    run_ring<32, 1024, 4>("LDS-DMA ring of 4", x, e, ids, out, 2 * n_units64, mfma ? 100 : 0, 4 * b32 + tile, cus, t);
    run_ring<64, 512, 2>("LDS-DMA ring of 2 (counted)", x, e, ids, out, n_units64, mfma ? 100 : 0, 2 * b64 + tile, cus, t);
    if (mfma) {   // ... and with the matrix work of unpadded tiles (the 4 x 4 blocks fit 256 registers): 77 %
      run_ring<32, 512, 3>("ring of 3, unpadded tiles", x, e, ids, out, 2 * n_units64, 77, 3 * b32 + tile, cus, t);
      run_ring<32, 512, 4>("ring of 4, unpadded tiles", x, e, ids, out, 2 * n_units64, 77, 4 * b32 + tile, cus, t);
    }
  }
  return 0;
}

// rowstore_micro.hip -- what does the matrix-core pair stage pay for storing its output rows straight from the
// accumulators?  v_mfma_f64_16x16x4 leaves lane (li = lane & 15, lk = lane >> 4) with D[row lk + 4 g][col li], g = 0..3,
// per 16 x 16 tile: a store instruction is 64 lanes x 8 bytes = four 128-byte row segments.  Against it: the same
// 64 x 52 rows (BASELINE's config 5: 1M rows of 416 bytes = 416 MB per launch) written as 16 or 32 contiguous bytes per
// lane, plain and non-temporal.  Prints us per 416 MB and TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
constexpr int kW = 52, kRows = 64;

template <bool NT>
__device__ __forceinline__ void put(double *p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0: accumulator layout, 8 bytes per lane (wave = 16 rows x 4 column tiles; 4 waves per unit)
// MODE 1: 16 bytes per lane, row-major   MODE 2: 32 bytes per lane, row-major
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void store_rows(double *__restrict__ out, int n_units) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
    double *base = out + size_t(u) * kRows * kW;
    const double v = double(u) + tid;
    if (MODE == 0) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int col = 16 * n + li;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (col < kW) put<NT>(base + (16 * wave + lk + 4 * g) * kW + col, v + g);
      }
    } else if (MODE == 1) {
      for (int c = tid; c < kRows * kW / 2; c += 256) {
        double2 w; w.x = v; w.y = v + 1;
        if (NT) { __builtin_nontemporal_store(w.x, base + 2 * c); __builtin_nontemporal_store(w.y, base + 2 * c + 1); }
        else *reinterpret_cast<double2 *>(base + 2 * c) = w;
      }
    } else {
      for (int c = tid; c < kRows * kW / 4; c += 256) {
        double4 w; w.x = v; w.y = v + 1; w.z = v + 2; w.w = v + 3;
        if (NT) __builtin_nontemporal_store(*reinterpret_cast<__attribute__((ext_vector_type(4))) double *>(&w),
                                            reinterpret_cast<__attribute__((ext_vector_type(4))) double *>(base + 4 * c));
        else *reinterpret_cast<double4 *>(base + 4 * c) = w;
      }
    }
  }
}

template <int MODE, bool NT>
void run(const char *name, double *out, int n_units, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) store_rows<MODE, NT><<<grid, 256>>>(out, n_units);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) store_rows<MODE, NT><<<grid, 256>>>(out, n_units);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000 / reps, bytes = double(n_units) * kRows * kW * 8;
  printf("%-44s %-13s grid %5d: %7.1f us  %5.2f TB/s\n", name, NT ? "non-temporal" : "plain", grid, us, bytes / us / 1e6);
}

int main() {
  const int n_units = 15616;
  double *out; CK(hipMalloc((void **)&out, size_t(n_units) * kRows * kW * 8));
  for (int grid : {768, 3904, 15616}) {
    run<0, false>("accumulator layout, 8 B per lane", out, n_units, grid);
    run<0, true>("accumulator layout, 8 B per lane", out, n_units, grid);
    run<1, false>("row-major, 16 B per lane", out, n_units, grid);
    run<1, true>("row-major, 16 B per lane (as 2 x 8 B)", out, n_units, grid);
    run<2, false>("row-major, 32 B per lane", out, n_units, grid);
    run<2, true>("row-major, 32 B per lane", out, n_units, grid);
  }
  return 0;
}

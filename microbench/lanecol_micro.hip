// lanecol_micro.hip -- the dense pair stage for long rows (K, L ~ 50) as "lane = output column":
// a wave keeps one column of the rating tile per lane in registers, the pair's input row comes in
// through SCALAR loads (lane-uniform), so the mat-vec is DP v_fma_f64 with an SGPR operand per pair
// and touches no LDS.  ROLE 0: T[q][l] = sum_d x[q][d] tile[d][l];  ROLE 1: S[k][l] += x[q][k] e[q][l].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

typedef const double __attribute__((address_space(4))) * cptr;

template <int DP, int ROLE, int XB>
__global__ __launch_bounds__(256) void lanecol(const double* __restrict__ tile, const double* __restrict__ xtab,
                                               const double* __restrict__ etab, const int* __restrict__ item,
                                               double* __restrict__ out, double* __restrict__ slabs,
                                               int pairs_per_wave, int n_pairs) {
  const int lane = threadIdx.x & 63;
  const int wave_g = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int l = lane < DP ? lane : DP - 1;
  double t[DP];
  if (ROLE == 0) {
#pragma unroll
    for (int d = 0; d < DP; ++d) t[d] = tile[d * DP + l];
  } else {
#pragma unroll
    for (int d = 0; d < DP; ++d) t[d] = 0.0;
  }
  const int q0 = wave_g * pairs_per_wave;
  const int q1 = min(q0 + pairs_per_wave, n_pairs);
  for (int q = q0; q < q1; ++q) {
    const cptr row = (cptr)(reinterpret_cast<uintptr_t>(xtab + static_cast<size_t>(q) * DP));
    if (ROLE == 0) {
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int d0 = 0; d0 < DP; d0 += XB) {
        double x[XB];
#pragma unroll
        for (int j = 0; j < XB; ++j) x[j] = row[d0 + j];
#pragma unroll
        for (int j = 0; j < XB; j += 2) {
          acc0 = fma(x[j], t[d0 + j], acc0);
          acc1 = fma(x[j + 1], t[d0 + j + 1], acc1);
        }
      }
      if (lane < DP) out[static_cast<size_t>(q) * DP + lane] = acc0 + acc1;
    } else {
      const int it = __builtin_amdgcn_readfirstlane(item[q]);
      const double e = etab[static_cast<size_t>(it) * DP + l];
#pragma unroll
      for (int d0 = 0; d0 < DP; d0 += XB) {
        double x[XB];
#pragma unroll
        for (int j = 0; j < XB; ++j) x[j] = row[d0 + j];
#pragma unroll
        for (int j = 0; j < XB; ++j) t[d0 + j] = fma(x[j], e, t[d0 + j]);
      }
    }
  }
  if (ROLE == 1 && lane < DP) {
#pragma unroll
    for (int d = 0; d < DP; ++d) slabs[(static_cast<size_t>(wave_g) * DP + d) * DP + lane] = t[d];
  }
}

template <int DP, int ROLE, int XB>
void run(int n_pairs, int ppw, const double* tile, const double* x, const double* e, const int* item,
         double* out, double* slabs) {
  const int waves = (n_pairs + ppw - 1) / ppw;
  const int blocks = (waves + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) lanecol<DP, ROLE, XB><<<blocks, 256>>>(tile, x, e, item, out, slabs, ppw, n_pairs);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) lanecol<DP, ROLE, XB><<<blocks, 256>>>(tile, x, e, item, out, slabs, ppw, n_pairs);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = 2.0 * n_pairs * DP * DP;
  printf("DP=%d role=%d xb=%2d pairs/wave=%4d blocks=%6d : %8.1f us  %6.2f TFLOP/s\n", DP, ROLE, XB, ppw, blocks,
         ms * 1000 / reps, flops / (ms * 1e-3 / reps) / 1e12);
}

int main() {
  const int n_pairs = 1000000, n_items = 100000;
  constexpr int DP = 52;
  std::vector<double> h((size_t)n_pairs * DP);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
  std::vector<int> hi(n_pairs);
  for (int q = 0; q < n_pairs; ++q) hi[q] = (int)(((size_t)q * 7919u) % n_items);
  double *tile, *x, *e, *out, *slabs; int* item;
  CK(hipMalloc((void**)&tile, DP * DP * 8)); CK(hipMalloc((void**)&x, h.size() * 8)); CK(hipMalloc((void**)&e, (size_t)n_items * DP * 8));
  CK(hipMalloc((void**)&out, h.size() * 8)); CK(hipMalloc((void**)&slabs, (size_t)(n_pairs / 64 + 8) * DP * DP * 8));
  CK(hipMalloc((void**)&item, n_pairs * 4));
  CK(hipMemcpy(tile, h.data(), DP * DP * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(e, h.data(), (size_t)n_items * DP * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(item, hi.data(), n_pairs * 4, hipMemcpyHostToDevice));
  for (int ppw : {64, 128, 256, 512}) {
    run<DP, 0, 4>(n_pairs, ppw, tile, x, e, item, out, slabs);
    run<DP, 0, 26>(n_pairs, ppw, tile, x, e, item, out, slabs);
    run<DP, 0, 52>(n_pairs, ppw, tile, x, e, item, out, slabs);
    run<DP, 1, 4>(n_pairs, ppw, tile, x, e, item, out, slabs);
    run<DP, 1, 26>(n_pairs, ppw, tile, x, e, item, out, slabs);
  }
  return 0;
}

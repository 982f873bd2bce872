// latency_micro.hip -- dependent-load latency and kernel-launch floor on this box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void chase(const int* __restrict__ next, int start, int steps, int* out, long long* cyc) {
  int p = start;
  long long t0 = clock64();
  for (int s = 0; s < steps; ++s) p = next[p];
  long long t1 = clock64();
  out[0] = p; cyc[0] = t1 - t0;
}
__global__ void empty_kernel(int* out) { if (threadIdx.x == 1234567) out[0] = 1; }
__global__ void touch(const double* in, double* out, int n) {  // each block: load -> dependent load -> store
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] * 2.0;
}
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int* dout; long long* dcyc; CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dcyc, 64));
  for (size_t mb : {1, 3, 16, 64, 512}) {
    const size_t n = mb * 1024 * 1024 / 4 / 32;  // one int per 128-byte line
    std::vector<int> perm(n); for (size_t i = 0; i < n; ++i) perm[i] = (int)i;
    std::mt19937 rng(3); std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int> next(n * 32, 0);
    for (size_t i = 0; i < n; ++i) next[(size_t)perm[i] * 32] = perm[(i + 1) % n] * 32;
    int* dn; CK(hipMalloc(&dn, next.size() * 4)); CK(hipMemcpy(dn, next.data(), next.size() * 4, hipMemcpyHostToDevice));
    const int steps = 4000;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(chase, 1, 1, 0, 0, dn, perm[0] * 32, steps, dout, dcyc);
      CK(hipDeviceSynchronize());
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chase, 1, 1, 0, 0, dn, perm[0] * 32, steps, dout, dcyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc; CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("pointer chase over %4zu MB (%zu lines): %7.1f ns per dependent load (%lld clock64 ticks/load)\n", mb, n, ms * 1e6 / steps, cyc / steps);
    CK(hipFree(dn));
  }
  // launch floor
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(empty_kernel, 256, 256, 0, 0, dout);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 1000; ++r) hipLaunchKernelGGL(empty_kernel, 256, 256, 0, 0, dout);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel, 256 blocks, back to back: %.2f us per launch\n", ms);
  double *a, *b; const int n = 2 * 1024 * 1024; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
  CK(hipMemset(a, 0, n * 8));
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(touch, n / 256, 256, 0, 0, a, b, n);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(touch, n / 256, 256, 0, 0, a, b, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("copy-scale 16 MB in -> 16 MB out (same buffers every launch): %.2f us per launch\n", ms * 5);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 200; ++r) { hipLaunchKernelGGL(touch, n / 256, 256, 0, 0, a, b, n); hipLaunchKernelGGL(touch, n / 256, 256, 0, 0, b, a, n); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("ping-pong a->b, b->a (consumer reads what the previous kernel wrote): %.2f us per launch\n", ms * 2.5);
  return 0;
}

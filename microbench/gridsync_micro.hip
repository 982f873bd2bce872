// gridsync_micro.hip -- what does a grid-wide barrier cost (cooperative launch) next to a kernel
// boundary?  One kernel does `iters` x { touch memory ; grid.sync() }.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void syncs(double *buf, int iters, int work) {
  cg::grid_group grid = cg::this_grid();
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t n = size_t(gridDim.x) * blockDim.x;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    for (int w = 0; w < work; ++w) acc += buf[(gid + size_t(w) * 7919 + it) % n];
    buf[gid] = acc;
    grid.sync();
  }
}
__global__ __launch_bounds__(256) void one(double *buf, int work) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t n = size_t(gridDim.x) * blockDim.x;
  double acc = 0.0;
  for (int w = 0; w < work; ++w) acc += buf[(gid + size_t(w) * 7919) % n];
  buf[gid] = acc;
}

int main() {
  int dev = 0, cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  for (int per_cu : {1, 2, 4, 6}) {
    const int blocks = cus * per_cu;
    double *buf;
    CK(hipMalloc((void **)&buf, size_t(blocks) * 256 * 8));
    CK(hipMemset(buf, 0, size_t(blocks) * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int work : {0, 4}) {
      int iters = 200;
      void *args[] = {&buf, &iters, &work};
      CK(hipLaunchCooperativeKernel((void *)syncs, dim3(blocks), dim3(256), args, 0, 0));
      CK(hipEventRecord(e0));
      CK(hipLaunchCooperativeKernel((void *)syncs, dim3(blocks), dim3(256), args, 0, 0));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const float coop = ms * 1000 / iters;
      for (int i = 0; i < 10; ++i) one<<<blocks, 256>>>(buf, work);
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) one<<<blocks, 256>>>(buf, work);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%4d blocks (%d per CU), work %d: grid.sync step %6.2f us   separate launches %6.2f us per step\n",
             blocks, per_cu, work, coop, ms * 1000 / iters);
    }
    CK(hipFree(buf));
  }
  return 0;
}

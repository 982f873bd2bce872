// mfma4_micro.hip -- v_mfma_f64_4x4x4_4b_f64 on gfx950: which lane holds which operand / result entry, and what an
// instruction costs next to v_mfma_f64_16x16x4_f64 (round 4: the 4-wide remainders of K = L = 50 tiles, 52 = 3*16 + 4).
//   hipcc -O3 --offload-arch=gfx950 microbench/mfma4_micro.hip -o /tmp/mfma4 && /tmp/mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// one wave; for every (la, lb): A = 1 in lane la only, B = 1 in lane lb only -> which lane's D is 1
__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m ? __builtin_ctzll(m) + 64 * (__builtin_popcountll(m) - 1) : -1;
    }
}

template <int KIND, int NACC>
__global__ void rate(double* out, long long* cyc, int reps) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  d4 acc16[NACC];
  double acc4[NACC];
  for (int i = 0; i < NACC; ++i) { acc16[i] = d4{0, 0, 0, 0}; acc4[i] = 0.0; }
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
      else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += KIND == 0 ? acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3] : acc4[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  int* d_out; CK(hipMalloc(&d_out, 64 * 64 * sizeof(int)));
  probe<<<1, 64>>>(d_out);
  std::vector<int> h(64 * 64);
  CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(int), hipMemcpyDeviceToHost));
  // per A lane: the set of B lanes it meets and where the products land
  printf("A lane -> (B lane : D lane) pairs that give a non-zero result\n");
  for (int la = 0; la < 64; ++la) {
    printf("a%02d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf(" b%02d->d%02d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  double* d_o; long long* d_c;
  CK(hipMalloc(&d_o, 1024 * 256 * sizeof(double))); CK(hipMalloc(&d_c, sizeof(long long)));
  const int reps = 2000;
  auto run = [&](const char* name, auto kern, int nacc, int blocks, int threads) {
    kern<<<blocks, threads>>>(d_o, d_c, reps);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); kern<<<blocks, threads>>>(d_o, d_c, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, d_c, sizeof(c), hipMemcpyDeviceToHost));
    printf("%-28s %d acc, %4d x %3d threads: %7.1f clock64 ticks per instruction (one wave's view), %8.3f ms\n", name, nacc, blocks, threads,
           double(c) / (double(reps) * nacc), ms);
  };
  run("16x16x4 f64", rate<0, 1>, 1, 1, 64);
  run("16x16x4 f64", rate<0, 4>, 4, 1, 64);
  run("4x4x4_4b f64", rate<1, 1>, 1, 1, 64);
  run("4x4x4_4b f64", rate<1, 4>, 4, 1, 64);
  run("4x4x4_4b f64", rate<1, 8>, 8, 1, 64);
  run("16x16x4 f64 (4 waves/CU x 256 CUs x 4)", rate<0, 4>, 4, 1024, 256);
  run("4x4x4_4b f64 (same)", rate<1, 4>, 4, 1024, 256);
  return 0;
}

// flag_micro.hip -- what does a ONE-DIRECTIONAL hand-over between two roles of ONE grid cost, next to a kernel
// boundary?  (DESIGN: "one launch per iteration" for small problems.)  P producer workgroups write a table and
// each does one release-increment of a per-group counter (8 groups: workgroups are dealt to the XCDs round robin);
// the last arrival of a group does one release-increment of the top counter.  M consumer workgroups poll the top
// counter (relaxed loads + s_sleep), then one acquire fence, then gather what the producers wrote and check it.
// The grid is launched with hipLaunchCooperativeKernel: co-residency of all P + M workgroups is GUARANTEED (the
// launch fails loudly when it does not fit), so nothing rests on dispatch order.  Every poll loop is bounded: a
// consumer that does not see the flag within kSpinLimit polls sets an error word and leaves -- the grid always drains.
// Measured per (P, M): arrival of the LAST producer -> release of the FIRST / LAST consumer (100 MHz clock stamps),
// the whole launch, and the same work as two ordinary launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

constexpr int kThreads = 256, kGroups = 8, kSpinLimit = 400000;

struct Args {
  const double *in; double *mid; double *out;
  unsigned *sub;             // [kGroups] arrivals per group (monotone: round * members)
  unsigned *top;             // groups that are complete (monotone: round * kGroups)
  unsigned *err;             // != 0: a consumer gave up
  unsigned long long *stamp; // [P + M] arrival resp. release time
  int P, M, W, round, nt;    // W doubles written per producer thread; nt: producer rows as non-temporal stores
  int one_release;           // 1: only the LAST arrival of a group does the release (one L2 write-back per XCD), the others arrive relaxed
  unsigned *xcc_mismatch;    // workgroups whose XCC_ID is not blockIdx % 8
};

__device__ __forceinline__ double value_of(int b, int w, int t, int round) { return double(b) * 1.5 + double(w) * 0.25 + double(t) * 0.001 + double(round); }

__device__ __forceinline__ void produce(const Args &a, int b) {
  const int t = threadIdx.x;
  const double x = a.in[(size_t(b) * kThreads + t) & 0xFFFF];   // (a load first, as a real role would have)
  for (int w = 0; w < a.W; ++w) {
    double *dst = a.mid + (size_t(b) * a.W + w) * kThreads + t;
    const double v = value_of(b, w, t, a.round) + x;
    if (a.nt) __builtin_nontemporal_store(v, dst); else *dst = v;
  }
}
__device__ __forceinline__ double consume(const Args &a, int c) {
  const int t = threadIdx.x;
  double acc = 0.0;
  for (int w = 0; w < a.W; ++w) {   // rows of pseudo-random producers: other XCDs' output as likely as not
    const int b = int((unsigned(c) * 2654435761u + unsigned(w) * 40503u + unsigned(t >> 4) * 97u) % unsigned(a.P));
    acc += a.mid[(size_t(b) * a.W + w) * kThreads + t] - value_of(b, w, t, a.round);
  }
  return acc;   // == W * in[..] contributions; with in == 0: exactly 0 when every row was visible
}

__global__ __launch_bounds__(kThreads) void flagged(Args a) {
  const int b = blockIdx.x;
  if (b < a.P) {
    produce(a, b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its stores are in the L2 before the barrier
    __syncthreads();
    if (threadIdx.x == 0) {
      const int g = b % kGroups, members = (a.P - g + kGroups - 1) / kGroups;
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
      if (int(xcc) != g) atomicAdd(a.xcc_mismatch, 1u);
      if (a.one_release) {
        // this workgroup's stores have reached its XCD's L2 (the barrier above followed every wave's s_waitcnt);
        // a relaxed arrival, and the group's LAST arrival writes the whole L2 back once for all of them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(a.sub + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == unsigned(members) * unsigned(a.round)) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_fetch_add(a.top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        const unsigned old = __hip_atomic_fetch_add(a.sub + g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == unsigned(members) * unsigned(a.round)) __hip_atomic_fetch_add(a.top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      }
      a.stamp[b] = wall_clock64();
    }
    return;
  }
  const int c = b - a.P;
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const unsigned target = unsigned(min(a.P, kGroups)) * unsigned(a.round);
    int spins = 0, good = 1;
    while (__hip_atomic_load(a.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kSpinLimit) { good = 0; atomicAdd(a.err, 1u); break; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (agent scope is the default for a HIP thread fence)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    a.stamp[b] = wall_clock64();
    ok = good;
  }
  __syncthreads();
  if (!ok) return;
  a.out[size_t(c) * kThreads + threadIdx.x] = consume(a, c);
}
__global__ __launch_bounds__(kThreads) void producers_only(Args a) { produce(a, blockIdx.x); }
__global__ __launch_bounds__(kThreads) void consumers_only(Args a) { a.out[size_t(blockIdx.x) * kThreads + threadIdx.x] = consume(a, blockIdx.x); }

int main() {
  int dev = 0, cus = 0, per_cu = 0, coop = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, flagged, kThreads, 0));
  printf("%d CUs, cooperative launch %s, %d workgroups of %d threads co-resident per CU\n", cus, coop ? "yes" : "NO", per_cu, kThreads);
  if (!coop) { printf("FAILED: no cooperative launch on this device\n"); return 1; }
  const int shapes[][3] = {{400, 245, 8}, {400, 245, 32}, {256, 256, 8}, {768, 256, 8}, {1565, 182, 16}, {1024, 1024, 8}, {8, 8, 8}};
  for (auto &sh : shapes) {
    for (int variant = 0; variant < 3; ++variant) {
      const int nt = variant == 1, one_release = variant == 2;
      Args a{};
      a.P = sh[0]; a.M = sh[1]; a.W = sh[2]; a.nt = nt; a.one_release = one_release;
      if (a.P + a.M > per_cu * cus) { printf("P=%d M=%d: FAILED to fit (%d slots) -- skipped, loudly\n", a.P, a.M, per_cu * cus); continue; }
      double *in, *mid, *out; unsigned *cnt; unsigned long long *stamp;
      CK(hipMalloc((void **)&in, 65536 * 8)); CK(hipMemset(in, 0, 65536 * 8));
      CK(hipMalloc((void **)&mid, size_t(a.P) * a.W * kThreads * 8));
      CK(hipMalloc((void **)&out, size_t(a.M) * kThreads * 8));
      CK(hipMalloc((void **)&cnt, 64 * 4)); CK(hipMemset(cnt, 0, 64 * 4));
      CK(hipMalloc((void **)&stamp, size_t(a.P + a.M) * 8));
      a.in = in; a.mid = mid; a.out = out; a.sub = cnt; a.top = cnt + 16; a.err = cnt + 32; a.xcc_mismatch = cnt + 48; a.stamp = stamp;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int reps = 200;
      std::vector<double> first, last;
      std::vector<unsigned long long> st(a.P + a.M);
      std::vector<double> res(size_t(a.M) * kThreads);
      double bad = 0.0;
      float ms_total = 0.f;
      int round = 0;
      auto launch = [&]() {
        a.round = ++round;
        void *args[] = {&a};
        CK(hipLaunchCooperativeKernel((void *)flagged, dim3(a.P + a.M), dim3(kThreads), args, 0, 0));
      };
      for (int r = 0; r < 5; ++r) launch();
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_total, e0, e1));
      // the same kernel through an ORDINARY launch (no co-residency guarantee from the runtime: the grid fits the
      // device -- checked above -- and a consumer that never sees the flag gives up after kSpinLimit polls)
      float ms_plain = 0.f;
      for (int r = 0; r < 5; ++r) { a.round = ++round; flagged<<<a.P + a.M, kThreads>>>(a); }
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) { a.round = ++round; flagged<<<a.P + a.M, kThreads>>>(a); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_plain, e0, e1));
      CK(hipMemcpy(res.data(), out, res.size() * 8, hipMemcpyDeviceToHost));
      for (double v : res) bad = std::max(bad, v < 0 ? -v : v);
      for (int r = 0; r < 15; ++r) {   // stamps and results: one launch at a time
        launch();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), stamp, st.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(res.data(), out, res.size() * 8, hipMemcpyDeviceToHost));
        for (double v : res) bad = std::max(bad, v < 0 ? -v : v);
        const unsigned long long arr = *std::max_element(st.begin(), st.begin() + a.P);
        const unsigned long long r0 = *std::min_element(st.begin() + a.P, st.end()), r1 = *std::max_element(st.begin() + a.P, st.end());
        first.push_back((double(r0) - double(arr)) / 100.0); last.push_back((double(r1) - double(arr)) / 100.0);
      }
      unsigned err = 0, mism = 0; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&mism, a.xcc_mismatch, 4, hipMemcpyDeviceToHost));
      // the same work as two ordinary launches
      a.round = round;
      for (int i = 0; i < 10; ++i) { producers_only<<<a.P, kThreads>>>(a); consumers_only<<<a.M, kThreads>>>(a); }
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) { producers_only<<<a.P, kThreads>>>(a); consumers_only<<<a.M, kThreads>>>(a); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
      // and each alone (launch floors)
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) producers_only<<<a.P, kThreads>>>(a);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float msp; CK(hipEventElapsedTime(&msp, e0, e1));
      std::sort(first.begin(), first.end()); std::sort(last.begin(), last.end());
      printf("P=%4d M=%4d W=%2d %s: last arrival -> first / last release  median %5.2f / %5.2f us (min %5.2f / %5.2f); one flagged launch cooperative %6.2f us / ordinary %5.2f us, two launches %6.2f us, producers alone %5.2f us; max |error| %.1e, gave up %u, producers off their XCD %u of %d\n",
             a.P, a.M, a.W, one_release ? "1 release/XCD" : (nt ? "nt stores    " : "plain        "), first[first.size() / 2], last[last.size() / 2], first[0], last[0],
             ms_total * 1000 / reps, ms_plain * 1000 / reps, ms2 * 1000 / reps, msp * 1000 / reps, bad, err, mism, a.P * (5 + reps + reps + 5 + 15));
      CK(hipFree(in)); CK(hipFree(mid)); CK(hipFree(out)); CK(hipFree(cnt)); CK(hipFree(stamp));
    }
  }
  return 0;
}

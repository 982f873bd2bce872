// gather_micro.hip -- which row layout serves seg_pass's random 160-byte row gathers fastest?
// Standalone: hipcc -O3 --offload-arch=gfx950 gather_micro.hip -o gather_micro && ./gather_micro
// Same loop shape as seg_pass_kernel (16 lanes per segment, 2 doubles per lane, B rows in
// flight per group), random indices, 100k segments x 10 gathers from a 100k-row table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ double dpp_move(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum16(double x) {
  x += dpp_move<0xB1>(x); x += dpp_move<0x4E>(x); x += dpp_move<0x141>(x); x += dpp_move<0x140>(x);
  return x;
}
__device__ __forceinline__ double2 ld(const double* p, bool nt) {
  if (nt) {
    double2 r;
    r.x = __builtin_nontemporal_load(p);
    r.y = __builtin_nontemporal_load(p + 1);
    return r;
  }
  return *reinterpret_cast<const double2*>(p);
}

// MODE 0: one table, row stride `stride` doubles (20 = packed, 32 = two aligned lines)
// MODE 1: main table 16 doubles/row (one 128-B line) + tail table 4 doubles/row
template <int MODE, int B, bool NT_MAIN, bool NT_STREAM>
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ fixed, const double* __restrict__ tab,
                                                     const double* __restrict__ tail, const int* __restrict__ off,
                                                     const int* __restrict__ idx, double* __restrict__ out, int nseg,
                                                     int stride, int k) {
  const int seg = blockIdx.x * 16 + threadIdx.x / 16, gl = threadIdx.x % 16;
  if (seg >= nseg) return;
  const bool act = gl * 2 < k;
  const int lo = act ? gl * 2 : 0;
  double2 f = *reinterpret_cast<const double2*>(fixed + (size_t)seg * 20 + lo);
  if (!act) { f.x = 0; f.y = 0; }
  double2 acc = {0.0, 0.0};
  const int beg = off[seg], end = off[seg + 1];
  for (int n = beg; n < end; n += B) {
    int id[B]; double2 g[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int* ip = idx + min(n + b, end - 1);
      id[b] = NT_STREAM ? __builtin_nontemporal_load(ip) : *ip;
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const double* p;
      if (MODE == 0) p = tab + (size_t)id[b] * stride + lo;
      else p = (lo < 16) ? tab + (size_t)id[b] * 16 + lo : tail + (size_t)id[b] * 4 + (lo - 16);
      g[b] = ld(p, NT_MAIN && (MODE == 0 || lo < 16));
    }
#pragma unroll
    for (int b = 0; b < B; ++b) if (n + b < end) {
      double part = fma(g[b].x, f.x, g[b].y * f.y);
      const double s = sum16(part);
      const double w = 1.0 / fmax(s, 2.2e-16);
      acc.x = fma(g[b].x, w, acc.x); acc.y = fma(g[b].y, w, acc.y);
    }
  }
  if (act) *reinterpret_cast<double2*>(out + (size_t)seg * 20 + lo) = acc;
}

// P1: every lane of the group preloads one index of the segment (one coalesced load), indices
// are then broadcast inside the group with ds_bpermute: the chain is off -> idx -> rows.
template <int MODE, int B, bool PIPE>
__global__ __launch_bounds__(256) void gather_p1(const double* __restrict__ fixed, const double* __restrict__ tab,
                                                 const double* __restrict__ tail, const int* __restrict__ off,
                                                 const int* __restrict__ idx, double* __restrict__ out, int nseg,
                                                 int stride, int k) {
  const int seg = blockIdx.x * 16 + threadIdx.x / 16, gl = threadIdx.x % 16;
  if (seg >= nseg) return;
  const bool act = gl * 2 < k;
  const int lo = act ? gl * 2 : 0;
  const int beg = off[seg], end = off[seg + 1];
  double2 f = *reinterpret_cast<const double2*>(fixed + (size_t)seg * 20 + lo);
  if (!act) { f.x = 0; f.y = 0; }
  double2 acc = {0.0, 0.0};
  auto rowptr = [&](int id) -> const double* {
    if (MODE == 0) return tab + (size_t)id * stride + lo;
    return (lo < 16) ? tab + (size_t)id * 16 + lo : tail + (size_t)id * 4 + (lo - 16);
  };
  auto consume = [&](const double2& g) {
    double part = fma(g.x, f.x, g.y * f.y);
    const double s = sum16(part);
    const double w = 1.0 / fmax(s, 2.2e-16);
    acc.x = fma(g.x, w, acc.x); acc.y = fma(g.y, w, acc.y);
  };
  for (int c = beg; c < end; c += 16) {
    const int cnt = min(16, end - c);
    const int mine = idx[c + min(gl, cnt - 1)];
    if (!PIPE) {
      for (int j = 0; j < cnt; j += B) {
        double2 g[B];
#pragma unroll
        for (int b = 0; b < B; ++b) g[b] = *reinterpret_cast<const double2*>(rowptr(__shfl(mine, min(j + b, cnt - 1), 16)));
#pragma unroll
        for (int b = 0; b < B; ++b) if (j + b < cnt) consume(g[b]);
      }
    } else {
      double2 cur[B], nxt[B];
#pragma unroll
      for (int b = 0; b < B; ++b) cur[b] = *reinterpret_cast<const double2*>(rowptr(__shfl(mine, min(b, cnt - 1), 16)));
      for (int j = 0; j < cnt; j += B) {
        if (j + B < cnt) {
#pragma unroll
          for (int b = 0; b < B; ++b) nxt[b] = *reinterpret_cast<const double2*>(rowptr(__shfl(mine, min(j + B + b, cnt - 1), 16)));
        }
#pragma unroll
        for (int b = 0; b < B; ++b) if (j + b < cnt) consume(cur[b]);
#pragma unroll
        for (int b = 0; b < B; ++b) cur[b] = nxt[b];
      }
    }
  }
  if (act) *reinterpret_cast<double2*>(out + (size_t)seg * 20 + lo) = acc;
}

// F1: no segments at all -- every group walks a strided list of triples with B independent
// row loads in flight (the memory system's ceiling for this gather shape).
template <int MODE, int B>
__global__ __launch_bounds__(256) void gather_flat(const double* __restrict__ fixed, const double* __restrict__ tab,
                                                   const double* __restrict__ tail, const int* __restrict__ off,
                                                   const int* __restrict__ idx, double* __restrict__ out, int nseg,
                                                   int stride, int k) {
  extern __shared__ double dummy[];
  const int ngroups = gridDim.x * 16;
  const int grp = blockIdx.x * 16 + threadIdx.x / 16, gl = threadIdx.x % 16;
  const bool act = gl * 2 < k;
  const int lo = act ? gl * 2 : 0;
  const int n = nseg * 10;
  double2 acc = {0.0, 0.0};
  for (int t = grp * B; t < n; t += ngroups * B) {
    int id[B]; double2 g[B];
#pragma unroll
    for (int b = 0; b < B; ++b) id[b] = idx[min(t + b, n - 1)];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const double* p;
      if (MODE == 0) p = tab + (size_t)id[b] * stride + lo;
      else p = (lo < 16) ? tab + (size_t)id[b] * 16 + lo : tail + (size_t)id[b] * 4 + (lo - 16);
      g[b] = *reinterpret_cast<const double2*>(p);
    }
#pragma unroll
    for (int b = 0; b < B; ++b) { acc.x += g[b].x; acc.y += g[b].y; }
  }
  if (act) *reinterpret_cast<double2*>(out + (size_t)(grp % nseg) * 20 + lo) = acc;
}

// P2: like P1 split, but every group works on TWO segments at once (seg 2g and 2g+1) so the
// offset -> index -> row chains of the two overlap.
template <int B>
__global__ __launch_bounds__(256) void gather_p2(const double* __restrict__ fixed, const double* __restrict__ tab,
                                                 const double* __restrict__ tail, const int* __restrict__ off,
                                                 const int* __restrict__ idx, double* __restrict__ out, int nseg,
                                                 int stride, int k) {
  const int grp = blockIdx.x * 16 + threadIdx.x / 16, gl = threadIdx.x % 16;
  const int sA = grp * 2, sB = grp * 2 + 1;
  if (sA >= nseg) return;
  const bool hasB = sB < nseg;
  const bool act = gl * 2 < k;
  const int lo = act ? gl * 2 : 0;
  const double* gbase = (lo < 16) ? tab + lo : tail + (lo - 16);
  const size_t gstride = (lo < 16) ? 16 : 4;
  int beg[2], end[2];
  beg[0] = off[sA]; end[0] = off[sA + 1]; beg[1] = hasB ? off[sB] : 0; end[1] = hasB ? off[sB + 1] : 0;
  double2 f[2], acc[2];
  f[0] = *reinterpret_cast<const double2*>(fixed + (size_t)sA * 20 + lo);
  f[1] = *reinterpret_cast<const double2*>(fixed + (size_t)(hasB ? sB : sA) * 20 + lo);
  if (!act) { f[0].x = f[0].y = f[1].x = f[1].y = 0; }
  acc[0].x = acc[0].y = acc[1].x = acc[1].y = 0;
  int mine[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) mine[s] = (end[s] > beg[s]) ? idx[beg[s] + min(gl, end[s] - beg[s] - 1)] : 0;  // segments <= 16 here
  const int cnt0 = end[0] - beg[0], cnt1 = end[1] - beg[1];
  const int cmax = max(cnt0, cnt1);
  for (int n = 0; n < cmax; n += B) {
    double2 g[2][B];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cnt = s ? cnt1 : cnt0;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int id = __shfl(mine[s], min(n + b, max(cnt - 1, 0)), 16);
        g[s][b] = *reinterpret_cast<const double2*>(gbase + (size_t)id * gstride);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cnt = s ? cnt1 : cnt0;
#pragma unroll
      for (int b = 0; b < B; ++b) if (n + b < cnt) {
        double part = fma(g[s][b].x, f[s].x, g[s][b].y * f[s].y);
        const double sm = sum16(part);
        const double w = 1.0 / fmax(sm, 2.2e-16);
        acc[s].x = fma(g[s][b].x, w, acc[s].x); acc[s].y = fma(g[s][b].y, w, acc[s].y);
      }
    }
  }
  if (act) {
    *reinterpret_cast<double2*>(out + (size_t)sA * 20 + lo) = acc[0];
    if (hasB) *reinterpret_cast<double2*>(out + (size_t)sB * 20 + lo) = acc[1];
  }
}

int main() {
  const int nseg = 100000, deg = 10, rows = 100000, k = 20;
  const int n = nseg * deg;
  std::mt19937 rng(1);
  std::vector<int> off(nseg + 1), idx(n);
  for (int s = 0; s <= nseg; ++s) off[s] = s * deg;
  for (int j = 0; j < n; ++j) idx[j] = rng() % rows;
  std::vector<double> tab((size_t)rows * 32), fixed((size_t)nseg * 20);
  for (auto& v : tab) v = 0.5 + (rng() % 1000) * 1e-3;
  for (auto& v : fixed) v = 0.5 + (rng() % 1000) * 1e-3;
  double *dtab, *dtail, *dfixed, *dout; int *doff, *didx;
  CK(hipMalloc(&dtab, tab.size() * 8)); CK(hipMalloc(&dtail, (size_t)rows * 4 * 8));
  CK(hipMalloc(&dfixed, fixed.size() * 8)); CK(hipMalloc(&dout, fixed.size() * 8));
  CK(hipMalloc(&doff, off.size() * 4)); CK(hipMalloc(&didx, idx.size() * 4));
  CK(hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtail, tab.data(), (size_t)rows * 4 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfixed, fixed.data(), fixed.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(doff, off.data(), off.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int lds_bytes = 0, grid_override = 0;
  auto run = [&](const char* name, auto kern, int stride, int kk) {
    const int grid = grid_override ? grid_override : (nseg + 15) / 16;
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, grid, 256, lds_bytes, 0, dfixed, dtab, dtail, doff, didx, dout, nseg, stride, kk);
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, 256, lds_bytes, 0, dfixed, dtab, dtail, doff, didx, dout, nseg, stride, kk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    printf("%-44s %8.2f us   %7.1f GB/s useful (%d-B rows)\n", name, us, (double)n * kk * 8 / us / 1e3, kk * 8);
  };
  run("packed stride20 B=4", gather_kernel<0, 4, false, false>, 20, 20);
  run("packed stride20 B=8", gather_kernel<0, 8, false, false>, 20, 20);
  run("packed stride20 B=2", gather_kernel<0, 2, false, false>, 20, 20);
  run("packed stride20 B=4 nt-stream", gather_kernel<0, 4, false, true>, 20, 20);
  run("packed stride20 B=4 nt-all", gather_kernel<0, 4, true, true>, 20, 20);
  run("aligned stride32 (2 lines) B=4", gather_kernel<0, 4, false, false>, 32, 20);
  run("aligned stride24 B=4", gather_kernel<0, 4, false, false>, 24, 20);
  run("split 16+4 B=4", gather_kernel<1, 4, false, false>, 16, 20);
  run("split 16+4 B=8", gather_kernel<1, 8, false, false>, 16, 20);
  run("split 16+4 B=4 nt-main", gather_kernel<1, 4, true, false>, 16, 20);
  run("split 16+4 B=4 nt-main nt-stream", gather_kernel<1, 4, true, true>, 16, 20);
  run("split 16+4 B=8 nt-main nt-stream", gather_kernel<1, 8, true, true>, 16, 20);
  run("P1 packed B=4", gather_p1<0, 4, false>, 20, 20);
  run("P1 packed B=8", gather_p1<0, 8, false>, 20, 20);
  run("P1 packed B=12", gather_p1<0, 12, false>, 20, 20);
  run("P1 packed B=16", gather_p1<0, 16, false>, 20, 20);
  run("P1 packed B=4 pipelined", gather_p1<0, 4, true>, 20, 20);
  run("P1 packed B=6 pipelined", gather_p1<0, 6, true>, 20, 20);
  run("P1 split B=4", gather_p1<1, 4, false>, 16, 20);
  grid_override = (nseg / 2 + 15) / 16;
  run("P2 split two segments per group B=4", gather_p2<4>, 16, 20);
  run("P2 split two segments per group B=6", gather_p2<6>, 16, 20);
  run("P2 split two segments per group B=10", gather_p2<10>, 16, 20);
  grid_override = 0;
  run("P1 split B=8", gather_p1<1, 8, false>, 16, 20);
  run("P1 split B=12", gather_p1<1, 12, false>, 16, 20);
  run("P1 split B=16", gather_p1<1, 16, false>, 16, 20);
  run("P1 split B=4 pipelined", gather_p1<1, 4, true>, 16, 20);
  run("P1 split B=6 pipelined", gather_p1<1, 6, true>, 16, 20);
  run("P1 K=16 one line B=12", gather_p1<0, 12, false>, 16, 16);
  for (int g : {512, 1024, 2048, 4096, 6250}) {
    grid_override = g;
    char nm[64];
    snprintf(nm, 64, "FLAT packed B=4 grid=%d", g); run(nm, gather_flat<0, 4>, 20, 20);
    snprintf(nm, 64, "FLAT packed B=8 grid=%d", g); run(nm, gather_flat<0, 8>, 20, 20);
    snprintf(nm, 64, "FLAT split  B=4 grid=%d", g); run(nm, gather_flat<1, 4>, 16, 20);
    snprintf(nm, 64, "FLAT split  B=8 grid=%d", g); run(nm, gather_flat<1, 8>, 16, 20);
    snprintf(nm, 64, "FLAT K=16   B=8 grid=%d", g); run(nm, gather_flat<0, 8>, 16, 16);
  }
  grid_override = 0;
  for (int l : {20 * 1024, 40 * 1024, 80 * 1024}) {
    lds_bytes = l;
    char nm[64];
    snprintf(nm, 64, "P1 split B=4, LDS %d KB/block (occupancy cap)", l / 1024); run(nm, gather_p1<1, 4, false>, 16, 20);
  }
  lds_bytes = 0;
  run("K=16 only: one aligned line B=4", gather_kernel<0, 4, false, false>, 16, 16);
  run("K=16 only: one aligned line B=8", gather_kernel<0, 8, false, false>, 16, 16);
  return 0;
}

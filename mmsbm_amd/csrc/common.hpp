// common.hpp -- error plumbing and the device helpers every kernel uses (DPP moves, group sums, vector loads)
// Included first by every translation unit of the library (prelude.hpp).
#pragma once

// Types that cross translation units (exceptions thrown in one and caught at the C ABI in another, members of the
// context) live in this named namespace; kernels, device helpers and small host helpers stay in each unit's own
// unnamed namespace.
namespace mmsbm_hip_impl {

// ======================================================================================
// errors
// ======================================================================================
struct ApiError : std::runtime_error {
  int code;
  ApiError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

}  // namespace mmsbm_hip_impl
using namespace mmsbm_hip_impl;

#define HIP_CHECK(expr)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      throw ApiError(MMSBM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

namespace {

constexpr double kEps = 2.220446049250313e-16;  // np.finfo(float).eps, src/kernels_numpy.py:51
constexpr int64_t kGpuLayoutMin = 100'000;       // triples from which the layout's sorts run on the device
constexpr int kBlock = 256;

// ======================================================================================
// device helpers
// ======================================================================================
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Sum over the G consecutive lanes of a group (G a power of two, groups aligned to G).
// Every step adds a value to its mirror image, so all lanes of a group end with the
// bitwise-identical sum.  Up to 16 lanes stay inside one DPP row (no LDS traffic).
template <int G>
__device__ __forceinline__ double group_sum(double x) {
  if (G >= 2) x += dpp_move<0xB1>(x);    // quad_perm [1,0,3,2]
  if (G >= 4) x += dpp_move<0x4E>(x);    // quad_perm [2,3,0,1]
  if (G >= 8) x += dpp_move<0x141>(x);   // row_half_mirror
  if (G >= 16) x += dpp_move<0x140>(x);  // row_mirror
  if (G >= 32) x += __shfl_xor(x, 16, 64);
  if (G >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

template <int VEC>
__device__ __forceinline__ void load_vec(const double *__restrict__ p, double (&v)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    const double2 t = *reinterpret_cast<const double2 *>(p + j);
    v[j] = t.x;
    v[j + 1] = t.y;
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(double *__restrict__ p, const double (&v)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    double2 t;
    t.x = v[j];
    t.y = v[j + 1];
    *reinterpret_cast<double2 *>(p + j) = t;
  }
}

// Output rows that the NEXT launch reads and this one never touches again (T, A, theta') can go out as non-temporal
// stores: their lines do not wait in an L2 for the end-of-kernel write-back.  `nt` is uniform (a kernel argument).
typedef double nt_dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_out2(double *p, const double2 v, bool nt) {
  if (nt) {
    nt_dbl2 t;
    t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<nt_dbl2 *>(p));
  } else {
    *reinterpret_cast<double2 *>(p) = v;
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec_out(double *__restrict__ p, const double (&v)[VEC], bool nt) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    double2 t;
    t.x = v[j];
    t.y = v[j + 1];
    store_out2(p + j, t, nt);
  }
}
// ... and a row that is read once (the segment's own row) can come in as a non-temporal load
template <int VEC>
__device__ __forceinline__ void load_vec_in(const double *__restrict__ p, double (&v)[VEC], bool nt) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    if (nt) {
      const nt_dbl2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_dbl2 *>(p + j));
      v[j] = t.x;
      v[j + 1] = t.y;
    } else {
      const double2 t = *reinterpret_cast<const double2 *>(p + j);
      v[j] = t.x;
      v[j + 1] = t.y;
    }
  }
}

}  // namespace

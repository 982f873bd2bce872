// layout_gpu.hpp -- the sort stage of the data layout ON the device (MI355X), for large inputs.
//
// layout.hpp sorts the triples on the host (threaded counting sorts): 14 ms at 1M ratings, but
// 0.22 s at 10M and 1.6 s at 100M -- more than the EM run it prepares (400 iterations at 100M
// ratings x K = L = 10 take 0.84 s).  The reference's own front door is worse (O(U*N) scans,
// src/mmsbm.py:100-122), which is why SURVEY 8(f) N3 lists it.  Here the same arrays are produced
// by the GPU from the three id columns the context uploads anyway:
//
//   key[n]  = ((rating * I + item) << 32) | user            -> one 64-bit radix sort = pair order
//   heads   = key's upper half changes                      -> inclusive scan = pair id per triple
//   (user, pair id) stable radix sort by user               -> user order
//   (item, pair id) stable radix sort by item over the pairs -> pairs of each item
//   offsets = lower bounds of 0..n in the sorted keys
//
// The radix sorts and the scan are rocPRIM device primitives (rocprim::radix_sort_keys / _pairs,
// rocprim::inclusive_scan: library calls for a library job -- this is set-up work outside the EM loop,
// not one of the hot kernels); the kernels around them are below.  The result is IDENTICAL to layout.hpp's (same orders, same tie rules: both
// sorts are stable), checked by tests on the GPU box.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "layout.hpp"

// (implemented in tu_layout.hip: the rocPRIM headers are compiled there and nowhere else)
namespace mmsbm {
namespace gpu_layout {

struct DeviceArrays {  // the two N-sized arrays stay on the device (ownership passes to the caller)
  int32_t *pair_user = nullptr, *user_pair = nullptr;
};

// d_user / d_item / d_rating: the validated id columns on the device (n_obs each).  Fills every
// host-side member of `L` that layout.hpp's sort stage fills EXCEPT pair_user and user_pair, which
// stay on the device (range_cuts() below serves the one host-side consumer they have).
// finish_layout() must follow.
void sort_stage(hipStream_t s, int64_t n_obs, int32_t n_users, int32_t n_items, int32_t n_ratings,
                const int32_t *d_user, const int32_t *d_item, const int32_t *d_rating,
                Layout &L, DeviceArrays &dev);

// cuts[s * (n_ranges + 1) + r] for the segments `off` (host copy) over the device index array d_idx
std::vector<int32_t> range_cuts(hipStream_t s, const std::vector<int32_t> &off, const int32_t *d_idx,
                                int32_t table_rows, int32_t n_ranges);

}  // namespace gpu_layout
}  // namespace mmsbm

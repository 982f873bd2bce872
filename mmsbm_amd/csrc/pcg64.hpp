// pcg64.hpp -- numpy's default bit generator (PCG64 = PCG XSL-RR 128/64, O'Neill 2014) with
// O(log n) jump-ahead, usable from host and device code.
//
// The reference draws a restart's initial theta, eta and p from
// ``np.random.default_rng(child_seed)`` in that order (src/mmsbm.py:224-233).  To initialise a
// restart ON the device with exactly the same numbers, thread t jumps the stream to its own
// offset and draws from there: the state after n steps of  s <- s * M + inc  is
// M^n s + inc (M^n - 1)/(M - 1)  (mod 2^128), evaluated by repeated squaring.
//
// numpy facts restated here (numpy/random/src/pcg64/pcg64.h, _common.pxd, distributions.c):
//   * multiplier M = 0x2360ED051FC65DA44385DF649FCCF645;
//   * next64 = step, then output of the NEW state: rotr64(hi ^ lo, hi >> 58);
//   * Generator.random() float64 = (next64 >> 11) * 2^-53, one 64-bit output per double.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PCG64_HD __host__ __device__ __forceinline__
#else
#define PCG64_HD inline
#endif

namespace pcg64 {

typedef unsigned __int128 u128;

PCG64_HD u128 make128(uint64_t hi, uint64_t lo) { return (static_cast<u128>(hi) << 64) | lo; }
PCG64_HD u128 multiplier() { return make128(2549297995355413924ULL, 4865540595714422341ULL); }

struct Stream {
  u128 state, inc;
};

PCG64_HD void step(Stream &g) { g.state = g.state * multiplier() + g.inc; }

// the stream `delta` draws further on
PCG64_HD void advance(Stream &g, uint64_t delta) {
  u128 acc_mult = 1, acc_plus = 0, cur_mult = multiplier(), cur_plus = g.inc;
  while (delta > 0) {
    if (delta & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta >>= 1;
  }
  g.state = acc_mult * g.state + acc_plus;
}

PCG64_HD uint64_t next64(Stream &g) {
  step(g);
  const uint64_t hi = static_cast<uint64_t>(g.state >> 64), lo = static_cast<uint64_t>(g.state);
  const uint64_t x = hi ^ lo;
  const unsigned rot = static_cast<unsigned>(hi >> 58);
  return (x >> rot) | (x << ((64 - rot) & 63));
}

PCG64_HD double next_double(Stream &g) {
  return static_cast<double>(next64(g) >> 11) * (1.0 / 9007199254740992.0);
}

}  // namespace pcg64

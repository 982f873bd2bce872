// rowtab.hpp -- RowTab: a table of rows that are gathered by index (theta, A), kept as whole cache lines + a compact tail
#pragma once

namespace {

// A table of rows that are GATHERED by index (theta, A).  A 160-byte row (K = 20) straddles two
// 128-byte cache lines; the table is therefore kept as a "main" part of whole 128-byte lines
// (mw = 16 * floor(Kp/16) doubles per row, line aligned) plus a compact "tail" part
// (tw = Kp - mw doubles per row), so a gather misses on one line of the big main part and
// hits the small, cache-resident tail part.  mw == row width and tw == 0 describes a plain table.
struct RowTab {
  double *main;      // row r, off < mw:  main + r * rs_m + off
  double *tail;      // row r, off >= mw: tail + r * rs_t + (off - mw)
  int mw, tw;        // widths of the two parts (tw == 0: a plain table)
  int rs_m, rs_t;    // row strides in doubles
  size_t so_m, so_t; // distance between the copies of two consecutive restart slots (see below)
};
// Restart slots.  The two GATHERED tables (theta, A) keep the slots' copies of a row side by side:
// row r = [slot 0 | slot 1 | ...], so rs_m = n_slots * mw, so_m = mw (and the same for the tail part).
// One index then serves every slot and a gather of row r for all slots is ONE contiguous piece of
// n_slots * 160 bytes at K = 20 -- whole 128-byte lines, no separate 32-byte tail access.  Streamed
// tables (C, T, eta) are plain per-slot copies: rs_m = width, so_m = the table's size.
__device__ __forceinline__ double *rowtab_ptr(const RowTab &t, size_t row, int off) {
  return off < t.mw ? t.main + row * t.rs_m + off : t.tail + row * t.rs_t + (off - t.mw);
}
__device__ __forceinline__ RowTab slot_tab(RowTab t, size_t slot) {
  t.main += slot * t.so_m;
  t.tail += slot * t.so_t;
  return t;
}

}  // namespace

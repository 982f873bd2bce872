// launch.hpp -- what every launching translation unit shares: the launch macros (with the optional launch log), views of
// the context's tables, and the stage entry points that live in other translation units
// Included by every translation unit of the library (prelude.hpp), after context.hpp.
#pragma once

// ---- stage entry points: each is defined in the translation unit that holds its kernels ----------------------------
namespace mmsbm_hip_impl {

// tu_seg.hip -- the two triple passes.  with_pairs / with_users select the segment sets of this launch (both: one
// launch, the pair segments' workgroups first); `st` is the stream it goes to.
void stage_seg(mmsbm_hip_ctx *c, bool commit, bool with_pairs, bool with_users, hipStream_t st);
// tu_pair.hip (vector ALUs) / tu_mfma.hip (matrix cores) -- T = P^T C and the K x L slabs for p
void stage_dense_valu(mmsbm_hip_ctx *c);
void stage_dense_mfma(mmsbm_hip_ctx *c);
// ... and A[q,:] from (eta, pT) of parameter buffer `slot` into atab[a_slot] -- or, with `grid` set, the same mat-vec
// over every (item, rating) combination into the plain table btab (prod_dist / predict)
void stage_matvec_a_valu(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid);
void stage_matvec_a_mfma(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid);
int mfma_a_blocks_per_cu(const mmsbm_hip_ctx *c);  // workgroups of the matrix-core A launch a CU holds (occupancy query)
// tu_etap.hip -- eta_new ; p_new, pT_new, raw n_p
void stage_eta_p(mmsbm_hip_ctx *c, bool commit);
// tu_fused.hip -- small problems: the iteration in two launches
void stage_fused_pairs(mmsbm_hip_ctx *c);
void stage_fused_tail(mmsbm_hip_ctx *c, bool commit);
// tu_once.hip -- once-per-run kernels.  likelihood of the selected slot (the caller holds a OneSlot): kernels onto the
// context's stream, no wait; returns the number of partial sums in lik_part
int likelihood_enqueue(mmsbm_hip_ctx *c);
void init_rows_launch(mmsbm_hip_ctx *c, const uint64_t pcg64_state[4]);
void omegas_launch(mmsbm_hip_ctx *c, double *dev_out, int64_t n_elems);
void prod_dist_launch(mmsbm_hip_ctx *c, const int32_t *du, const int32_t *di, double *dout, int64_t n_pairs);
// B = p_r eta_i over every (item, rating) combination, then one group of lanes per row.
// mode 0: dist[m][r] = P[m, r];  mode 1: dist += P, block_out = the restart's six sums per workgroup
int rows_launch(mmsbm_hip_ctx *c, int mode, const int32_t *pu, const int32_t *pi, const int32_t *preal,
                const double *weights, double *dist, double *block_out, int64_t n_rows, int first);
int score_rows_launch(mmsbm_hip_ctx *c, bool finish);  // the per-row scoring kernels; returns the number of workgroups
int score_stats_count();

// dispatchers (mmsbm_hip.hip): the form the context's shape and options select
void stage_dense(mmsbm_hip_ctx *c);
void stage_matvec_a(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid = false);
void ensure_a(mmsbm_hip_ctx *c);  // atab[cur] = A of the current parameters, for every slot the next launches cover

// ---- launch log (MMSBM_HIP_LAUNCH_LOG=<file>; mmsbm_hip.hip) ---------------------------------------------------------
// Host-side only: which kernel instantiations a process launched, how often, and under which test
// (MMSBM_HIP_LAUNCH_TAG at a kernel's first launch).  Appended to the file when a context is destroyed and at exit;
// scripts/kernel_coverage.py diffs it against the compiled set.  Nothing in the kernels knows about it.
extern bool g_launch_log_on;
void launch_log_note(const void *host_fn);

}  // namespace mmsbm_hip_impl

namespace {

template <class K>
inline void note_launch(K kernel) {
  if (g_launch_log_on) launch_log_note(reinterpret_cast<const void *>(kernel));
}

// One kernel launch.  KERNEL in parentheses when its template arguments hold commas.
#define LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                        \
  do {                                                                                                       \
    note_launch(KERNEL);                                                                                     \
    hipLaunchKernelGGL(KERNEL, GRID, dim3(BLOCK), static_cast<uint32_t>(LDS), STREAM, __VA_ARGS__);          \
  } while (0)
// ... of a stage: with the stage's event pair attached to the launch when the stage is timed as that kernel
// (LaunchScope::ext), else plain.
#define LAUNCH_IN(LS, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                  \
  do {                                                                                                       \
    note_launch(KERNEL);                                                                                     \
    if ((LS).ext())                                                                                          \
      hipExtLaunchKernelGGL(KERNEL, GRID, dim3(BLOCK), static_cast<uint32_t>(LDS), STREAM, (LS).e0, (LS).e1, 0, __VA_ARGS__); \
    else                                                                                                     \
      hipLaunchKernelGGL(KERNEL, GRID, dim3(BLOCK), static_cast<uint32_t>(LDS), STREAM, __VA_ARGS__);        \
  } while (0)

template <class K>
void allow_big_lds(K kernel, size_t bytes) {
  if (bytes > kLdsBudget)
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(bytes)));
}

// split (main/tail) tables -- see RowTab: theta and A are the gathered ones, eta/C/T stream
inline RowTab plain_tab(double *base, int width, size_t slot_stride = 0) {
  return RowTab{base, base, width, 0, width, 0, slot_stride, 0};
}
// A gathered table (theta, A) as seen from restart slot `slot`: the n_slots copies of every row are
// interleaved (RowTab), main parts of all rows first, then the tail parts.
inline RowTab gather_tab(const mmsbm_hip_ctx *c, double *base, size_t rows, int slot) {
  int mw = c->split_rows ? (c->kp / 16) * 16 : c->kp;  // (split_rows is always on today)
  if (mw == 0) mw = c->kp;
  const int tw = c->kp - mw, ns = c->n_slots;
  return RowTab{base + static_cast<size_t>(slot) * mw,
                base + rows * static_cast<size_t>(ns) * mw + static_cast<size_t>(slot) * tw,
                mw, tw, ns * mw, ns * tw, static_cast<size_t>(mw), static_cast<size_t>(tw)};
}
// (`b` = which of the two ping-pong buffers; the restart slot is c->base_slot)
inline RowTab theta_tab(const mmsbm_hip_ctx *c, int b) {
  return gather_tab(c, c->theta[b].ptr, static_cast<size_t>(c->n_users), c->base_slot);
}
inline RowTab a_tab(const mmsbm_hip_ctx *c, int b) {
  return gather_tab(c, c->atab[b].ptr, static_cast<size_t>(c->n_pairs), c->base_slot);
}
inline dim3 slot_grid(const mmsbm_hip_ctx *c, int blocks) {
  return dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(c->launch_slots), 1);
}
// Single-restart entry points: launches and copies cover the selected slot only.
struct OneSlot {
  mmsbm_hip_ctx *c;
  int b, n;
  explicit OneSlot(mmsbm_hip_ctx *ctx) : c(ctx), b(ctx->base_slot), n(ctx->launch_slots) {
    c->base_slot = c->sel;
    c->launch_slots = 1;
  }
  ~OneSlot() { c->base_slot = b; c->launch_slots = n; }
};

// Non-temporal stores for the rows the next launch gathers (T, A, theta'): one restart per launch and rows of up to
// 32 groups.  Measured per iteration, plain -> non-temporal (scripts/ab_fused.sh, variants side by side on one box): C1
// 11.3 -> 11.2 us, C2 20.3 -> 19.5, 600k ratings at K = L = 20 67.2 -> 66.0, C3 95.3 -> 93.8, 3M 302.3 -> 297.2, 10M
// 886 -> 880; with restart slots nothing or a loss (C3 x 2 165.8 -> 166.9, C3 x 8 591 -> 597), theta' at K = L = 50
// +0.3 %.  C rows, eta' and the slabs the same way: +0.3, +0.3 and +1.0 us at C3 -- they stay plain stores.  Loads:
// the segments' own rows in seg_pass as non-temporal loads C3 94.1 -> 93.5 (kept, same condition); the T rows in
// item_sum +1.6 us, the index stream +0.3, C rows in T + S and the slabs in p_update nothing.
// Only for data whose segments are all short and alike (no work lists on either side): with heavy-tailed degrees the
// rows of busy users and popular items are gathered again and again and a long segment's own row is read by every one
// of its pieces -- there every one of these hints costs (1M ratings, K = L = 20, per iteration, none / T and A rows /
// theta' rows / own-row loads / all: log-normal degrees 90.2 / 90.6 / 91.5 / 95.9 / 96.8 us, Zipf(1.2) 93.9 / 95.5 /
// 99.2 / 103.6 / 104.7; 100k ratings of 943 users: 28.2 / 29.2 / 28.2 / 28.4 / 29.4; scripts/nt_time.py).
inline int nt_on(const mmsbm_hip_ctx *c) {
  const bool plain_data = (c->lay.pair_work.items.empty() && c->lay.user_work.items.empty()) || (c->nt_out & 8);  // (8: tuning, whatever the data)
  return (plain_data && c->launch_slots == 1 && c->kp <= 32 && c->lp <= 32) ? (c->nt_out & 7) : 0;
}

inline bool mfma_possible(const mmsbm_hip_ctx *c) {
  return !c->wide && c->kp <= kMfmaMaxDim && c->lp <= kMfmaMaxDim && c->lds_mt <= kLdsMax && c->lds_ma <= kLdsMax;
}

}  // namespace

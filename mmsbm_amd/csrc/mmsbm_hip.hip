// mmsbm_hip.hip -- MI355X (gfx950 / CDNA4) EM core for the Mixed-Membership Stochastic
// Block Model, behind the C ABI of include/mmsbm_hip.h.
//
// What the reference computes per EM iteration (src/kernels_numpy.py:43-79 followed by
// src/mmsbm.py:248-250), for every observed triple n = (u, i, r):
//
//   omega[n,k,l] = theta[u,k] eta[i,l] p[k,l,r]        s_n = sum_kl omega[n,k,l]
//   n_theta[u,k] = sum_{n: u_n=u} sum_l omega/max(s_n,eps)      (and the same for eta, p)
//
// materialising the (N,K,L) tensor several times.  This file never builds omega.  With
// a "pair" q = a distinct (item, rating) combination and
//
//   A[q,k]   = sum_l p[k,l,r_q] eta[i_q,l]                       (pair_matvec, K-side)
//   w_n      = 1 / max(theta[u_n,:] . A[q_n,:], eps)
//   n_theta[u,k] = theta[u,k] * sum_{n in user u} A[q_n,k] w_n   (seg_pass, user segments)
//   C[q,k]   = sum_{n in pair q} theta[u_n,k] w_n                (seg_pass, pair segments)
//   T[q,l]   = sum_k p[k,l,r_q] C[q,k]                           (pair_matvec, L-side)
//   n_eta[i,l] = eta[i,l] * sum_{q in item i} T[q,l]             (item_sum)
//   n_p[k,l,r] = p[k,l,r] * sum_{q: r_q=r} C[q,k] eta[i_q,l]     (p_partial, p_update)
//
// which is the same sum re-associated: O(N K + Q K L) flops instead of O(N K L), two
// row-gather passes over the triples (sorted user-major and (rating,item)-major), every
// reduction a segmented reduction in registers (no atomics: results are bitwise
// reproducible), all arithmetic float64: the triple passes and the small-tile pair stage on the vector ALU,
// the pair stage of big rating tiles (K x L > 1024) on the matrix cores (v_mfma_f64_16x16x4_f64).
//
// Written for gfx950 only: 64-wide wavefronts, DPP cross-lane reductions, LDS-staged
// rating tiles.

#include "prelude.hpp"
#include "layout_gpu.hpp"
#include "pcg64.hpp"

#include <unistd.h>

#include <map>
#include <mutex>

namespace {

thread_local std::string g_last_error;

template <class F>
int guarded(F &&f) {
  try {
    f();
    return MMSBM_OK;
  } catch (const ApiError &e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::invalid_argument &e) {
    g_last_error = e.what();
    return MMSBM_E_INVALID;
  } catch (const std::bad_alloc &) {
    g_last_error = "host allocation failed";
    return MMSBM_E_INTERNAL;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return MMSBM_E_INTERNAL;
  } catch (...) {  // anything that is not a std::exception (a library's own type, a thrown int): a status, never a
                   // process death across the C ABI
    g_last_error = "unknown exception (not derived from std::exception)";
    return MMSBM_E_INTERNAL;
  }
}

}  // namespace

// ---- launch log (launch.hpp) -------------------------------------------------------------------------------------------
namespace mmsbm_hip_impl {

bool g_launch_log_on = std::getenv("MMSBM_HIP_LAUNCH_LOG") != nullptr && std::getenv("MMSBM_HIP_LAUNCH_LOG")[0] != 0;

namespace {
struct LaunchLog {
  struct Rec { long long count = 0; std::string name, tag; };
  std::mutex mu;
  std::map<const void *, Rec> seen;   // host-side kernel handle -> launches since the last flush, name, tag of the first one
  void note(const void *fn) {
    std::lock_guard<std::mutex> lock(mu);
    Rec &r = seen[fn];
    if (r.count++ == 0 && r.name.empty()) {
      // (the name is asked for HERE, with the runtime alive and the kernel about to be launched -- not at exit, when
      // the runtime may already be gone)
      const char *name = hipKernelNameRefByPtr(fn, nullptr);
      char buf[64];
      std::snprintf(buf, sizeof buf, "%p", fn);
      r.name = (name && name[0]) ? name : buf;
      const char *t = std::getenv("MMSBM_HIP_LAUNCH_TAG");
      r.tag = t ? t : "-";
    }
  }
  // one line per kernel: pid <tab> launches <tab> mangled name <tab> tag; appended in ONE write, so that the lines of
  // several processes sharing the file do not interleave.  Counts restart from zero, names and tags are kept.  No HIP
  // call in here: it also runs when the library is unloaded.
  void flush() {
    std::lock_guard<std::mutex> lock(mu);
    const char *path = std::getenv("MMSBM_HIP_LAUNCH_LOG");
    if (!path || !path[0]) return;
    std::string out;
    for (auto &kv : seen) {
      if (kv.second.count == 0) continue;
      out += std::to_string(static_cast<long long>(getpid())) + "\t" + std::to_string(kv.second.count) + "\t" +
             kv.second.name + "\t" + kv.second.tag + "\n";
      kv.second.count = 0;
    }
    if (out.empty()) return;
    if (FILE *f = std::fopen(path, "a")) {
      std::fwrite(out.data(), 1, out.size(), f);
      std::fclose(f);
    }
  }
  ~LaunchLog() { flush(); }
};
LaunchLog &launch_log() {
  static LaunchLog log;   // (flushed once more when the library is unloaded / the process exits)
  return log;
}
}  // namespace

void launch_log_note(const void *host_fn) { launch_log().note(host_fn); }

// ---- the form of the pair stage the context's shape and options select ------------------------------------------------
void stage_dense(mmsbm_hip_ctx *c) {  // T = P^T C  and the K x L slabs for p
  if (c->n_chunks == 0) return;
  if (c->mfma_big || (!c->wide && c->mfma)) stage_dense_mfma(c);
  else stage_dense_valu(c);
}
void stage_matvec_a(mmsbm_hip_ctx *c, int slot, int a_slot, bool grid) {
  if (c->mfma_big || (!c->wide && c->mfma)) stage_matvec_a_mfma(c, slot, a_slot, grid);
  else stage_matvec_a_valu(c, slot, a_slot, grid);
}

// atab[cur] = A of the current parameters, for every slot the next launches cover (see mmsbm_hip_ctx::a_ok)
void ensure_a(mmsbm_hip_ctx *c) {
  bool ok = true;
  for (int s = c->base_slot; s < c->base_slot + c->launch_slots; ++s) ok = ok && c->a_ok[static_cast<size_t>(s)];
  if (ok) return;
  const bool prof = c->profiling;  // (not a launch of the iteration being profiled)
  c->profiling = false;
  stage_matvec_a(c, c->cur, c->cur);
  c->profiling = prof;
  for (int s = c->base_slot; s < c->base_slot + c->launch_slots; ++s) c->a_ok[static_cast<size_t>(s)] = 1;
}

}  // namespace mmsbm_hip_impl

namespace {

// Units per workgroup for a launch whose workgroups each walk a run of 64-pair units of one rating, `slots` of them
// resident at a time: the launch lasts rounds x (units + a prologue of ~0.6 unit-times: the tile into LDS, the item ids,
// the first rows' latency -- measured at C5, EXPERIMENTS.md), and rounds is an INTEGER (with 768 slots C5's 15,616 units in
// runs of 8 are 1,952 workgroups = 2.54 rounds, paid as 3 = 24 unit-times; in runs of 11 they are 1.9 rounds, paid as 2 = 22).
// Workgroups are counted by the rule the run list is built by (layout.hpp: padded_chunk_count).
inline int balanced_run_units(const std::vector<int32_t> &rating_off, int slots, int lo, int hi) {
  int best = hi;
  double best_cost = 1e300;
  const bool align = mmsbm::chunk_align_on();
  const int n_ratings = static_cast<int>(rating_off.size()) - 1;
  for (int u = hi; u >= lo; --u) {
    long long wgs = 0;
    for (int r = 0; r < n_ratings; ++r)
      wgs += mmsbm::padded_chunk_count(rating_off[static_cast<size_t>(r) + 1] - rating_off[static_cast<size_t>(r)],
                                       u * kUnitPairs, n_ratings, align);
    const double cost = static_cast<double>((wgs + slots - 1) / std::max(slots, 1)) * (u + 0.6);
    if (cost < best_cost * 0.97) { best_cost = cost; best = u; }   // (longer runs win ties: fewer prologues)
  }
  return best;
}

// The A launch of the matrix-core pair stage writes rows only -- nothing ties its workgroups to the T + S launch's
// slabs -- so it walks the same units in runs of its own length: `units` 64-pair units per workgroup, or (0) as many as
// make its last round of workgroups (nearly) full (balanced_run_units; C5: runs of 11 units where the T + S launch takes
// 8).  Same rows, bit for bit.  n_a_chunks = 0: the T + S launch's own list serves.
void build_a_runs(mmsbm_hip_ctx *c, int units) {
  c->n_a_chunks = 0;
  c->a_chunks.release();
  if (!c->mfma || c->n_pairs <= 0) return;
  const int now = c->mv_chunk_pairs / kUnitPairs, most = kMfmaChunkPairs / kUnitPairs;
  int a_units = units > 0 ? std::min(units, most)
                          : balanced_run_units(c->lay.rating_off, mfma_a_blocks_per_cu(c) * c->n_cus, std::max(2, now / 2),
                                               std::min(2 * now, most));
  c->a_units = a_units;
  if (a_units == now) return;
  mmsbm::Layout tmp;   // (only the fields build_mv_chunks reads and writes)
  tmp.n_ratings = c->lay.n_ratings; tmp.rating_off = c->lay.rating_off;
  mmsbm::build_mv_chunks(tmp, a_units * kUnitPairs);
  c->a_chunks.upload(tmp.mv_chunks, c->stream);
  HIP_CHECK(hipStreamSynchronize(c->stream));   // (`tmp` is a local)
  c->n_a_chunks = static_cast<int>(tmp.mv_chunks.size());
}

// ---- small problems: the iteration in two launches (fused_small.hpp) -------------------------------------
bool fused_shape_ok(const mmsbm_hip_ctx *c) {  // (everything but the data: the kernels exist for this shape)
  // (rows of up to 24 groups: beyond that the four-launch pair stage runs 512-thread workgroups, whose split of a
  // unit's pairs among the copies of the slab grid -- hence the association order of S -- 256 threads cannot mirror)
  return c->code_k <= 1 && c->code_l <= 1 && c->pb_threads_t == kBlock && c->pb_threads_a == kBlock && !c->tl_t &&
         !c->tl_a && c->pb_nacc == 1 && c->pb_kt == 2 && c->pb_spb * c->pb_nsub <= kBlock && !c->wide &&
         !c->mfma && !c->mfma_big && !c->direct_out && c->mv_chunk_pairs == mmsbm::kMvChunkPairs && c->n_chunks > 0 &&
         pairs_fused_lds(c->kp, c->lp) <= kLdsBudget;
}
bool fused_possible(const mmsbm_hip_ctx *c) {
  // A work list that only ORDERS whole segments is fine (the pair units ignore it -- every segment's result is its
  // own -- and the user pass follows it as before); segments cut into pieces need all pieces of a segment inside one
  // workgroup: the lists create() builds for small problems (fs_pairs / fs_users; round 4)
  return fused_shape_ok(c) && (c->lay.pair_work.splits.empty() || c->fs_pairs) &&
         (c->lay.user_work.splits.empty() || c->fs_users);
}

void mark_a(mmsbm_hip_ctx *c, bool ok) {
  for (int s = c->base_slot; s < c->base_slot + c->launch_slots; ++s) c->a_ok[static_cast<size_t>(s)] = ok ? 1 : 0;
}

// Two launches while the launches' work is small: ratings x restart slots x (K + L, padded) <= 14M (measured per
// iteration, two / four launches.  One slot, K = L = 10: 100k ratings 20.2 / 29.3 us, 300k 28.7 / 37.0, 500k 37.8 /
// 43.5; K = L = 16: 400k 36.0 / 41.7; K = L = 20: 100k 28.4 / 37.6, 300k 50.1 / 50.1, 600k 69.4 / 67.5, 1M 105.0 / 95.6.  100k ratings at K = L = 10
// with 2 / 4 / 8 / 16 slots: 26.5 / 32.4, 40.9 / 42.8, 68.0 / 64.9, 119.9 / 115.9 -- with several slots the four-launch
// form shares the index stream among them)
// Data whose segments are cut into pieces pays a combine launch in the separate form (five launches): there the two
// launches win a little further out (log-normal popularity, separate / two, scripts/fused_split_sweep.py: 400k ratings
// K = L = 10 38.4 / 31.3 us, 700k 43.8 / 39.8; 300k K = L = 20 45.1 / 41.4, 600k 54.0 / 65.5; 1M x 100k x 20k K = L = 10
// 77.3 / 92.6) -- up to 18M.
constexpr long long kFusedWorkMax = 14000000, kFusedWorkMaxSplit = 18000000, kFusedRatingsMax = 1500000;
bool use_fused(const mmsbm_hip_ctx *c) {
  // (fused_possible again: options set after create() -- "mfma", "direct" ... -- change what it depends on)
  const bool cut = !c->lay.pair_work.splits.empty() || !c->lay.user_work.splits.empty();
  return c->fused && fused_possible(c) &&
         (c->fused_forced || c->n_obs * c->launch_slots * (c->kp + c->lp) <= (cut ? kFusedWorkMaxSplit : kFusedWorkMax));
}

void launch_iteration(mmsbm_hip_ctx *c, bool commit) {
  if (use_fused(c)) {
    stage_fused_pairs(c);  // (writes A of the current parameters on its way)
    stage_fused_tail(c, commit);
    if (commit) c->cur ^= 1;
    mark_a(c, !commit);
    return;
  }
  ensure_a(c);
  stage_seg(c, commit, true, true, c->stream);
  stage_dense(c);
  stage_eta_p(c, commit);
  if (commit) {
    stage_matvec_a(c, c->cur ^ 1, c->cur ^ 1);
    c->cur ^= 1;
    mark_a(c, true);
  }
}

// n committed iterations: graph replays of two iterations each when enabled, the rest eager
void run_iterations(mmsbm_hip_ctx *c, int n) {
  if (c->graph_mode && !c->profiling) {
    while (n >= 2) {
      const int slot = c->cur;
      if (!c->graph_exec[slot]) {
        if (!use_fused(c)) ensure_a(c);  // (outside the capture: a replay must not repeat it)
        hipGraph_t graph = nullptr;
        HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        try {
          launch_iteration(c, true);
          launch_iteration(c, true);
        } catch (...) {
          (void)hipStreamEndCapture(c->stream, &graph);
          if (graph) (void)hipGraphDestroy(graph);
          c->cur = slot;
          throw;
        }
        HIP_CHECK(hipStreamEndCapture(c->stream, &graph));
        hipError_t e = hipGraphInstantiate(&c->graph_exec[slot], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess)
          throw ApiError(MMSBM_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        // capture only records: cur is back where it started and nothing has run yet
      }
      HIP_CHECK(hipGraphLaunch(c->graph_exec[slot], c->stream));
      // (a replay runs none of launch_iteration's host code: the bookkeeping of "atab[cur] holds A of the current
      // parameters" has to be repeated here -- true after the four-launch form, false after the two-launch one)
      mark_a(c, !use_fused(c));
      n -= 2;
    }
  }
  for (; n > 0; --n) launch_iteration(c, true);
}

void require_params(const mmsbm_hip_ctx *c) {  // the selected slot
  if (!c) throw std::invalid_argument("null context");
  if (!c->have[c->sel]) throw std::invalid_argument("set_params has not been called");
}
void require_all_params(const mmsbm_hip_ctx *c) {  // every slot: the iteration advances all of them
  if (!c) throw std::invalid_argument("null context");
  for (int s = 0; s < c->n_slots; ++s)
    if (!c->have[s])
      throw std::invalid_argument(c->n_slots == 1 ? std::string("set_params has not been called")
                                                  : "set_params has not been called for slot " +
                                                        std::to_string(s));
}

// (Re)allocate the per-restart state for `slots` parameter sets; nothing is kept.
void alloc_state(mmsbm_hip_ctx *c, int slots) {
  hipStream_t s = c->stream;
  HIP_CHECK(hipStreamSynchronize(s));
  c->drop_graphs();
  const size_t klr = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  auto zeroed = [&](SlotBuf &b, size_t per_slot) {
    b.alloc_slots(per_slot, slots);
    HIP_CHECK(hipMemsetAsync(b.ptr, 0, sizeof(double) * std::max<size_t>(b.count, 1), s));
  };
  for (int b = 0; b < 2; ++b) {
    zeroed(c->theta[b], static_cast<size_t>(c->n_users) * c->kp);
    zeroed(c->eta[b], static_cast<size_t>(c->n_items) * c->lp);
    zeroed(c->p[b], klr);
    zeroed(c->pt[b], klr);
    zeroed(c->atab[b], static_cast<size_t>(c->n_pairs) * c->kp);
  }
  // (the scratch tables too: every entry a kernel reads is written by the launch before it -- audited, EXPERIMENTS.md
  // -- but padding rows and the slabs of empty units then hold zeros rather than whatever the pages held before)
  zeroed(c->ctab, static_cast<size_t>(c->n_pairs) * c->kp);
  zeroed(c->ttab, static_cast<size_t>(c->n_pairs) * c->lp);
  zeroed(c->partial, c->lay.mv_chunks.size() * c->kp * c->lp);
  zeroed(c->npr, klr);
  zeroed(c->pair_parts, static_cast<size_t>(c->lay.pair_work.n_parts) * c->kp);
  zeroed(c->user_parts, static_cast<size_t>(c->lay.user_work.n_parts) * c->kp);
  HIP_CHECK(hipStreamSynchronize(s));
  c->n_slots = slots;
  c->sel = 0;
  c->base_slot = 0;
  c->launch_slots = slots;
  c->cur = 0;
  c->have.assign(static_cast<size_t>(slots), 0);
  c->a_ok.assign(static_cast<size_t>(slots), 0);
}

// host (rows, d) row-major  <->  device RowTab (rows, dp) zero-padded, main + tail parts, staged
// through pinned memory in the one-slot layout (main rows, then tail rows): one contiguous copy when
// the context has one slot, a strided (2-D) copy per part when the slots' rows are interleaved
bool tab_is_packed(const RowTab &t, int rows) {
  return t.rs_m == t.mw && t.rs_t == t.tw && (t.tw == 0 || t.tail == t.main + static_cast<size_t>(rows) * t.mw);
}
void copy_rows(mmsbm_hip_ctx *c, const RowTab &t, double *stage, int rows, bool to_device) {
  const size_t e = sizeof(double);
  hipStream_t xs = c->xfer ? c->xfer : c->stream;
  if (rows == 0) return;
  if (tab_is_packed(t, rows)) {
    if (to_device)
      HIP_CHECK(hipMemcpyAsync(t.main, stage, e * rows * (t.mw + t.tw), hipMemcpyHostToDevice, xs));
    else
      HIP_CHECK(hipMemcpyAsync(stage, t.main, e * rows * (t.mw + t.tw), hipMemcpyDeviceToHost, xs));
    return;
  }
  double *stage_t = stage + static_cast<size_t>(rows) * t.mw;
  if (to_device) {
    HIP_CHECK(hipMemcpy2DAsync(t.main, e * t.rs_m, stage, e * t.mw, e * t.mw, rows, hipMemcpyHostToDevice, xs));
    if (t.tw > 0)
      HIP_CHECK(hipMemcpy2DAsync(t.tail, e * t.rs_t, stage_t, e * t.tw, e * t.tw, rows, hipMemcpyHostToDevice, xs));
  } else {
    HIP_CHECK(hipMemcpy2DAsync(stage, e * t.mw, t.main, e * t.rs_m, e * t.mw, rows, hipMemcpyDeviceToHost, xs));
    if (t.tw > 0)
      HIP_CHECK(hipMemcpy2DAsync(stage_t, e * t.tw, t.tail, e * t.rs_t, e * t.tw, rows, hipMemcpyDeviceToHost, xs));
  }
}
void zero_rows(mmsbm_hip_ctx *c, const RowTab &t, int rows) {
  const size_t e = sizeof(double);
  if (rows == 0) return;
  if (tab_is_packed(t, rows)) {
    HIP_CHECK(hipMemsetAsync(t.main, 0, e * rows * (t.mw + t.tw), c->stream));
    return;
  }
  HIP_CHECK(hipMemset2DAsync(t.main, e * t.rs_m, 0, e * t.mw, rows, c->stream));
  if (t.tw > 0) HIP_CHECK(hipMemset2DAsync(t.tail, e * t.rs_t, 0, e * t.tw, rows, c->stream));
}
void upload_rows(mmsbm_hip_ctx *c, const RowTab &t, const double *host, int rows, int d) {
  const int dp = t.mw + t.tw, mw = t.mw, tw = t.tw;
  double *stage = c->pin.take(static_cast<size_t>(rows) * dp);
  double *tail = stage + static_cast<size_t>(rows) * mw;
  const int wm = std::min(d, mw), wt = std::max(0, d - mw);
  for_row_blocks(rows, dp, [=](int a, int b) {
    for (int r = a; r < b; ++r) {
      const double *src = host + static_cast<size_t>(r) * d;
      double *m = stage + static_cast<size_t>(r) * mw;
      std::memcpy(m, src, sizeof(double) * wm);
      for (int j = wm; j < mw; ++j) m[j] = 0.0;
      if (tw > 0) {
        double *tl = tail + static_cast<size_t>(r) * tw;
        std::memcpy(tl, src + mw, sizeof(double) * wt);
        for (int j = wt; j < tw; ++j) tl[j] = 0.0;
      }
    }
  });
  copy_rows(c, t, stage, rows, true);
}
// enqueue the device -> pinned copy; unpack_rows after the stream has been synchronised
double *download_rows(mmsbm_hip_ctx *c, const RowTab &t, int rows) {
  const int dp = t.mw + t.tw;
  double *stage = c->pin.take(static_cast<size_t>(rows) * dp);
  copy_rows(c, t, stage, rows, false);
  return stage;
}
void unpack_rows(double *host, const double *stage, const RowTab &t, int rows, int d) {
  const int dp = t.mw + t.tw, mw = t.mw, tw = t.tw;
  const double *tail = stage + static_cast<size_t>(rows) * mw;
  const int wm = std::min(d, mw), wt = std::max(0, d - mw);
  for_row_blocks(rows, dp, [=](int a, int b) {
    for (int r = a; r < b; ++r) {
      double *dst = host + static_cast<size_t>(r) * d;
      std::memcpy(dst, stage + static_cast<size_t>(r) * mw, sizeof(double) * wm);
      if (tw > 0 && wt > 0) std::memcpy(dst + mw, tail + static_cast<size_t>(r) * tw, sizeof(double) * wt);
    }
  });
}
size_t rows_doubles(const mmsbm_hip_ctx *c) {  // staging for theta + eta + p + pT of one slot
  return static_cast<size_t>(c->n_users) * c->kp + static_cast<size_t>(c->n_items) * c->lp +
         2 * static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
}

// device p layout [R][kp][lp] (internal k, l)  <->  host pr (K, L, R) external
void p_host_to_dev(const mmsbm_hip_ctx *c, const double *pr, double *p, double *pt) {
  const int R = c->n_ratings, K = c->k, L = c->l, kp = c->kp, lp = c->lp;
  std::fill(p, p + static_cast<size_t>(R) * kp * lp, 0.0);
  std::fill(pt, pt + static_cast<size_t>(R) * kp * lp, 0.0);
  for (int k = 0; k < K; ++k)
    for (int l = 0; l < L; ++l)
      for (int r = 0; r < R; ++r) {
        // internal (k,l) == external (l,k) when swapped
        const size_t h = c->swapped ? (static_cast<size_t>(l) * c->ext_l + k) * R + r
                                    : (static_cast<size_t>(k) * c->ext_l + l) * R + r;
        const double v = pr[h];
        p[(static_cast<size_t>(r) * kp + k) * lp + l] = v;
        pt[(static_cast<size_t>(r) * lp + l) * kp + k] = v;
      }
}
void p_dev_to_host(const mmsbm_hip_ctx *c, const double *p, double *pr) {
  const int R = c->n_ratings, K = c->k, L = c->l, kp = c->kp, lp = c->lp;
  for (int k = 0; k < K; ++k)
    for (int l = 0; l < L; ++l)
      for (int r = 0; r < R; ++r) {
        const size_t h = c->swapped ? (static_cast<size_t>(l) * c->ext_l + k) * R + r
                                    : (static_cast<size_t>(k) * c->ext_l + l) * R + r;
        pr[h] = p[(static_cast<size_t>(r) * kp + k) * lp + l];
      }
}

// (theta, eta, p) tables of one slot -> host arrays in host layout; any output may be null
void fetch_params(mmsbm_hip_ctx *c, const RowTab &tt, const RowTab &et, const double *p_dev,
                  double *theta, double *eta, double *pr) {
  hipStream_t xs = c->xfer ? c->xfer : c->stream;
  HIP_CHECK(hipStreamSynchronize(xs));
  c->pin.reset(rows_doubles(c));
  const size_t klr = static_cast<size_t>(c->n_ratings) * c->kp * c->lp;
  const double *st = theta ? download_rows(c, tt, c->n_users) : nullptr;
  const double *se = eta ? download_rows(c, et, c->n_items) : nullptr;
  double *sp = nullptr;
  if (pr) {
    sp = c->pin.take(klr);
    HIP_CHECK(hipMemcpyAsync(sp, p_dev, sizeof(double) * klr, hipMemcpyDeviceToHost, xs));
  }
  HIP_CHECK(hipStreamSynchronize(xs));
  if (theta) unpack_rows(theta, st, tt, c->n_users, c->k);
  if (eta) unpack_rows(eta, se, et, c->n_items, c->l);
  if (pr) p_dev_to_host(c, sp, pr);
}

void collect_profile(mmsbm_hip_ctx *c, float *mean_us, int *launches, int n_iters) {
  std::vector<double> tot(K_COUNT, 0.0);
  std::vector<int> cnt(K_COUNT, 0);
  for (auto &pe : c->prof_events) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, pe.second.first, pe.second.second));
    tot[pe.first] += ms * 1000.0;
    cnt[pe.first]++;
    (void)hipEventDestroy(pe.second.first);
    (void)hipEventDestroy(pe.second.second);
  }
  c->prof_events.clear();
  for (int i = 0; i < K_COUNT; ++i) {
    mean_us[i] = cnt[i] ? static_cast<float>(tot[i] / cnt[i]) : 0.f;
    if (launches) launches[i] = n_iters > 0 ? cnt[i] / n_iters : 0;
  }
}

}  // namespace


// ======================================================================================
// C ABI
// ======================================================================================
extern "C" {

int mmsbm_hip_abi_version(void) { return MMSBM_HIP_ABI_VERSION; }

#ifndef MMSBM_BUILD_ID
#define MMSBM_BUILD_ID "unknown"
#endif
const char *mmsbm_hip_build_id(void) { return MMSBM_BUILD_ID; }

const char *mmsbm_hip_last_error(void) { return g_last_error.c_str(); }

int mmsbm_hip_device_count(int *count) {
  return guarded([&] {
    if (!count) throw std::invalid_argument("null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
      *count = 0;
      throw ApiError(MMSBM_E_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
  });
}

int mmsbm_hip_device_info(int device, char *name, int name_len, int *compute_units,
                          int64_t *global_mem_bytes) {
  return guarded([&] {
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
      std::snprintf(name, static_cast<size_t>(name_len), "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (global_mem_bytes) *global_mem_bytes = static_cast<int64_t>(prop.totalGlobalMem);
  });
}

int mmsbm_hip_device_pci(int device, char *bus_id, int bus_id_len) {
  return guarded([&] {
    if (!bus_id || bus_id_len < 16) throw std::invalid_argument("bus_id buffer of at least 16 bytes needed");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw ApiError(MMSBM_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    HIP_CHECK(hipDeviceGetPCIBusId(bus_id, bus_id_len, device));
  });
}

int mmsbm_hip_device_mem(int device, int64_t *free_bytes, int64_t *total_bytes) {
  return guarded([&] {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw ApiError(MMSBM_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    int prev = 0;
    HIP_CHECK(hipGetDevice(&prev));
    HIP_CHECK(hipSetDevice(device));
    size_t f = 0, t = 0;
    const hipError_t e = hipMemGetInfo(&f, &t);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) throw ApiError(MMSBM_E_HIP, std::string("hipMemGetInfo: ") + hipGetErrorString(e));
    if (free_bytes) *free_bytes = static_cast<int64_t>(f);
    if (total_bytes) *total_bytes = static_cast<int64_t>(t);
  });
}

int mmsbm_hip_create(int device, int64_t n_obs, int32_t n_users, int32_t n_items,
                     int32_t n_ratings, int32_t k_groups, int32_t l_groups,
                     const int32_t *user, const int32_t *item, const int32_t *rating,
                     int swap_sides, mmsbm_hip_ctx **out) {
  return guarded([&] {
    if (!out) throw std::invalid_argument("null out");
    *out = nullptr;
    if (k_groups <= 0 || l_groups <= 0) throw std::invalid_argument("K and L must be positive");
    if (static_cast<int64_t>(k_groups) * l_groups > (int64_t(1) << 26))
      throw ApiError(MMSBM_E_UNSUPPORTED, "K x L beyond 2^26 entries per rating tile");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw ApiError(MMSBM_E_NODEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");

    // MMSBM_HIP_TIMING=1: where the time of building a context goes (stderr)
    const bool timing = std::getenv("MMSBM_HIP_TIMING") != nullptr;
    auto clk = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[mmsbm_hip_create] %-28s %8.2f ms\n", what,
                   std::chrono::duration<double, std::milli>(now - clk).count());
      clk = now;
    };
    std::unique_ptr<mmsbm_hip_ctx> c(new mmsbm_hip_ctx());
    c->device = device;
    c->swapped = swap_sides > 0 || (swap_sides < 0 && n_users < n_items);
    c->n_obs = n_obs;
    c->ext_users = n_users; c->ext_items = n_items; c->ext_k = k_groups; c->ext_l = l_groups;
    c->n_ratings = n_ratings;
    const int32_t *iu = user, *ii = item;
    if (c->swapped) {
      c->n_users = n_items; c->n_items = n_users; c->k = l_groups; c->l = k_groups;
      iu = item; ii = user;
    } else {
      c->n_users = n_users; c->n_items = n_items; c->k = k_groups; c->l = l_groups;
    }
    c->kp = pad_dim(c->k); c->lp = pad_dim(c->l);
    c->code_k = group_code(c->kp); c->code_l = group_code(c->lp);
    {
      // four waves share the chunks of 4 outputs of a short row; long rows get up to 8 waves
      // (measured: 320 threads do not beat 256 at L = 20, 512 beat 256 by 15 % at L = 50)
      auto threads_for = [](int nch) { return nch <= 6 ? kBlock : kPairBlockMax; };
      c->pb_threads_t = threads_for(c->lp / 4);
      c->pb_threads_a = threads_for(c->kp / 4);
      const int nthr = c->pb_threads_t;
      c->pb_kt = ((c->kp / 2) * (c->lp / 4) <= kBlock / 2) ? 2 : 4;
      const int nslot = (c->kp / c->pb_kt) * (c->lp / 4);
      if (nslot <= nthr / 2) {
        c->pb_spb = nslot; c->pb_nacc = 1;
        const int room = (c->kp * (kUnitPairs + 1) + kUnitPairs * c->lp) /
                         (nslot * 4 * c->pb_kt);  // hand-over area
        c->pb_nsub = std::max(1, std::min(std::min(nthr / nslot, 8), 1 + room));
      }
      else { c->pb_spb = nthr; int n = 1; while (n * nthr < nslot) n *= 2; c->pb_nacc = n; }
    }
    // the rating tile sits in LDS when it is too big for the scalar cache -- unless that does not
    // fit beside the rows, then it is read through scalar loads after all (slower, but it runs)
    c->tl_t = tile_in_lds(c->kp, c->lp);
    c->tl_a = tile_in_lds(c->lp, c->kp);
    c->lds_t = pair_block_lds(c->kp, c->lp, c->tl_t);
    c->lds_a = pair_block_lds(c->lp, c->kp, c->tl_a);
    if (c->lds_t > kLdsMax) { c->tl_t = false; c->lds_t = pair_block_lds(c->kp, c->lp, false); }
    if (c->lds_a > kLdsMax) { c->tl_a = false; c->lds_a = pair_block_lds(c->lp, c->kp, false); }
    // still too large for the 64-pair LDS stage (roughly K + L > 300): the plain wide-row kernels
    c->wide = c->lds_t > kLdsMax || c->lds_a > kLdsMax || c->pb_nacc > 4 ||
              std::getenv("MMSBM_HIP_FORCE_WIDE") != nullptr;
    c->split_rows = true;
    // small problems leave most CUs a couple of workgroups: the triple passes are then bound by the rounds of
    // dependent gathers per segment, and eight rows in flight per group beat four (C1 16.8 -> 16.1 us, C2 29.7 ->
    // 29.1 us per iteration); at C3 four are better (95.3 vs 97.3 us)
    c->seg_batch = n_obs <= 300000 ? 8 : 4;

    lap("checks");
    // The sorts: on the host for small inputs (14 ms at 1M ratings), on the device beyond
    // kGpuLayoutMin triples (layout_gpu.hpp; MMSBM_HIP_GPU_LAYOUT=0/1 forces either).  The id columns
    // are uploaded first in both cases (the element-wise kernels keep them in the original order).
    HIP_CHECK(hipSetDevice(device));
    {  // (before anything is sized from it: chunk lengths, persistent grids)
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        c->n_cus = cus;
    }
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    bool gpu_layout = n_obs >= kGpuLayoutMin;
    if (const char *g = std::getenv("MMSBM_HIP_GPU_LAYOUT")) gpu_layout = std::atoi(g) != 0;
    // (the device sort packs (rating, item) into 31 bits; sparser key spaces stay on the host)
    if (static_cast<uint64_t>(n_ratings) * static_cast<uint64_t>(c->n_items) >= (uint64_t(1) << 31)) gpu_layout = false;
    mmsbm::gpu_layout::DeviceArrays dev_idx;
    if (gpu_layout) {
      mmsbm::validate_triples(n_obs, c->n_users, c->n_items, n_ratings, iu, ii, rating);
      lap("id checks");
    }
    {
      const size_t bytes = sizeof(int32_t) * static_cast<size_t>(n_obs);
      c->orig_u.alloc(n_obs); c->orig_i.alloc(n_obs); c->orig_r.alloc(n_obs);
      if (n_obs > 0 && gpu_layout) {  // (host layout: uploaded after its own id checks, below)
        HIP_CHECK(hipMemcpyAsync(c->orig_u.ptr, iu, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->orig_i.ptr, ii, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(c->orig_r.ptr, rating, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));  // the caller's buffers are free again
        lap("id upload");
      }
    }
    if (gpu_layout) {
      try {
        mmsbm::gpu_layout::sort_stage(c->stream, n_obs, c->n_users, c->n_items, n_ratings, c->orig_u.ptr,
                                      c->orig_i.ptr, c->orig_r.ptr, c->lay, dev_idx);
      } catch (const std::invalid_argument &) {
        throw;
      } catch (const std::exception &e) {
        throw ApiError(MMSBM_E_HIP, std::string("device layout: ") + e.what());
      } catch (...) {
        throw ApiError(MMSBM_E_INTERNAL, "device layout: unknown exception (not derived from std::exception)");
      }
      c->pair_user.ptr = dev_idx.pair_user; c->pair_user.count = static_cast<size_t>(n_obs);
      c->user_pair.ptr = dev_idx.user_pair; c->user_pair.count = static_cast<size_t>(n_obs);
      mmsbm::finish_layout(c->lay, 512);
      lap("device layout (sorts)");
    } else {
      mmsbm::build_layout(n_obs, c->n_users, c->n_items, n_ratings, iu, ii, rating, 512, c->lay);
      lap("host layout (sorts)");
    }
    // big K x L tiles: four 64-pair units per pair_block workgroup (4x fewer slabs to write + add)
    const std::vector<mmsbm::Chunk> units64 = c->lay.mv_chunks;  // likelihood_units_kernel: <= 64 pairs
    c->n_lik_units = static_cast<int>(units64.size());
    // ... and both launches on the matrix cores where the tile has left the scalar cache (pair_mfma_kernel),
    // with eight units per workgroup while that still leaves every CU a few rounds of workgroups (C5: T+S
    // 358 -> 342 us, half the slabs for eta_p: 123 -> 111 us; 768 or 1,024 pairs per workgroup are slower)
    c->lds_mt = pair_mfma_lds(c->kp, c->lp, true);
    c->lds_ma = pair_mfma_lds(c->lp, c->kp, false);
    c->mfma = mfma_possible(c.get()) && c->kp * c->lp > 1024 && std::getenv("MMSBM_HIP_NO_MFMA") == nullptr;
    // K or L beyond 64: the blocked matrix-core kernels take over from the lane-per-pair stage with its tile in
    // scalar loads and from the wide-row kernels
    // Skinny tiles too (a side below 16 groups, e.g. 600 x 5 or 3 x 1,024): three quarters of a 16-wide tile are
    // padding there, and it is still several times faster than the alternatives -- the wide-row kernels have one
    // thread per output column (8 of 256 threads busy at L = 5), the lane-per-pair stage streams a 38 KB tile
    // through the scalar cache.  1M ratings, T+S / A launch: 600 x 5 2,336 / 124 -> 380 / 115 us, 1,024 x 3
    // 5,822 / 152 -> 621 / 171, 8 x 520 471 / 927 -> 247 / 191, 3 x 1,024 470 / 3,219 -> 430 / 346, 300 x 8
    // 310 / 91 -> 197 / 53 (scripts/skinny_time.py, round 3).
    c->mfma_big = !c->mfma && c->kp * c->lp > 1024 && std::getenv("MMSBM_HIP_NO_MFMA") == nullptr;
    int big_chunk = 4 * mmsbm::kMvChunkPairs;
    if (c->mfma && c->lay.n_pairs >= 2 * big_chunk * 4 * c->n_cus) big_chunk *= 2;
    if (const char *e = std::getenv("MMSBM_HIP_MFMA_CHUNK")) big_chunk = std::min(std::max(std::atoi(e) / 64 * 64, 64), kMfmaChunkPairs);  // (tuning)
    if (c->wide) mmsbm::build_mv_chunks(c->lay, kWideChunkPairs);
    else if (c->kp * c->lp > 1024) mmsbm::build_mv_chunks(c->lay, big_chunk);
    c->mv_chunk_pairs = c->wide ? kWideChunkPairs : (c->kp * c->lp > 1024 ? big_chunk : mmsbm::kMvChunkPairs);
    // long rows: the mat-vec's outputs go to memory straight from registers (C5: -6 % on both
    // pair_block launches); short rows are cheaper transposed through LDS and copied out flat
    // (C3: direct stores cost +1.1 / +1.7 us)
    c->direct_out = c->kp * c->lp > 1024;
    // ... and the A launch as a persistent four-unit pipeline where the tile sits in LDS and
    // everything fits (C5: 312 -> 259 us)
    c->lds_qa = (static_cast<size_t>(kQuadUnits) * c->lp * (kUnitPairs + 1) + static_cast<size_t>(c->lp) * c->kp) *
                sizeof(double);
    c->quad_a = !c->wide && c->kp * c->lp > 1024 && c->tl_a && c->pb_threads_a == kPairBlockMax &&
                c->lds_qa <= kLdsMax - 2048 && c->lp <= kQuadMaxL && big_chunk == 4 * mmsbm::kMvChunkPairs;
    c->n_pairs = c->lay.n_pairs;
    c->n_chunks = static_cast<int>(c->lay.mv_chunks.size());
    // dense data: XCD-local work lists (layout.hpp) -- every segment cut at fixed borders of the
    // gathered index, each range's work on one XCD, whose L2 then holds that slice of the table
    if (c->kp > kMaxGroupRow) {  // rows beyond the widest group-of-lanes instantiation: seg_wide_kernel, whole segments
      c->lay.pair_work = mmsbm::WorkList();
      c->lay.user_work = mmsbm::WorkList();
    } else if (std::getenv("MMSBM_HIP_NO_RANGES") == nullptr) {
      const int per = kBlock / group_lanes(c->code_k);
      const size_t row_bytes = static_cast<size_t>(c->kp) * sizeof(double);
      const int64_t mean_p = c->n_pairs > 0 ? n_obs / c->n_pairs : 0, mean_u = n_obs / std::max(c->n_users, 1);
      int rp = mmsbm::range_count(static_cast<size_t>(c->n_users) * row_bytes, mean_p);   // pair pass gathers theta
      int ru = mmsbm::range_count(static_cast<size_t>(c->n_pairs) * row_bytes, mean_u);   // user pass gathers A
      // ... unless the table's hot rows (what one L2 keeps by itself) already take most of the gathers:
      // theta rows are gathered once per triple of that user, A rows once per triple of that pair
      // Where the rows one L2 keeps by itself already take half of the gathers (heavy-tailed gather
      // counts, or a table only a few times an L2) there is little left to win: measured +16 % (50M
      // ratings, log-normal item popularity, 8.5 MB table) and +24 % (Zipf(1.2) degrees) if cut anyway.
      const int64_t fit = static_cast<int64_t>(mmsbm::kRangeSliceBytes / row_bytes);
      if (rp > 1 && mmsbm::hot_fraction(c->lay.user_off, fit) > 0.5) rp = 1;
      if (ru > 1 && mmsbm::hot_fraction(c->lay.pair_off, fit) > 0.5) ru = 1;
      if (const char *f = std::getenv("MMSBM_HIP_RANGES")) {  // tuning: "pairs,users" forced range counts
        int a = 0, b = 0;
        if (std::sscanf(f, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 1 && a <= 512 && b <= 512) { rp = a; ru = b; }
      }
      // (device layout: the index arrays live on the device, so the borders are found there and only the
      // cut positions -- segments x (ranges + 1) integers -- come back)
      std::vector<int32_t> cuts_p, cuts_u;
      if (gpu_layout && rp > 1) cuts_p = mmsbm::gpu_layout::range_cuts(c->stream, c->lay.pair_off, c->pair_user.ptr, c->n_users, rp);
      if (gpu_layout && ru > 1) cuts_u = mmsbm::gpu_layout::range_cuts(c->stream, c->lay.user_off, c->user_pair.ptr, c->n_pairs, ru);
      if (rp > 1)
        mmsbm::build_worklist_ranges(c->lay.pair_off, c->lay.pair_user.data(), c->n_users, rp,
                                     mmsbm::item_length(n_obs, c->n_pairs), per, c->lay.pair_work,
                                     gpu_layout ? cuts_p.data() : nullptr);
      if (ru > 1)
        mmsbm::build_worklist_ranges(c->lay.user_off, c->lay.user_pair.data(), c->n_pairs, ru,
                                     mmsbm::item_length(n_obs, c->n_users), per, c->lay.user_work,
                                     gpu_layout ? cuts_u.data() : nullptr);
      c->ranges_pairs = rp;
      c->ranges_users = ru;
    }
    lap("xcd-local work lists");
    // Small problems with uneven degrees (round 4): segments are cut into pieces above; give every workgroup of the
    // two-launch form WHOLE segments (layout.hpp: FusedLists) so that the pieces' partial rows meet in its LDS instead
    // of in a combine launch.  Pair side: the 64-pair units are rebuilt with at most 64 work items each (both forms
    // of the iteration then use these units: their S sums associate the same way); a pair of more than 64 pieces, or a
    // user whose pieces' partial rows exceed the LDS budget, leaves the data with the separate launches.
    if (n_obs <= kFusedRatingsMax && fused_shape_ok(c.get()) && std::getenv("MMSBM_HIP_NO_FUSED_SPLIT") == nullptr &&
        (!c->lay.pair_work.splits.empty() || !c->lay.user_work.splits.empty())) {
      const int lanes = group_lanes(c->code_k);
      if (!c->lay.pair_work.splits.empty()) {
        const mmsbm::SegPieces sp = mmsbm::segment_pieces(c->lay.pair_off, c->lay.pair_work);
        // (the capped unit list REPLACES the plain one -- both forms of the iteration then run it, so that their S sums
        // associate the same way -- but only once it is certain that the two-launch form can use it: kept aside until then)
        const std::vector<mmsbm::Chunk> plain_chunks = c->lay.mv_chunks;
        const std::vector<int32_t> plain_off = c->lay.mv_chunk_off;
        if (mmsbm::build_mv_chunks_capped(c->lay, sp, mmsbm::kMvChunkPairs, kUnitPairs)) {
          const mmsbm::FusedLists fl = mmsbm::build_fused_pairs(c->lay, sp);
          if (pairs_fused_lds(c->kp, c->lp, fl.max_parts) <= kLdsMax) {
            c->fp_units.upload(fl.units, c->stream); c->fp_items.upload(fl.items, c->stream);
            c->fp_splits.upload(fl.splits, c->stream);
            HIP_CHECK(hipStreamSynchronize(c->stream));  // (`fl` is a local)
            c->fp_max_parts = fl.max_parts;
            c->fs_pairs = true;
          }
        }
        if (!c->fs_pairs) { c->lay.mv_chunks = plain_chunks; c->lay.mv_chunk_off = plain_off; }
        c->n_chunks = static_cast<int>(c->lay.mv_chunks.size());
      }
      if (!c->lay.user_work.splits.empty()) {
        const mmsbm::SegPieces sp = mmsbm::segment_pieces(c->lay.user_off, c->lay.user_work);
        int ucap = kBlock / lanes;  // work items per workgroup: one round of its groups of lanes
        if (const char *e = std::getenv("MMSBM_HIP_FUSED_UCAP")) ucap = std::max(1, std::atoi(e));  // (tuning)
        const mmsbm::FusedLists fl = mmsbm::build_fused_users(sp, ucap);
        if (static_cast<size_t>(fl.max_parts) * c->kp * sizeof(double) <= kFusedSplitLds) {
          c->fu_units.upload(fl.units, c->stream); c->fu_items.upload(fl.items, c->stream);
          c->fu_splits.upload(fl.splits, c->stream);
          HIP_CHECK(hipStreamSynchronize(c->stream));
          c->fu_max_parts = fl.max_parts;
          c->fu_blocks = static_cast<int>(fl.units.size());
          c->fs_users = true;
        }
      }
      lap("whole-segment lists (two launches)");
    }
    // small problems (where eight rows in flight pay, above): two launches per iteration instead of four
    c->fused = n_obs <= kFusedRatingsMax && fused_possible(c.get()) && std::getenv("MMSBM_HIP_NO_FUSED") == nullptr;

    hipStream_t s = c->stream;
    c->pair_off.upload(c->lay.pair_off, s);
    c->pair_item.upload(c->lay.pair_item, s);
    c->user_off.upload(c->lay.user_off, s);
    if (!gpu_layout) {  // (the device layout left these two where they were built)
      c->pair_user.upload(c->lay.pair_user, s);
      c->user_pair.upload(c->lay.user_pair, s);
    }
    c->item_off.upload(c->lay.item_off, s);
    c->item_pairs.upload(c->lay.item_pairs, s);
    c->item_deg.upload(c->lay.item_deg, s);
    // at least half of all (item, rating) combinations occur and R is small: fixed-width grid of pair ids
    if (n_ratings <= 16 && static_cast<int64_t>(c->n_pairs) * 2 >= static_cast<int64_t>(c->n_items) * n_ratings &&
        std::getenv("MMSBM_HIP_NO_ITEMGRID") == nullptr) {
      std::vector<int32_t> grid(static_cast<size_t>(c->n_items) * n_ratings, -1);
      for (int r = 0; r < n_ratings; ++r)
        for (int32_t q = c->lay.rating_off[static_cast<size_t>(r)]; q < c->lay.rating_off[static_cast<size_t>(r) + 1]; ++q)
          grid[static_cast<size_t>(c->lay.pair_item[static_cast<size_t>(q)]) * n_ratings + r] = q;
      c->item_grid.upload(grid, s);
      HIP_CHECK(hipStreamSynchronize(s));  // `grid` is a local
    }
    c->mv_chunks.upload(c->lay.mv_chunks, s);
    build_a_runs(c.get(), 0);
    c->lik_units.upload(units64, s);
    c->mv_chunk_off.upload(c->lay.mv_chunk_off, s);
    c->pair_items.upload(c->lay.pair_work.items, s);
    c->user_items.upload(c->lay.user_work.items, s);
    c->pair_splits.upload(c->lay.pair_work.splits, s);
    c->user_splits.upload(c->lay.user_work.splits, s);
    if (!gpu_layout && n_obs > 0) {
      const size_t bytes = sizeof(int32_t) * static_cast<size_t>(n_obs);
      HIP_CHECK(hipMemcpyAsync(c->orig_u.ptr, iu, bytes, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(c->orig_i.ptr, ii, bytes, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(c->orig_r.ptr, rating, bytes, hipMemcpyHostToDevice, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));  // nothing of the caller's (or this function's) host memory is still being read
    lap("index uploads");
    alloc_state(c.get(), 1);
    c->lik_part.alloc(4096);
    HIP_CHECK(hipStreamSynchronize(s));
    lap("state allocation");
    *out = c.release();
  });
}

int mmsbm_hip_destroy(mmsbm_hip_ctx *ctx) {
  return guarded([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (g_launch_log_on) launch_log().flush();
    delete ctx;
  });
}

int mmsbm_hip_dims(const mmsbm_hip_ctx *ctx, int64_t dims[8]) {
  return guarded([&] {
    if (!ctx || !dims) throw std::invalid_argument("null argument");
    dims[0] = ctx->n_obs; dims[1] = ctx->ext_users; dims[2] = ctx->ext_items;
    dims[3] = ctx->n_ratings; dims[4] = ctx->ext_k; dims[5] = ctx->ext_l;
    dims[6] = ctx->n_pairs; dims[7] = ctx->swapped ? 1 : 0;
  });
}

int mmsbm_hip_degrees(const mmsbm_hip_ctx *ctx, int64_t *d_user, int64_t *d_item) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    const mmsbm::Layout &L = ctx->lay;
    int64_t *du = ctx->swapped ? d_item : d_user;  // internal users
    int64_t *di = ctx->swapped ? d_user : d_item;  // internal items
    if (du)
      for (int u = 0; u < L.n_users; ++u)
        du[u] = std::max<int64_t>(L.user_off[u + 1] - L.user_off[u], 1);
    if (di)
      for (int i = 0; i < L.n_items; ++i) di[i] = std::max<int64_t>(L.item_deg[i], 1);
  });
}

int mmsbm_hip_set_params(mmsbm_hip_ctx *ctx, const double *theta, const double *eta,
                         const double *pr) {
  return guarded([&] {
    if (!ctx || !theta || !eta || !pr) throw std::invalid_argument("null argument");
    use_device(ctx);
    OneSlot one(ctx);
    const double *it = ctx->swapped ? eta : theta;  // internal theta rows = internal users
    const double *ie = ctx->swapped ? theta : eta;
    const int cur = ctx->cur, sl = ctx->sel;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // nothing in flight still reads the staging area
    ctx->pin.reset(rows_doubles(ctx));
    upload_rows(ctx, theta_tab(ctx, cur), it, ctx->n_users, ctx->k);
    upload_rows(ctx, plain_tab(ctx->eta[cur].at(sl), ctx->lp), ie, ctx->n_items, ctx->l);
    const size_t klr = static_cast<size_t>(ctx->n_ratings) * ctx->kp * ctx->lp;
    double *p = ctx->pin.take(klr), *pt = ctx->pin.take(klr);
    p_host_to_dev(ctx, pr, p, pt);
    HIP_CHECK(hipMemcpyAsync(ctx->p[cur].at(sl), p, sizeof(double) * klr, hipMemcpyHostToDevice,
                             ctx->stream));
    HIP_CHECK(hipMemcpyAsync(ctx->pt[cur].at(sl), pt, sizeof(double) * klr, hipMemcpyHostToDevice,
                             ctx->stream));
    stage_matvec_a(ctx, cur, cur);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the staging area is free again
    ctx->have[sl] = 1;
    ctx->a_ok[sl] = 1;
  });
}

int mmsbm_hip_init_params(mmsbm_hip_ctx *ctx, const uint64_t pcg64_state[4], const double *pr) {
  return guarded([&] {
    if (!ctx || !pcg64_state || !pr) throw std::invalid_argument("null argument");
    use_device(ctx);
    OneSlot one(ctx);
    const int cur = ctx->cur, sl = ctx->sel;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const size_t klr = static_cast<size_t>(ctx->n_ratings) * ctx->kp * ctx->lp;
    ctx->pin.reset(2 * klr);
    double *p = ctx->pin.take(klr), *pt = ctx->pin.take(klr);
    p_host_to_dev(ctx, pr, p, pt);
    HIP_CHECK(hipMemcpyAsync(ctx->p[cur].at(sl), p, sizeof(double) * klr, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(ctx->pt[cur].at(sl), pt, sizeof(double) * klr, hipMemcpyHostToDevice, ctx->stream));
    zero_rows(ctx, theta_tab(ctx, cur), ctx->n_users);  // padding columns
    zero_rows(ctx, plain_tab(ctx->eta[cur].at(sl), ctx->lp), ctx->n_items);
    init_rows_launch(ctx, pcg64_state);
    stage_matvec_a(ctx, cur, cur);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->have[sl] = 1;
    ctx->a_ok[sl] = 1;
  });
}

int mmsbm_hip_pcg64_doubles(const uint64_t pcg64_state[4], uint64_t offset, int64_t n, double *out) {
  return guarded([&] {
    if (!pcg64_state || (n > 0 && !out) || n < 0) throw std::invalid_argument("bad argument");
    pcg64::Stream g{pcg64::make128(pcg64_state[0], pcg64_state[1]), pcg64::make128(pcg64_state[2], pcg64_state[3])};
    pcg64::advance(g, offset);
    for (int64_t j = 0; j < n; ++j) out[j] = pcg64::next_double(g);
  });
}

int mmsbm_hip_get_params(mmsbm_hip_ctx *ctx, double *theta, double *eta, double *pr) {
  return guarded([&] {
    require_params(ctx);
    use_device(ctx);
    OneSlot one(ctx);
    double *it = ctx->swapped ? eta : theta;
    double *ie = ctx->swapped ? theta : eta;
    const int cur = ctx->cur, sl = ctx->sel;
    fetch_params(ctx, theta_tab(ctx, cur), plain_tab(ctx->eta[cur].at(sl), ctx->lp),
                 ctx->p[cur].at(sl), it, ie, pr);
  });
}

int mmsbm_hip_set_slots(mmsbm_hip_ctx *ctx, int n_slots) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (n_slots < 1 || n_slots > 65535) throw std::invalid_argument("n_slots must be in [1, 65535]");
    use_device(ctx);
    if (n_slots == ctx->n_slots) {
      ctx->have.assign(static_cast<size_t>(n_slots), 0);
      ctx->a_ok.assign(static_cast<size_t>(n_slots), 0);
      ctx->sel = 0;
      return;
    }
    try {
      alloc_state(ctx, n_slots);
    } catch (...) {  // most likely out of device memory: leave a consistent one-slot context
      (void)hipGetLastError();
      try { alloc_state(ctx, 1); } catch (...) {}
      throw;
    }
  });
}

int mmsbm_hip_select_slot(mmsbm_hip_ctx *ctx, int slot) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (slot < 0 || slot >= ctx->n_slots)
      throw std::invalid_argument("slot " + std::to_string(slot) + " out of range (the context has " +
                                  std::to_string(ctx->n_slots) + ")");
    ctx->sel = slot;
  });
}

int mmsbm_hip_slots(const mmsbm_hip_ctx *ctx, int *n_slots, int *selected,
                    int64_t *bytes_per_slot) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (n_slots) *n_slots = ctx->n_slots;
    if (selected) *selected = ctx->sel;
    if (bytes_per_slot) {
      size_t d = ctx->ctab.stride + ctx->ttab.stride + ctx->partial.stride + ctx->npr.stride +
                 ctx->pair_parts.stride + ctx->user_parts.stride;
      for (int b = 0; b < 2; ++b)
        d += ctx->theta[b].stride + ctx->eta[b].stride + ctx->p[b].stride + ctx->pt[b].stride +
             ctx->atab[b].stride;
      *bytes_per_slot = static_cast<int64_t>(d * sizeof(double));
    }
  });
}

int mmsbm_hip_em_iterate(mmsbm_hip_ctx *ctx, int n_iters) {
  return guarded([&] {
    require_all_params(ctx);
    if (n_iters < 0) throw std::invalid_argument("n_iters must be >= 0");
    use_device(ctx);
    run_iterations(ctx, n_iters);
  });
}

int mmsbm_hip_synchronize(mmsbm_hip_ctx *ctx) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    use_device(ctx);
    // Short waits are polled (hipStreamQuery returns within a microsecond or two of the last kernel;
    // a blocking wait is woken by an interrupt tens of microseconds later, which is several percent
    // of a 2 ms run of 20 iterations); after 50 ms the thread blocks like any other waiter.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(ctx->stream);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) HIP_CHECK(e);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}

int mmsbm_hip_update_coefficients(mmsbm_hip_ctx *ctx, double *n_theta, double *n_eta,
                                  double *n_pr) {
  return guarded([&] {
    require_params(ctx);
    use_device(ctx);
    OneSlot one(ctx);
    launch_iteration(ctx, false);
    const int nxt = ctx->cur ^ 1, sl = ctx->sel;
    double *it = ctx->swapped ? n_eta : n_theta;
    double *ie = ctx->swapped ? n_theta : n_eta;
    fetch_params(ctx, theta_tab(ctx, nxt), plain_tab(ctx->eta[nxt].at(sl), ctx->lp),
                 ctx->npr.at(sl), it, ie, n_pr);
  });
}

namespace {
double likelihood_finish(mmsbm_hip_ctx *ctx, int nb) {  // workgroup sums added in workgroup order
  std::vector<double> part(nb);
  HIP_CHECK(hipMemcpyAsync(part.data(), ctx->lik_part.ptr, sizeof(double) * nb, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  for (double v : part) tot += v;
  return tot;
}
}  // namespace

int mmsbm_hip_likelihood(mmsbm_hip_ctx *ctx, double *out) {
  return guarded([&] {
    require_params(ctx);
    if (!out) throw std::invalid_argument("null out");
    use_device(ctx);
    OneSlot one(ctx);
    *out = likelihood_finish(ctx, likelihood_enqueue(ctx));
  });
}

int mmsbm_hip_result(mmsbm_hip_ctx *ctx, double *theta, double *eta, double *pr, double *likelihood) {
  return guarded([&] {
    require_params(ctx);
    if (!likelihood) throw std::invalid_argument("null likelihood");
    use_device(ctx);
    OneSlot one(ctx);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the parameters are final
    const int nb = likelihood_enqueue(ctx);        // the likelihood kernels run ...
    struct Xfer {                                   // ... while the tables travel on the copy stream and are unpacked
      mmsbm_hip_ctx *c;
      explicit Xfer(mmsbm_hip_ctx *x) : c(x) { c->xfer = c->copy_stream; }
      ~Xfer() { c->xfer = nullptr; }
    };
    {
      Xfer on(ctx);
      double *it = ctx->swapped ? eta : theta;
      double *ie = ctx->swapped ? theta : eta;
      fetch_params(ctx, theta_tab(ctx, ctx->cur), plain_tab(ctx->eta[ctx->cur].at(ctx->sel), ctx->lp),
                   ctx->p[ctx->cur].at(ctx->sel), it, ie, pr);
    }
    *likelihood = likelihood_finish(ctx, nb);
  });
}

int mmsbm_hip_compute_omegas(mmsbm_hip_ctx *ctx, double *out, int64_t capacity_elems) {
  return guarded([&] {
    require_params(ctx);
    if (!out) throw std::invalid_argument("null out");
    use_device(ctx);
    const int64_t kl = static_cast<int64_t>(ctx->k) * ctx->l;
    const int64_t n_elems = ctx->n_obs * kl;
    if (n_elems > capacity_elems)
      throw ApiError(MMSBM_E_TOOLARGE, "omega tensor larger than the caller's buffer");
    if (n_elems > (int64_t(1) << 31))  // 16 GiB: the factorised path exists so nobody needs this
      throw ApiError(MMSBM_E_TOOLARGE, "omega tensor above the 2^31-element cap");
    if (n_elems == 0) return;
    DevBuf<double> dev;
    dev.alloc(static_cast<size_t>(n_elems));
    OneSlot one(ctx);
    omegas_launch(ctx, dev.ptr, n_elems);
    HIP_CHECK(hipMemcpyAsync(out, dev.ptr, sizeof(double) * n_elems, hipMemcpyDeviceToHost,
                             ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
  });
}

namespace {
// ---- prod_dist / predict through B = p_r eta_i over every (item, rating) combination ---------------
// Worth it when the rows to score are not far fewer than the items (B costs I R K L multiply-adds, a row
// then R K instead of R K L) and B fits comfortably: otherwise the per-row kernels run.
bool rows_fast_ok(const mmsbm_hip_ctx *c, int64_t n_rows) {
  if (!c->predict_fast || n_rows <= 0) return false;
  const uint64_t combos = static_cast<uint64_t>(c->n_items) * static_cast<uint64_t>(c->n_ratings);
  if (combos >= (uint64_t(1) << 31) - 2048 || n_rows * 4 < c->n_items) return false;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  return combos * static_cast<uint64_t>(c->kp) * sizeof(double) <= (free_b + c->btab.count * sizeof(double)) / 2;
}
void ensure_pair_grid(mmsbm_hip_ctx *c) {  // q = r * I + i, chunks of the pair stage's size per rating
  const size_t combos = static_cast<size_t>(c->n_items) * c->n_ratings;
  if (c->grid_item.count != combos || c->grid_n_chunks == 0) {
    std::vector<int32_t> item(combos);
    std::vector<mmsbm::Chunk> chunks;
    for (int r = 0; r < c->n_ratings; ++r) {
      const int32_t base = static_cast<int32_t>(static_cast<size_t>(r) * c->n_items);
      for (int i = 0; i < c->n_items; ++i) item[static_cast<size_t>(base) + i] = i;
      for (int i = 0; i < c->n_items; i += c->mv_chunk_pairs)
        chunks.push_back(mmsbm::Chunk{r, base + i, base + std::min(i + c->mv_chunk_pairs, c->n_items), 0});
    }
    c->grid_item.upload(item, c->stream);
    c->grid_chunks.upload(chunks, c->stream);
    HIP_CHECK(hipStreamSynchronize(c->stream));  // (host vectors are locals)
    c->grid_n_chunks = static_cast<int>(chunks.size());
  }
  if (c->btab.count < combos * c->kp) c->btab.alloc(combos * c->kp);
}
// eligibility + the table's memory; false (and no error left behind) sends the caller to the per-row kernels
bool rows_fast_prepare(mmsbm_hip_ctx *c, int64_t n_rows) {
  if (!rows_fast_ok(c, n_rows)) return false;
  try {
    ensure_pair_grid(c);
  } catch (const ApiError &) {  // out of memory after all (another context took it): not an error of this call
    (void)hipGetLastError();
    c->btab.release();
    return false;
  }
  return true;
}
}  // namespace

int mmsbm_hip_prod_dist(mmsbm_hip_ctx *ctx, int64_t n_pairs, const int32_t *user,
                        const int32_t *item, double *out) {
  return guarded([&] {
    require_params(ctx);
    if (n_pairs < 0) throw std::invalid_argument("negative n_pairs");
    if (n_pairs == 0) return;
    if (!user || !item || !out) throw std::invalid_argument("null argument");
    for (int64_t m = 0; m < n_pairs; ++m)
      if (user[m] < 0 || user[m] >= ctx->ext_users || item[m] < 0 || item[m] >= ctx->ext_items)
        throw std::invalid_argument("prod_dist: id out of range at row " + std::to_string(m));
    use_device(ctx);
    const int64_t n_elems = n_pairs * ctx->n_ratings;
    if ((n_elems + kBlock - 1) / kBlock > (int64_t(1) << 31) - 1)
      throw ApiError(MMSBM_E_TOOLARGE, "prod_dist: too many pairs for one launch");
    DevBuf<int32_t> du, di;
    DevBuf<double> dout;
    du.alloc(n_pairs); di.alloc(n_pairs); dout.alloc(n_elems);
    const int32_t *iu = ctx->swapped ? item : user;
    const int32_t *ii = ctx->swapped ? user : item;
    HIP_CHECK(hipMemcpyAsync(du.ptr, iu, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(di.ptr, ii, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    OneSlot one(ctx);
    if (rows_fast_prepare(ctx, n_pairs)) rows_launch(ctx, 0, du.ptr, di.ptr, nullptr, nullptr, dout.ptr, nullptr, n_pairs, 1);
    else prod_dist_launch(ctx, du.ptr, di.ptr, dout.ptr, n_pairs);
    HIP_CHECK(hipMemcpyAsync(out, dout.ptr, sizeof(double) * n_elems, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->btab.release();  // (the B table is scratch: restart slots are sized from the free memory)
  });
}

namespace {
void score_launch(mmsbm_hip_ctx *ctx, bool finish, double *stats) {
  const int n_stats = score_stats_count();
  const bool fast = !finish && rows_fast_prepare(ctx, ctx->ps_rows);
  const int per_block = fast ? kBlock / group_lanes(ctx->code_k) : kBlock;
  const int64_t nb64 = (ctx->ps_rows + per_block - 1) / per_block;
  const int nb = static_cast<int>(nb64);
  if (ctx->ps_part.count < static_cast<size_t>(nb) * n_stats) ctx->ps_part.alloc(static_cast<size_t>(nb) * n_stats);
  if (nb > 0 && fast)
    rows_launch(ctx, 1, ctx->ps_u.ptr, ctx->ps_i.ptr, ctx->ps_r.ptr, ctx->ps_w.ptr, ctx->ps_sum.ptr, ctx->ps_part.ptr,
                ctx->ps_rows, ctx->ps_added == 0 ? 1 : 0);
  else if (nb > 0)
    score_rows_launch(ctx, finish);
  std::vector<double> part(static_cast<size_t>(nb) * n_stats);
  if (nb > 0)
    HIP_CHECK(hipMemcpyAsync(part.data(), ctx->ps_part.ptr, sizeof(double) * part.size(),
                             hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (int j = 0; j < n_stats; ++j) stats[j] = 0.0;
  for (int b = 0; b < nb; ++b)
    for (int j = 0; j < n_stats; ++j) stats[j] += part[static_cast<size_t>(b) * n_stats + j];
}
}  // namespace

int mmsbm_hip_predict_begin(mmsbm_hip_ctx *ctx, int64_t n_rows, const int32_t *user,
                            const int32_t *item, const int32_t *rating,
                            const double *rating_weights) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    if (n_rows < 0) throw std::invalid_argument("negative n_rows");
    if (!rating_weights || (n_rows > 0 && (!user || !item || !rating)))
      throw std::invalid_argument("null argument");
    if (n_rows > (int64_t(1) << 31) - kBlock)
      throw ApiError(MMSBM_E_TOOLARGE, "predict: too many rows for one launch");
    for (int64_t m = 0; m < n_rows; ++m)
      if (user[m] < 0 || user[m] >= ctx->ext_users || item[m] < 0 || item[m] >= ctx->ext_items ||
          rating[m] < 0 || rating[m] >= ctx->n_ratings)
        throw std::invalid_argument("predict: id out of range at row " + std::to_string(m));
    use_device(ctx);  // (arguments are fine: from here on the previous session is gone)
    ctx->ps_rows = -1;
    const int32_t *iu = ctx->swapped ? item : user;
    const int32_t *ii = ctx->swapped ? user : item;
    hipStream_t s = ctx->stream;
    ctx->ps_u.alloc(n_rows); ctx->ps_i.alloc(n_rows); ctx->ps_r.alloc(n_rows);
    ctx->ps_sum.alloc(static_cast<size_t>(n_rows) * ctx->n_ratings);
    ctx->ps_w.alloc(ctx->n_ratings);
    if (n_rows > 0) {
      HIP_CHECK(hipMemcpyAsync(ctx->ps_u.ptr, iu, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(ctx->ps_i.ptr, ii, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(ctx->ps_r.ptr, rating, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
    }
    HIP_CHECK(hipMemcpyAsync(ctx->ps_w.ptr, rating_weights, sizeof(double) * ctx->n_ratings,
                             hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));  // the caller's buffers are free again
    ctx->ps_rows = n_rows;
    ctx->ps_added = 0;
  });
}

int mmsbm_hip_predict_add(mmsbm_hip_ctx *ctx, double stats[6]) {
  return guarded([&] {
    require_params(ctx);
    if (!stats) throw std::invalid_argument("null stats");
    if (ctx->ps_rows < 0) throw std::invalid_argument("predict_begin has not been called");
    use_device(ctx);
    OneSlot one(ctx);
    score_launch(ctx, false, stats);
    ctx->ps_added++;
  });
}

int mmsbm_hip_predict_finish(mmsbm_hip_ctx *ctx, double *mean_dist, double stats[6]) {
  return guarded([&] {
    if (!ctx || !stats) throw std::invalid_argument("null argument");
    if (ctx->ps_rows < 0) throw std::invalid_argument("predict_begin has not been called");
    if (ctx->ps_added < 1) throw std::invalid_argument("predict_finish before any predict_add");
    use_device(ctx);
    OneSlot one(ctx);
    const int64_t rows = ctx->ps_rows;
    score_launch(ctx, true, stats);
    ctx->ps_rows = -1;  // the session is over whatever happens next
    ctx->btab.release();
    if (mean_dist && rows > 0) {
      HIP_CHECK(hipMemcpyAsync(mean_dist, ctx->ps_sum.ptr, sizeof(double) * rows * ctx->n_ratings,
                               hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
  });
}

int mmsbm_hip_time_iterations(mmsbm_hip_ctx *ctx, int n_iters, float *elapsed_ms) {
  return guarded([&] {
    require_all_params(ctx);
    if (!elapsed_ms || n_iters < 0) throw std::invalid_argument("bad argument");
    use_device(ctx);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, ctx->stream));
    run_iterations(ctx, n_iters);
    HIP_CHECK(hipEventRecord(e1, ctx->stream));
    HIP_CHECK(hipEventSynchronize(e1));
    HIP_CHECK(hipEventElapsedTime(elapsed_ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

int mmsbm_hip_kernel_count(void) { return K_COUNT; }

const char *mmsbm_hip_kernel_name(int index) {
  return (index >= 0 && index < K_COUNT) ? kKernelNames[index] : "";
}

int mmsbm_hip_profile_iterations(mmsbm_hip_ctx *ctx, int n_iters, float *mean_us,
                                 int *launches_per_iter) {
  return guarded([&] {
    require_all_params(ctx);
    if (!mean_us || n_iters <= 0) throw std::invalid_argument("bad argument");
    use_device(ctx);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->profiling = true;
    try {
      for (int it = 0; it < n_iters; ++it) launch_iteration(ctx, true);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } catch (...) {
      ctx->profiling = false;
      throw;
    }
    ctx->profiling = false;
    collect_profile(ctx, mean_us, launches_per_iter, n_iters);
  });
}

int mmsbm_hip_kernel_bytes(const mmsbm_hip_ctx *ctx, int index, int64_t *bytes_read,
                           int64_t *bytes_written) {
  return guarded([&] {
    if (!ctx || !bytes_read || !bytes_written) throw std::invalid_argument("null argument");
    const int64_t N = ctx->n_obs, U = ctx->n_users, I = ctx->n_items, R = ctx->n_ratings;
    const int64_t K = ctx->k, L = ctx->l, Q = ctx->n_pairs;
    const int64_t C = static_cast<int64_t>(ctx->lay.mv_chunks.size());
    int64_t rd = 0, wr = 0;
    switch (index) {
      case K_SEG:  // two passes: index + one gathered K-row per triple; fixed rows and offsets once
        rd = 2 * N * (4 + 8 * K) + (U + Q) * (8 * K + 4);
        wr = (U + Q) * 8 * K;
        break;
      case K_DENSE:  // C rows + gathered eta rows + item ids + one tile per block; T rows + slabs
        rd = Q * (8 * K + 8 * L + 4) + C * 8 * K * L;
        wr = Q * 8 * L + C * 8 * K * L;
        break;
      case K_ETAP:  // slabs + p ; T rows through the item CSR + eta
        rd = C * 8 * K * L + R * 8 * K * L + Q * (8 * L + 4) + I * (8 * L + 8);
        wr = 3 * R * 8 * K * L + I * 8 * L;
        break;
      case K_MATVEC_A: rd = Q * (8 * L + 4) + C * 8 * K * L; wr = Q * 8 * K; break;
      case K_FUSED_PAIRS:  // gathered eta rows + ids, the pair pass's index + theta row per triple, two tiles per block
        rd = Q * (8 * L + 4 + 4) + N * (4 + 8 * K) + 2 * C * 8 * K * L;
        wr = Q * (8 * K + 8 * L) + C * 8 * K * L;
        break;
      case K_FUSED_TAIL:   // the user pass + the slabs and p + T rows through the item lists
        rd = N * (4 + 8 * K) + U * (8 * K + 4) + C * 8 * K * L + R * 8 * K * L + Q * (8 * L + 4) + I * (8 * L + 8);
        wr = U * 8 * K + 3 * R * 8 * K * L + I * 8 * L;
        break;
      default: throw std::invalid_argument("kernel index out of range");
    }
    *bytes_read = rd * ctx->n_slots;  // one launch covers every restart slot
    *bytes_written = wr * ctx->n_slots;
  });
}

int mmsbm_hip_time_stage(mmsbm_hip_ctx *ctx, int stage, int reps, float *mean_us) {
  return guarded([&] {
    require_all_params(ctx);
#ifdef MMSBM_ABLATE
    ctx->ablate = stage >> 8;  // diagnostic build: bits 8.. = phases to skip
    stage &= 0xff;
    struct Reset { mmsbm_hip_ctx *c; ~Reset() { c->ablate = 0; } } reset{ctx};
#else
    if (stage >> 8)
      throw ApiError(MMSBM_E_UNSUPPORTED, "time_stage: phase ablation (stage bits 8+) needs the diagnostic build (-DMMSBM_ABLATE, csrc/unity.hip)");
#endif
    if (!mean_us || reps <= 0 || stage < 0 || stage >= K_COUNT)
      throw std::invalid_argument("bad argument");
    if ((stage == K_FUSED_PAIRS || stage == K_FUSED_TAIL) && !fused_possible(ctx))
      throw std::invalid_argument("time_stage: the two-launch kernels do not apply to this shape / data");
    use_device(ctx);
    ensure_a(ctx);
    auto one = [&] {
      switch (stage) {
        case K_SEG: stage_seg(ctx, true, true, true, ctx->stream); break;
        case K_DENSE: stage_dense(ctx); break;
        case K_ETAP: stage_eta_p(ctx, true); break;
        case K_FUSED_PAIRS: stage_fused_pairs(ctx); break;
        case K_FUSED_TAIL: stage_fused_tail(ctx, true); break;
        default: stage_matvec_a(ctx, ctx->cur, ctx->cur ^ 1); break;
      }
    };
    for (int w = 0; w < 3; ++w) one();
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; ++r) one();
    HIP_CHECK(hipEventRecord(e1, ctx->stream));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *mean_us = ms * 1000.f / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  });
}

int mmsbm_hip_set_option(mmsbm_hip_ctx *ctx, const char *name, double value) {
  return guarded([&] {
    if (!ctx || !name) throw std::invalid_argument("null argument");
    const std::string key(name);
    if (key == "graph") {
      ctx->graph_mode = value != 0.0;
    } else if (key == "lik_fast") {  // 0: a logarithm per element; 1: logarithm tables, a group of lanes per triple
                                     // (round 2); 2: where it applies a wave per pair (lik_fact.hpp), else as 1
      if (value != 0.0 && value != 1.0 && value != 2.0) throw std::invalid_argument("lik_fast: 0, 1 or 2");
      ctx->lik_mode = static_cast<int>(value);
    } else if (key == "lik_g") {
      const int g = static_cast<int>(value);
      if (g != 0 && g != 1 && g != 2 && g != 4 && g != 8) throw std::invalid_argument("lik_g: 0, 1, 2, 4 or 8");
      ctx->lik_g = g;
    } else if (key == "quad") {  // 0: the A launch through pair_block like every other shape
      ctx->quad_a = value != 0.0 && !ctx->wide && ctx->kp * ctx->lp > 1024 && ctx->tl_a &&
                    ctx->pb_threads_a == kPairBlockMax && ctx->lds_qa <= kLdsMax - 2048 && ctx->lp <= kQuadMaxL;
    } else if (key == "fused") {  // two launches per iteration (small tiles, unsplit segments); any problem size
      if (value != 0.0 && !fused_possible(ctx)) throw std::invalid_argument("fused: not available for this shape / data");
      ctx->fused = value != 0.0;
      ctx->fused_forced = value != 0.0;
    } else if (key == "nt_out") {  // 0: plain stores for every output row; bits: 1 T and A rows, 2 theta' rows, 4 own-row loads
      if (value < 0 || value > 15) throw std::invalid_argument("nt_out: 0 .. 15");
      ctx->nt_out = static_cast<int>(value);
    } else if (key == "predict_fast") {  // 0: prod_dist / predict through the per-row kernels (R K L multiply-adds per row)
      ctx->predict_fast = value != 0.0;
    } else if (key == "a_units") {  // 64-pair units per workgroup of the matrix-core A launch; 0: the library's own balance
      if (value < 0 || value > kMfmaChunkPairs / kUnitPairs) throw std::invalid_argument("a_units: 0 .. 16");
      use_device(ctx);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      build_a_runs(ctx, static_cast<int>(value));
    } else if (key == "mfma") {  // the pair stage on the matrix cores: 0 off, 1 on (one-block form if K, L <= 64,
                                 // else the blocked form), 2 the blocked form whatever the shape
      ctx->mfma = value == 1.0 && mfma_possible(ctx);
      ctx->mfma_big = value != 0.0 && !ctx->mfma && ctx->mv_chunk_pairs <= kMfmaChunkPairs;
    } else {
      throw std::invalid_argument("unknown option: " + key);
    }
    ctx->drop_graphs();
  });
}

int mmsbm_hip_get_option(const mmsbm_hip_ctx *ctx, const char *name, double *value) {
  return guarded([&] {
    if (!ctx || !name || !value) throw std::invalid_argument("null argument");
    const std::string key(name);
    if (key == "graph") *value = ctx->graph_mode;
    else if (key == "quad") *value = ctx->quad_a;
    else if (key == "mfma") *value = ctx->mfma ? 1.0 : (ctx->mfma_big ? 2.0 : 0.0);
    else if (key == "predict_fast") *value = ctx->predict_fast;
    else if (key == "fused") *value = ctx->fused;
    else if (key == "nt_out") *value = nt_on(ctx);
    else if (key == "launches") *value = use_fused(ctx) ? 2 : 4;  // read-only: launches per iteration at the current slot count
    else if (key == "wide") *value = ctx->wide;
    else if (key == "lik_fast") *value = ctx->lik_mode;
    else if (key == "lik_g") *value = ctx->lik_g;
    else if (key == "ranges_pairs") *value = ctx->ranges_pairs;   // read-only: XCD-local work lists,
    else if (key == "ranges_users") *value = ctx->ranges_users;   // ranges per pass (1 = off)
    else if (key == "chunk_pairs") *value = ctx->mv_chunk_pairs;   // read-only: pairs per pair-stage workgroup at most
    else if (key == "n_chunks") *value = ctx->n_chunks;            // read-only: pair-stage workgroups (= slabs), padding included
    else if (key == "a_units") *value = ctx->a_units;     // 64-pair units per workgroup of the matrix-core A launch (0: not on the matrix cores)
    else if (key == "a_chunks") *value = ctx->n_a_chunks;   // read-only: workgroups of the matrix-core A launch when it walks runs of its own (0: the T + S launch's)
    else if (key == "items_pairs") *value = static_cast<double>(ctx->lay.pair_work.items.size());
    else if (key == "items_users") *value = static_cast<double>(ctx->lay.user_work.items.size());
    else if (key == "splits_pairs") *value = static_cast<double>(ctx->lay.pair_work.splits.size());  // read-only: segments cut into pieces
    else if (key == "splits_users") *value = static_cast<double>(ctx->lay.user_work.splits.size());
    else if (key == "fused_split") *value = (ctx->fs_pairs ? 1 : 0) + (ctx->fs_users ? 2 : 0);  // read-only: whole-segment lists built (1 pair side, 2 user side)
    else throw std::invalid_argument("unknown option: " + key);
  });
}

int mmsbm_hip_set_graph_mode(mmsbm_hip_ctx *ctx, int enabled) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("null context");
    ctx->graph_mode = enabled != 0;
    ctx->drop_graphs();
  });
}

#ifdef MMSBM_STAMPS
// diagnostic build: copy the phase stamps of the last pair-stage launch (blocks x 16 words)
int mmsbm_hip_debug_stamps(unsigned long long *out, int n_words) {
  return guarded([&] {
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n_words));
  });
}
#endif

// ---- host-only layout helpers (no device needed; used by the CPU tests) --------------------
struct mmsbm_hip_layout {
  mmsbm::Layout lay;
};

int mmsbm_hip_layout_build(int64_t n_obs, int32_t n_users, int32_t n_items, int32_t n_ratings,
                           const int32_t *user, const int32_t *item, const int32_t *rating,
                           int32_t target_chunks, mmsbm_hip_layout **out) {
  return guarded([&] {
    if (!out) throw std::invalid_argument("null out");
    *out = nullptr;
    std::unique_ptr<mmsbm_hip_layout> h(new mmsbm_hip_layout());
    mmsbm::build_layout(n_obs, n_users, n_items, n_ratings, user, item, rating, target_chunks,
                        h->lay);
    *out = h.release();
  });
}

int mmsbm_hip_layout_free(mmsbm_hip_layout *h) {
  delete h;
  return MMSBM_OK;
}

// which: 0 pair_off 1 pair_user 2 pair_item 3 rating_off 4 user_off 5 user_pair 6 item_off
//        7 item_pairs 8 item_deg 9 chunk_off; 4-int records: 10 chunks 11 mv_chunks
//        12 pair work items 13 user work items 14 pair splits 15 user splits
int mmsbm_hip_layout_array(const mmsbm_hip_layout *h, int which, int32_t *out, int64_t capacity,
                           int64_t *count) {
  return guarded([&] {
    if (!h || !count) throw std::invalid_argument("null argument");
    const mmsbm::Layout &L = h->lay;
    const std::vector<int32_t> *v = nullptr;
    switch (which) {
      case 0: v = &L.pair_off; break;
      case 1: v = &L.pair_user; break;
      case 2: v = &L.pair_item; break;
      case 3: v = &L.rating_off; break;
      case 4: v = &L.user_off; break;
      case 5: v = &L.user_pair; break;
      case 6: v = &L.item_off; break;
      case 7: v = &L.item_pairs; break;
      case 8: v = &L.item_deg; break;
      case 9: v = &L.chunk_off; break;
      case 10: case 11: case 12: case 13: case 14: case 15: break;
      default: throw std::invalid_argument("unknown layout array");
    }
    if (which >= 10) {  // arrays of 4-int records
      const void *src = nullptr;
      size_t n = 0;
      switch (which) {
        case 10: src = L.chunks.data(); n = L.chunks.size(); break;
        case 11: src = L.mv_chunks.data(); n = L.mv_chunks.size(); break;
        case 12: src = L.pair_work.items.data(); n = L.pair_work.items.size(); break;
        case 13: src = L.user_work.items.data(); n = L.user_work.items.size(); break;
        case 14: src = L.pair_work.splits.data(); n = L.pair_work.splits.size(); break;
        default: src = L.user_work.splits.data(); n = L.user_work.splits.size(); break;
      }
      *count = static_cast<int64_t>(n) * 4;
      if (out && n) {
        if (capacity < *count) throw ApiError(MMSBM_E_TOOLARGE, "buffer too small");
        std::memcpy(out, src, sizeof(int32_t) * *count);
      }
      return;
    }
    *count = static_cast<int64_t>(v->size());
    if (out) {
      if (capacity < *count) throw ApiError(MMSBM_E_TOOLARGE, "buffer too small");
      std::memcpy(out, v->data(), sizeof(int32_t) * v->size());
    }
  });
}

// side 0: pair segments (the 64-pair units rebuilt with at most cap_items work items each), 1: user segments
// (workgroups of at most cap_items items).  which: 0 units, 1 items, 2 splits (4-int records), 3 the unit list as
// chunks (side 0 only), 4 { max partial rows of a workgroup, built (0 / 1) } (2 ints)
int mmsbm_hip_layout_fused(const mmsbm_hip_layout *h, int side, int32_t cap_items, int which, int32_t *out,
                           int64_t capacity, int64_t *count) {
  return guarded([&] {
    if (!h || !count) throw std::invalid_argument("null argument");
    if ((side != 0 && side != 1) || cap_items < 1 || which < 0 || which > 4) throw std::invalid_argument("bad argument");
    mmsbm::Layout lay = h->lay;  // (the capped unit list replaces mv_chunks: work on a copy)
    mmsbm::FusedLists fl;
    bool built = true;
    if (side == 0) {
      const mmsbm::SegPieces sp = mmsbm::segment_pieces(lay.pair_off, lay.pair_work);
      built = mmsbm::build_mv_chunks_capped(lay, sp, mmsbm::kMvChunkPairs, cap_items);
      if (built) fl = mmsbm::build_fused_pairs(lay, sp);
    } else {
      fl = mmsbm::build_fused_users(mmsbm::segment_pieces(lay.user_off, lay.user_work), cap_items);
    }
    const int32_t info[2] = {fl.max_parts, built ? 1 : 0};
    const void *src = nullptr;
    size_t n = 0;
    switch (which) {
      case 0: src = fl.units.data(); n = fl.units.size() * 4; break;
      case 1: src = fl.items.data(); n = fl.items.size() * 4; break;
      case 2: src = fl.splits.data(); n = fl.splits.size() * 4; break;
      case 3: src = lay.mv_chunks.data(); n = side == 0 ? lay.mv_chunks.size() * 4 : 0; break;
      default: src = info; n = 2; break;
    }
    *count = static_cast<int64_t>(n);
    if (out && n) {
      if (capacity < *count) throw ApiError(MMSBM_E_TOOLARGE, "buffer too small");
      std::memcpy(out, src, sizeof(int32_t) * n);
    }
  });
}

int mmsbm_hip_selftest_throw(int kind) {
  return guarded([&] {
    struct NotStd { int x; };
    switch (kind) {
      case 0: return;
      case 1: throw std::invalid_argument("selftest: invalid argument");
      case 2: throw std::runtime_error("selftest: runtime error");
      case 3: throw std::bad_alloc();
      case 4: throw 42;
      default: throw NotStd{kind};
    }
  });
}

}  // extern "C"

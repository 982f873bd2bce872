// context.hpp -- host side: padding rules, device buffers, the context (mmsbm_hip_ctx), launch bookkeeping
// Included by every translation unit of the library (prelude.hpp).
#pragma once

namespace {

// ======================================================================================
// host side
// ======================================================================================
int pad_dim(int d) {  // multiples of 4: 32-byte row granules, whole chunks of 4 outputs
  if (d <= 256) return (d + 3) / 4 * 4;
  if (d <= 512) return (d + 7) / 8 * 8;
  if (d <= 1024) return (d + 15) / 16 * 16;
  return (d + 31) / 32 * 32;  // (whole lanes of the 64 x 32 instantiation of the triple passes)
}

// (G, VEC) instantiation for a padded row length: code 0..6
// Four doubles (32 bytes) per lane: fewer lanes per row means more rows per wave instruction,
// i.e. less vector-ALU work (dot product, DPP reduction, division) per triple.
int group_code(int dp) {
  if (dp <= 16) return 0;   // G=4  VEC=4
  if (dp <= 32) return 1;   // G=8  VEC=4
  if (dp <= 64) return 2;   // G=16 VEC=4
  if (dp <= 128) return 3;  // G=32 VEC=4
  if (dp <= 256) return 4;  // G=64 VEC=4
  if (dp <= 512) return 5;  // G=64 VEC=8
  return 6;                 // G=64 VEC=16 (up to 1,024 groups)
}
constexpr int kMaxGroupRow = 1024;  // columns the widest (G, VEC) covers; wider rows: seg_wide_kernel / block loops
int group_lanes(int code) {
  static const int g[7] = {4, 8, 16, 32, 64, 64, 64};
  return g[code];
}

#define DISPATCH_GV(code, CALL)                                   \
  switch (code) {                                                 \
    case 0: CALL(4, 4); break;                                    \
    case 1: CALL(8, 4); break;                                    \
    case 2: CALL(16, 4); break;                                   \
    case 3: CALL(32, 4); break;                                   \
    case 4: CALL(64, 4); break;                                   \
    case 5: CALL(64, 8); break;                                   \
    default: CALL(64, 16); break;                                 \
  }

}  // namespace

namespace mmsbm_hip_impl {  // (members of mmsbm_hip_ctx: one type in every translation unit)

template <class T>
struct DevBuf {
  T *ptr = nullptr;
  size_t count = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  void alloc(size_t n) {
    release();
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ptr), std::max<size_t>(n, 1) * sizeof(T)));
    count = n;  // (only once the memory is there: a failed allocation leaves an empty buffer)
  }
  void upload(const std::vector<T> &h, hipStream_t s) {
    alloc(h.size());
    if (!h.empty())
      HIP_CHECK(hipMemcpyAsync(ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
};

// A per-restart table: `slots` copies, `stride` doubles apart (whole 128-byte lines, so every
// copy keeps the alignment of the first).
struct SlotBuf : DevBuf<double> {
  size_t stride = 0;
  void alloc_slots(size_t per_slot, int slots) {
    stride = (per_slot + 15) / 16 * 16;
    alloc(stride * static_cast<size_t>(slots));
  }
  double *at(int slot) const { return ptr + static_cast<size_t>(slot) * stride; }
};

// Pinned host staging: parameter rows travel as ONE contiguous copy in the device layout
// (packed / unpacked on the host by a few threads) instead of strided 2-D copies from
// pageable memory.
struct PinBuf {
  double *ptr = nullptr;
  size_t cap = 0, used = 0;
  PinBuf() = default;
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() {
    if (ptr) (void)hipHostFree(ptr);
  }
  void reset(size_t need) {
    used = 0;
    if (need <= cap) return;
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    cap = 0;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ptr), need * sizeof(double), hipHostMallocDefault));
    cap = need;
  }
  double *take(size_t n) {
    double *r = ptr + used;
    used += n;
    return r;
  }
};

}  // namespace mmsbm_hip_impl

namespace {

// fn(first_row, last_row) over [0, rows), on up to 8 host threads when the table is large
template <class F>
void for_row_blocks(int rows, size_t row_doubles, F &&fn) {
  const size_t total = static_cast<size_t>(rows) * row_doubles;
  unsigned nt = total < (size_t(1) << 19) ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (nt <= 1) {
    fn(0, rows);
    return;
  }
  std::vector<std::thread> th;
  const int per = (rows + static_cast<int>(nt) - 1) / static_cast<int>(nt);
  for (unsigned t = 0; t < nt; ++t) {
    const int a = static_cast<int>(t) * per, b = std::min(rows, a + per);
    if (a < b) th.emplace_back([&fn, a, b] { fn(a, b); });
  }
  for (auto &x : th) x.join();
}

enum KernelId { K_SEG = 0, K_DENSE, K_ETAP, K_MATVEC_A, K_FUSED_PAIRS, K_FUSED_TAIL, K_COUNT };
// The four launches of an iteration -- or, for small problems, the two of fused_small.hpp.
const char *const kKernelNames[K_COUNT] = {"seg_pass_kernel", "pair_block_kernel(T+S)", "eta_p_kernel",
                                           "pair_block_kernel(A)", "pairs_fused_kernel", "tail_fused_kernel"};


}  // namespace


struct mmsbm_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // parameter downloads that overlap kernels of `stream` (mmsbm_hip_result)
  hipStream_t xfer = nullptr;         // when set: the stream copy_rows / fetch_params use instead of `stream`
  bool swapped = false;
  // external dims
  int64_t n_obs = 0;
  int ext_users = 0, ext_items = 0, ext_k = 0, ext_l = 0;
  // internal dims ("item" = the side paired with the rating)
  int n_users = 0, n_items = 0, n_ratings = 0, k = 0, l = 0, kp = 0, lp = 0;
  int n_pairs = 0, n_chunks = 0;
  int code_k = 0, code_l = 0;
  bool direct_out = false;  // pair_block: output rows stored straight from registers (no LDS transpose)
#ifdef MMSBM_ABLATE
  int ablate = 0;           // diagnostic build (-DMMSBM_ABLATE): phases the pair stage / eta_p skip, set by mmsbm_hip_time_stage
#endif
  bool split_rows = false;  // theta and A kept as 128-byte main lines + tail rows (RowTab)
  int pb_threads_t = kBlock, pb_threads_a = kBlock;  // pair_block workgroup sizes (T+S mode, A mode)
  int pb_kt = 4;  // pair_block S phase: k-rows per register tile (2 when K x L is small)
  int pb_spb = kBlock, pb_nacc = 1, pb_nsub = 1;  // pair_block S phase: threads per slot-grid copy, slots per thread
  size_t lds_t = 0, lds_a = 0;
  bool tl_t = false, tl_a = false;  // rating tile staged in LDS (T+S launch / A launch)
  bool quad_a = false;  // the A launch runs pair_quad_a_kernel (long rows)
  // prod_dist / predict through B[(item, rating), :] = p_r eta_i (predict_rows_kernel)
  DevBuf<int32_t> grid_item;          // item of pair q = r * I + i, every (item, rating) combination
  DevBuf<mmsbm::Chunk> grid_chunks;   // its rating-homogeneous chunks
  int grid_n_chunks = 0, mv_chunk_pairs = mmsbm::kMvChunkPairs;
  DevBuf<double> btab;                // [I * R][kp], of the slot being scored
  bool predict_fast = true;
  int seg_batch = 4;  // row gathers a group of seg_pass keeps in flight (4, or 8)
  bool fused = false;          // small problems: two launches per iteration (fused_small.hpp)
  bool fused_forced = false;   // option "fused" = 1: whatever the size (else: while ratings x restart slots x (K + L) <= 14M)
  // ... with segments that are cut into pieces (uneven degrees): every segment's pieces inside one workgroup
  // (layout.hpp: FusedLists), pair side (the units of pairs_fused_kernel) and / or user side (blocks of tail_fused_kernel)
  bool fs_pairs = false, fs_users = false;
  DevBuf<mmsbm::FusedUnit> fp_units, fu_units;
  DevBuf<mmsbm::WorkItem> fp_items, fu_items;
  DevBuf<mmsbm::FusedSplit> fp_splits, fu_splits;
  int fp_max_parts = 0, fu_max_parts = 0, fu_blocks = 0;
  std::vector<char> a_ok;      // per slot: atab[cur] holds A of the CURRENT parameters (the fused form computes A at
                               // the start of an iteration, so after a committed fused iteration it does not)
  bool mfma = false;    // both pair-stage launches run pair_mfma_kernel (tiles beyond the scalar cache, K, L <= 64)
  size_t lds_mt = 0, lds_ma = 0;
  bool mfma_big = false;  // K or L beyond 64: the blocked forms (mfma_rows_kernel + mfma_slab_kernel)
  bool wide = false;    // K, L beyond the LDS stage: wide_matvec / wide_slab kernels (any size)
  int nt_out = 7;          // option "nt_out" (bits: 1 T and A rows, 2 theta' rows as non-temporal stores, 4 the segments' own rows as non-temporal loads) where that pays (nt_on, stages.hpp)
  int ranges_pairs = 1, ranges_users = 1;  // XCD-local work lists: ranges the gathered table is cut into
  int n_cus = 256;
  size_t lds_qa = 0;
  mmsbm::Layout lay;  // host copy (degrees, sizes)
  DevBuf<int32_t> pair_off, pair_user, pair_item, user_off, user_pair, item_off, item_pairs,
      item_deg, mv_chunk_off, orig_u, orig_i, orig_r;
  DevBuf<int32_t> item_grid;  // [n_items][n_ratings] pair ids (-1: none); only for dense (item, rating) grids
  DevBuf<mmsbm::Chunk> mv_chunks;
  DevBuf<mmsbm::Chunk> lik_units;  // 64-pair units for the likelihood kernel (mv_chunks may hold 256)
  DevBuf<mmsbm::Chunk> a_chunks;   // matrix-core A launch: its own runs of units (balanced_run_units, stages.hpp), when they differ from mv_chunks
  int n_a_chunks = 0, a_units = 0;   // (a_units: 64-pair units per workgroup of that launch; option "a_units")
  int n_lik_units = 0;
  DevBuf<mmsbm::WorkItem> pair_items, user_items;   // only when some segment is long
  DevBuf<mmsbm::SplitSeg> pair_splits, user_splits;
  // Per-restart state, one copy per slot.  A context carries n_slots independent restarts
  // (parameter sets) over the SAME triples; em_iterate advances all of them with one set of
  // launches (blockIdx.y = slot), the single-restart entry points act on slot `sel`.
  int n_slots = 1, sel = 0;
  int base_slot = 0, launch_slots = 1;  // what the next launches cover: [base_slot, base_slot + launch_slots)
  SlotBuf pair_parts, user_parts;
  SlotBuf theta[2], eta[2], p[2], pt[2], atab[2], ctab, ttab, partial, npr;
  DevBuf<double> lik_part;
  DevBuf<double> lg_theta, lg_eta, lg_p;  // logarithm tables of the selected slot (likelihood)
  int lik_mode = 2;                       // option "lik_fast": 0 log per element, 1 log tables, 2 a wave per pair where it applies
  int lik_g = 0;                          // option "lik_g": lanes per triple (0 = automatic)
  PinBuf pin;  // host staging for set_params / get_params / update_coefficients
  // predict/score session (mmsbm_hip_predict_begin .. finish)
  DevBuf<int32_t> ps_u, ps_i, ps_r;
  DevBuf<double> ps_sum, ps_w, ps_part;
  int64_t ps_rows = -1;  // -1: no session open
  int ps_added = 0;
  int cur = 0;
  std::vector<char> have;  // per slot: set_params has been called
  bool graph_mode = false;  // replay a captured two-iteration hipGraph instead of eager launches
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};  // indexed by `cur` at capture time
  // per-launch profiling
  bool profiling = false;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;

  void drop_graphs() {
    for (auto &g : graph_exec) {
      if (g) (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
  }
  ~mmsbm_hip_ctx() {
    drop_graphs();
    for (auto &pe : prof_events) {
      (void)hipEventDestroy(pe.second.first);
      (void)hipEventDestroy(pe.second.second);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {


// Optional event pair of one stage (mmsbm_hip_profile_iterations).  Normally the events are recorded on the
// stream in front of and behind the stage's launches, which adds the launch gaps to what they measure
// (seg_pass at C3: 64.8 us against 62.2 in the kernel trace).  A stage that is ONE kernel can hand the pair
// to the launch itself instead (hipExtLaunchKernelGGL): the events then carry the kernel's own begin and end
// time stamps -- what rocprofv3 reports.
struct LaunchScope {
  mmsbm_hip_ctx *c;
  int id;
  bool kernel_events;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  LaunchScope(mmsbm_hip_ctx *ctx, int kid, bool one_kernel = false) : c(ctx), id(kid), kernel_events(one_kernel) {
    if (c->profiling) {
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      if (!kernel_events) HIP_CHECK(hipEventRecord(e0, c->stream));
    }
  }
  bool ext() const { return c->profiling && kernel_events; }  // the launch takes e0 / e1
  void done() {
    HIP_CHECK(hipGetLastError());
    if (c->profiling) {
      if (!kernel_events) HIP_CHECK(hipEventRecord(e1, c->stream));
      c->prof_events.push_back({id, {e0, e1}});
    }
  }
};

void use_device(const mmsbm_hip_ctx *c) { HIP_CHECK(hipSetDevice(c->device)); }


}  // namespace

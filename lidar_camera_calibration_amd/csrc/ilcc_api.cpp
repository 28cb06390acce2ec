// ilcc_api.cpp -- host side of the C-ABI (include/ilcc_hip.h): handle, HBM buffers, the stage
// pipeline on one HIP stream, HIP-event timing, and the two file contracts of the path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "ilcc_internal.h"

using namespace ilcc;

#ifndef ILCC_SLOTS
#define ILCC_SLOTS 4
#endif
constexpr int kSlots = ILCC_SLOTS;   // batches in flight per handle (submit/wait); slot 0 serves the synchronous calls.
                            // Every slot has its own stream: the process needs GPU_MAX_HW_QUEUES >= 5 (HIP default: 4), otherwise two
                            // slot streams share one hardware queue and serialise (measured: 113 k instead of 164 k frames/s)

// One in-flight batch: its own stream, events, stage buffers and pinned result staging.
struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t ev[10]{};   // 0..6 stage boundaries; 7, 8: end of K6's seed + refinement passes / start of its full pass; 9: behind K1's count pass
  hipEvent_t k6ev[4]{};  // inside the K6 stage: after K5w, after the seed launch, after the refinement launch, after the anchor launch (launch-by-launch event spans)
  hipEvent_t k6_done = nullptr;
  bool allocated = false;
  // device buffers
  float4* d_xyzi = nullptr;   // staging for host-input calls
  float* d_clicks = nullptr;
  uint64_t* d_off = nullptr;
  ilcc_result* d_res = nullptr;
  float4 *d_roi = nullptr, *d_cluster = nullptr, *d_board = nullptr, *d_pca = nullptr, *d_optim = nullptr;
  float2 *d_yz = nullptr, *d_walk_yz = nullptr;
  uint8_t* d_walk_lab = nullptr;
  uint32_t *d_walk_mi = nullptr, *d_walk_nrim = nullptr;
  uint8_t *d_lab = nullptr, *d_cls = nullptr;
  uint32_t *d_nlab = nullptr, *d_walk = nullptr, *d_counts = nullptr, *d_parent = nullptr, *d_count = nullptr;
  unsigned long long* d_masks = nullptr;
  uint32_t *d_hash_head = nullptr, *d_hash_next = nullptr;
  uint32_t *d_flags = nullptr, *d_nfinite = nullptr, *d_counts_fin = nullptr;   // the online caller's two tiers (Ctx::frame_flags, n_finite, crop_fin)
  uint32_t* h_online = nullptr;   // pinned: flags then finite counts of the last online batch
  uint32_t* d_list = nullptr;     // tier 2: [0] count, [1..] listed frames
  void* d_list_frames = nullptr;  // tier 2: 64 bytes per frame
  GridPartial *d_partial = nullptr, *d_partial2 = nullptr, *d_partial3 = nullptr, *d_partial4 = nullptr;
  SolveRec* d_solverec = nullptr;
  uint32_t *d_bound = nullptr, *d_bound_sub = nullptr;
  uint32_t* d_tie_count = nullptr;
  GridPartial* d_tie_list = nullptr;
  unsigned long long* d_iters = nullptr;
  uint32_t *d_grp_alive = nullptr, *d_grp_mask = nullptr;   // K6's common pre-pass per (frame, group of kThetaGroup thetas): state word, rejected-tile mask
  size_t grp_alive_cap = 0, grp_mask_cap = 0;   // sized from (max_frames, the handle's grid) in alloc_slot / ilcc_set_params, never in the submit path
  float* h_rec_dev = nullptr;               // the DEVICE address of h_rec (pinned, mapped): K9 stores the batch's compact records straight into it
  unsigned long long* h_iters_dev = nullptr;   // ... and of h_iters
  // pinned host staging
  float* h_rec = nullptr;
  ilcc_result* h_res = nullptr;
  unsigned long long* h_iters = nullptr;
  uint64_t* h_off = nullptr;   // pinned copy of `off`: the upload at the head of a batch must not stage through pageable memory
  // state of the batch in flight / last completed
  std::vector<uint64_t> off;
  uint32_t n_frames = 0;
  bool busy = false, grid = false;
  bool online = false;   // the batch in flight came from ilcc_submit_chessboard_by_point (no crop, front end only): its wait applies the by-point epilogue
  bool h_res_valid = false;  // h_res holds the last completed batch's full (trimmed) records
  bool compact = false;      // what was enqueued with the batch in flight: the K9 records (true) or the trimmed full records
  uint32_t rec_corners = 0;  // corners per K9 record of that batch
};

struct ilcc_handle {
  int device = 0;
  ilcc_params p{};
  uint32_t max_frames = 0;
  uint64_t max_points = 0;
  uint32_t max_theta = 0;
  uint32_t crop_chunks_cap = 0;
  Slot slots[kSlots];
  int next_slot = 0;    // round robin for ilcc_submit_*
  int last_slot = -1;   // slot whose batch the fetch calls look at
  int k6_last = -1;     // slot that issued the most recent K6 (K6 launches are chained: clean timing)
  // candidate tables (shared by all slots, read-only while batches are in flight)
  float *d_cth = nullptr, *d_sth = nullptr, *d_ay = nullptr, *d_az = nullptr;
  // decimated subset of the same tables: seeding pass of K6's branch and bound
  float *d_cth2 = nullptr, *d_sth2 = nullptr, *d_ay2 = nullptr, *d_az2 = nullptr;
  int32_t n_th2 = 0, n_ty2 = 0, n_tz2 = 0, seed_stride_th = 12, seed_stride_t = 2;
  // K7r: cos/sin of the theta lattice (grid step / refine_div, refine_th_margin grid steps beyond the grid on both sides)
  double2* d_th_lattice = nullptr;
  size_t th_lattice_cap = 0;
  int32_t th_lat_lo = 0, th_lat_hi = 0, hop_y = 0, hop_z = 0;
  double* d_solve = nullptr;   // 3 theta_t + 2 (cost, iterations) for the test entries
  RefineOut* d_refine_io = nullptr;
  uint32_t grid_lds_points = 1024;   // grows with the frames seen (finish()); frames above it take the global-memory path
  uint64_t group_prepass_skipped = 0;   // batches whose grid was eligible for k6_group_prepass but whose mask buffers were too small: a sizing bug if ever non-zero -- the first such batch leaves a note in ilcc_last_error (results are unaffected: the full pass runs without the common mask)
  uint32_t cluster_lds_points = 2048;   // K2's LDS capacity for cell-sorted ROI points per frame: grows likewise (<= 4096); larger frames sort into HBM
  uint32_t cluster_cells_cap = kClusterCellsMin;   // K2's LDS capacity in occupied cells per frame: grows with what the batches needed
  uint32_t list_grid = 1024;         // workgroups of the online caller's second-tier kernels (4 x the device's CUs)
  bool poisoned = false;             // a failed ilcc_set_params could not restore the device tables: every later call fails
  int32_t result_mode = ILCC_RESULTS_FULL;
  ilcc_timing timing{};
  hipEvent_t tl_ref = nullptr;       // ilcc_debug_timeline_*: reference event, rows of ILCC_TIMELINE_COLS doubles
  bool tl_on = false;
  std::vector<double> tl_rows;
  std::string err;
};

namespace {

thread_local std::string g_err;

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
      return ILCC_HIP_ERROR;                                                               \
    }                                                                                      \
  } while (0)

int32_t near_zero_index(double vmin, double step, int32_t n) {
  long c = std::lround(-vmin / step);
  if (c < 0) c = 0;
  if (c > n - 1) c = n - 1;
  return (int32_t)c;
}

bool params_ok(const ilcc_params& p, std::string& why) {
  auto bad = [&](const char* m) {
    why = m;
    return false;
  };
  if (!(p.grid_length > 0)) return bad("grid_length must be > 0");
  if (p.board_w < 2 || p.board_h < 2 || p.board_w > p.board_h) return bad("board_w/board_h: need 2 <= w <= h");
  if ((p.board_w - 1) * (p.board_h - 1) > ILCC_MAX_CORNERS) return bad("too many corners");
  if (p.hist_bins < 1 || p.hist_bins > 4096) return bad("hist_bins out of range");
  if (!(p.gray_rate > 0) || !(p.huber_delta > 0)) return bad("gray_rate / huber_delta must be > 0");
  if (p.ransac_hyp < 1 || p.ransac_hyp > 65536) return bad("ransac_hyp out of range");
  if (!(p.ransac_probability == p.ransac_probability) || p.ransac_probability >= 1.0) return bad("ransac_probability must be < 1");
  if (!(p.cluster_tol > 0) || p.cluster_min < 1 || p.cluster_max < p.cluster_min) return bad("cluster params");
  if (p.solver != ILCC_SOLVER_REFERENCE_LOCAL && p.solver != ILCC_SOLVER_GRID) return bad("solver");
  if (p.phase_mode < 0 || p.phase_mode > 2) return bad("phase_mode");
  if (p.grid_prune != 0 && p.grid_prune != 1) return bad("grid_prune must be 0 or 1");
  if (p.max_iterations < 0 || p.max_iterations > 100000) return bad("max_iterations");
  if (p.n_th < 1 || p.n_ty < 1 || p.n_tz < 1) return bad("grid sizes must be >= 1");
  if (p.n_th > 4096 || p.n_ty > 4096 || p.n_tz > 4096) return bad("grid axes are limited to 4096 candidates");
  // K6 keeps the (ty, tz) tables in LDS behind the staged points: the raised dynamic-LDS limit covers kGridTableMax floats
  if (p.n_ty + p.n_tz > kGridTableMax) return bad("n_ty + n_tz exceeds the LDS table capacity of the grid kernel");
  if (p.refine_div < 0 || p.refine_div > 64 || (p.refine_div & (p.refine_div - 1)) != 0)
    return bad("refine_div must be 0 or a power of two <= 64");
  if (p.refine_max_rounds < 0 || p.refine_max_rounds > 4096) return bad("refine_max_rounds");
  if (p.refine_th_margin < 0 || p.refine_th_margin > 4096) return bad("refine_th_margin");
  if (!(p.online_cluster_tol > 0)) return bad("online_cluster_tol must be > 0");
  if (!(p.ambiguity_eps == p.ambiguity_eps)) return bad("ambiguity_eps is NaN");
  if (!(p.min_cell_coverage == p.min_cell_coverage) || p.min_cell_coverage > 1.0) return bad("min_cell_coverage must be <= 1");
  if (p.board_w * p.board_h > kCoverageCellsMax) return bad("board has more squares than the coverage mask holds");
  if ((uint64_t)p.n_th * p.n_ty * p.n_tz * 2ull >= 0xFFFFFFFFull) return bad("grid too large");
  if (!(p.th_step > 0) || !(p.ty_step > 0) || !(p.tz_step > 0)) return bad("grid steps must be > 0");
  {
    // K7r's basin check compares with the positions ONE SQUARE away: lround(g / (step / div)) lattice units.  A step so
    // coarse that this rounds to 0 would compare the centre with itself (margin 0: every frame ILCC_AMBIGUOUS)
    const double div = (double)(p.refine_div > 0 ? p.refine_div : 1);
    if (std::lround(p.grid_length / (p.ty_step / div)) < 1 || std::lround(p.grid_length / (p.tz_step / div)) < 1)
      return bad("ty_step / tz_step too coarse: one board square is less than half a refinement-lattice step");
  }
  return true;
}

int32_t sync_all(ilcc_handle* h) {
  for (Slot& sl : h->slots)
    if (sl.allocated) HIP_TRY(h, hipStreamSynchronize(sl.stream));
  return ILCC_OK;
}

int32_t upload_tables(ilcc_handle* h) {
  const ilcc_params& p = h->p;
  if ((uint32_t)p.n_th > h->max_theta || (uint32_t)p.n_ty > h->max_theta || (uint32_t)p.n_tz > h->max_theta) {
    h->err = "grid axis longer than the handle's table capacity";
    return ILCC_CAPACITY;
  }
  hipStream_t st = h->slots[0].stream;
  std::vector<float> cth(p.n_th), sth(p.n_th), ay(p.n_ty), az(p.n_tz);
  const double g = p.grid_length;
  for (int k = 0; k < p.n_th; ++k) {
    const double th = p.th_min + k * p.th_step;
    cth[k] = (float)(std::cos(th) / g);
    sth[k] = (float)(std::sin(th) / g);
  }
  for (int a = 0; a < p.n_ty; ++a) ay[a] = (float)(((p.ty_min + a * p.ty_step) + p.board_w * g / 2.0) / g);
  for (int b = 0; b < p.n_tz; ++b) az[b] = (float)(((p.tz_min + b * p.tz_step) + p.board_h * g / 2.0) / g);
  HIP_TRY(h, hipMemcpyAsync(h->d_cth, cth.data(), sizeof(float) * p.n_th, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(h->d_sth, sth.data(), sizeof(float) * p.n_th, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(h->d_ay, ay.data(), sizeof(float) * p.n_ty, hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemcpyAsync(h->d_az, az.data(), sizeof(float) * p.n_tz, hipMemcpyHostToDevice, st));
  // seed subset: ~5 thetas (centred, so theta = 0 is one of them for a symmetric grid) x 8 x 8 translations --
  // the same float values as the full tables.  The seed only has to land in the right basin: the refinement
  // pass then evaluates everything around its argmin (theta +- half a seed stride, 8 x 8 translations), which
  // on the synthetic VLP-16 set yields the exact grid minimum as the bound in 46 of 46 frames (seed alone:
  // 3-20 x the minimum).  Measured on the 128-frame batch (K6 ms): translation stride 2: 1.05, 4: 0.91, 5: 0.83.
#ifndef ILCC_SEED_THETAS
#define ILCC_SEED_THETAS 5
#endif
#ifndef ILCC_REFINE_RADIUS_DIV
#define ILCC_REFINE_RADIUS_DIV 3   // refinement pass: theta radius = seed stride / 3 (measured in the pipelined bench: /2: 245.3 k, /3: 248.9 k, /4: 245.6 k frames/s)
#endif
  h->seed_stride_th = std::max(2, p.n_th / ILCC_SEED_THETAS);
  h->seed_stride_t = std::max(1, std::min(p.n_ty, p.n_tz) / 8);
  std::vector<float> cth2, sth2, ay2, az2;
  for (int k = h->seed_stride_th / 2; k < p.n_th; k += h->seed_stride_th) {
    cth2.push_back(cth[k]);
    sth2.push_back(sth[k]);
  }
  for (int a = 0; a < p.n_ty; a += h->seed_stride_t) ay2.push_back(ay[a]);
  for (int b = 0; b < p.n_tz; b += h->seed_stride_t) az2.push_back(az[b]);
  h->n_th2 = (int32_t)cth2.size();
  h->n_ty2 = (int32_t)ay2.size();
  h->n_tz2 = (int32_t)az2.size();
  if (h->n_th2 > 0) {
    HIP_TRY(h, hipMemcpyAsync(h->d_cth2, cth2.data(), sizeof(float) * cth2.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->d_sth2, sth2.data(), sizeof(float) * sth2.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->d_ay2, ay2.data(), sizeof(float) * ay2.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->d_az2, az2.data(), sizeof(float) * az2.size(), hipMemcpyHostToDevice, st));
  }
  // K7r: cos/sin of every theta lattice point, from the host's libm -- the oracle evaluates cos()/sin() of the
  // SAME doubles (th_min + q * (th_step / div)) with the same libm, so the kernel's per-point terms are the oracle's
  {
    const int div = p.refine_div > 0 ? p.refine_div : 1;
    h->th_lat_lo = -p.refine_th_margin * div;
    h->th_lat_hi = (p.n_th - 1 + p.refine_th_margin) * div;
    const size_t n = (size_t)(h->th_lat_hi - h->th_lat_lo) + 1;
    if (n > h->th_lattice_cap) {
      if (h->d_th_lattice) (void)hipFree(h->d_th_lattice);
      h->d_th_lattice = nullptr;
      h->th_lattice_cap = 0;
      HIP_TRY(h, hipMalloc((void**)&h->d_th_lattice, sizeof(double2) * n));
      h->th_lattice_cap = n;
    }
    std::vector<double2> tab(n);
    for (size_t i = 0; i < n; ++i) {
      const double th = p.th_min + (double)(h->th_lat_lo + (int64_t)i) * (p.th_step / (double)div);
      tab[i] = make_double2(std::cos(th), std::sin(th));
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_th_lattice, tab.data(), sizeof(double2) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));   // tab is a local
    h->hop_y = (int32_t)std::lround(p.grid_length / (p.ty_step / (double)div));
    h->hop_z = (int32_t)std::lround(p.grid_length / (p.tz_step / (double)div));
  }
  HIP_TRY(h, hipStreamSynchronize(st));
  return ILCC_OK;
}

void free_slot(Slot& sl) {
  void* bufs[] = {sl.d_grp_alive, sl.d_grp_mask, sl.d_xyzi, sl.d_clicks, sl.d_off, sl.d_res, sl.d_roi, sl.d_cluster, sl.d_board, sl.d_pca, sl.d_optim,
                  sl.d_yz, sl.d_walk_yz, sl.d_walk_lab, sl.d_walk_mi, sl.d_walk_nrim, sl.d_lab, sl.d_cls, sl.d_nlab, sl.d_walk, sl.d_counts, sl.d_parent, sl.d_count, sl.d_hash_head, sl.d_hash_next, sl.d_flags, sl.d_nfinite, sl.d_counts_fin, sl.d_list, sl.d_list_frames, sl.d_partial, sl.d_partial2, sl.d_partial3, sl.d_partial4, sl.d_masks,
                  sl.d_solverec, sl.d_bound, sl.d_bound_sub, sl.d_iters, sl.d_tie_count, sl.d_tie_list};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (sl.h_res) (void)hipHostFree(sl.h_res);
  if (sl.h_rec) (void)hipHostFree(sl.h_rec);
  if (sl.h_iters) (void)hipHostFree(sl.h_iters);
  if (sl.h_off) (void)hipHostFree(sl.h_off);
  if (sl.h_online) (void)hipHostFree(sl.h_online);
  for (auto& ev : sl.ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : sl.k6ev)
    if (ev) (void)hipEventDestroy(ev);
  if (sl.k6_done) (void)hipEventDestroy(sl.k6_done);
  if (sl.stream) (void)hipStreamDestroy(sl.stream);
  sl = Slot{};
}

// The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue
// serialise: a fourth batch in flight only pays with >= 5 queues.  The variable is read when the runtime initialises and
// belongs to the HOST process (the Python package exports it on import, the C++ mains in host/ first thing in main()):
// the library only reads it, and says so once when the fourth slot is first used with fewer than 5 queues configured.
void warn_hw_queues_once(int slot_index) {
  static std::once_flag once;   // handles of several threads (one per GPU) may reach this together
  if (slot_index < 3) return;
  std::call_once(once, [] {
    const char* v = std::getenv("GPU_MAX_HW_QUEUES");
    if (v && std::atoi(v) >= 5) return;
    std::fprintf(stderr,
                 "libilcc_hip: GPU_MAX_HW_QUEUES=%s: with fewer than 5 hardware queues the fourth batch in flight shares a "
                 "queue with another one and serialises; keep at most 3 tickets outstanding or export GPU_MAX_HW_QUEUES=8 "
                 "before the HIP runtime starts (the library does not modify the environment)\n", v ? v : "(unset: HIP's default is 4)");
  });
}

// K6's common pre-pass: does the handle's grid qualify, and what its output needs per frame
struct GroupPrepassPlan {
  bool box = false;   // the grid is eligible for the box pre-passes at all (the ONE place that decides it)
  bool on = false;
  uint32_t groups = 0, words = 0;   // theta groups per frame = ceil(n_th / kThetaGroup); mask words per group = ceil(tiles / 32)
};
GroupPrepassPlan group_prepass_plan(const ilcc_params& p) {
  GroupPrepassPlan g;
  // box pre-pass (k6_grid_cost.hip): its monotonicity argument needs a tile's box narrower than one square on both axes
  const bool box = p.grid_prune != 0 && 3.0 * p.ty_step < 0.9 * p.grid_length && 3.0 * p.tz_step < 0.9 * p.grid_length;
  // ... and, for the points the common pre-pass leaves out, every translation of the tables keeping the board's centre inside the board
  const double gl = p.grid_length;
  const double ty_hi = p.ty_min + (p.n_ty - 1) * p.ty_step, tz_hi = p.tz_min + (p.n_tz - 1) * p.tz_step;
  const bool centre_in = p.ty_min > -0.45 * p.board_w * gl && ty_hi < 0.45 * p.board_w * gl && p.tz_min > -0.45 * p.board_h * gl &&
                         tz_hi < 0.45 * p.board_h * gl;
  const uint32_t n_tiles = (uint32_t)(((p.n_ty + 3) / 4) * ((p.n_tz + 3) / 4));
  g.box = box;
  g.on = box && p.n_th >= kThetaGroup && centre_in && n_tiles <= 4096u;
  g.groups = (uint32_t)((p.n_th + kThetaGroup - 1) / kThetaGroup);
  g.words = (n_tiles + 31u) / 32u;
  return g;
}

// (re)size a slot's common pre-pass buffers for the handle's current grid.  hipFree synchronises the device, so this runs
// where no batch can be in flight: alloc_slot (a slot's first use) and ilcc_set_params -- never inside a submit.
int32_t size_group_prepass(ilcc_handle* h, Slot& sl) {
  const GroupPrepassPlan g = group_prepass_plan(h->p);
  if (!g.on) return ILCC_OK;
  const size_t need_alive = (size_t)h->max_frames * g.groups, need_mask = need_alive * g.words;
  if (need_alive > sl.grp_alive_cap) {
    if (sl.d_grp_alive) (void)hipFree(sl.d_grp_alive);
    sl.d_grp_alive = nullptr;
    sl.grp_alive_cap = 0;
    HIP_TRY(h, hipMalloc((void**)&sl.d_grp_alive, sizeof(uint32_t) * need_alive));
    sl.grp_alive_cap = need_alive;
  }
  if (need_mask > sl.grp_mask_cap) {
    if (sl.d_grp_mask) (void)hipFree(sl.d_grp_mask);
    sl.d_grp_mask = nullptr;
    sl.grp_mask_cap = 0;
    HIP_TRY(h, hipMalloc((void**)&sl.d_grp_mask, sizeof(uint32_t) * need_mask));
    sl.grp_mask_cap = need_mask;
  }
  return ILCC_OK;
}

int32_t alloc_slot(ilcc_handle* h, Slot& sl) {
  if (sl.allocated) return ILCC_OK;
  warn_hw_queues_once((int)(&sl - h->slots));
  const uint64_t np = h->max_points;
  const uint32_t mf = h->max_frames;
  HIP_TRY(h, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
  for (auto& ev : sl.ev) HIP_TRY(h, hipEventCreate(&ev));
  for (auto& ev : sl.k6ev) HIP_TRY(h, hipEventCreate(&ev));
  HIP_TRY(h, hipEventCreateWithFlags(&sl.k6_done, hipEventDisableTiming));
#define ALLOC(ptr, bytes) HIP_TRY(h, hipMalloc((void**)&(ptr), (size_t)(bytes)))
  ALLOC(sl.d_xyzi, sizeof(float4) * np);
  ALLOC(sl.d_clicks, sizeof(float) * 3 * mf);
  ALLOC(sl.d_off, sizeof(uint64_t) * (mf + 1));
  ALLOC(sl.d_res, sizeof(ilcc_result) * mf);
  ALLOC(sl.d_roi, sizeof(float4) * np);
  ALLOC(sl.d_cluster, sizeof(float4) * np);
  ALLOC(sl.d_board, sizeof(float4) * np);
  ALLOC(sl.d_pca, sizeof(float4) * np);
  ALLOC(sl.d_optim, sizeof(float4) * np);
  ALLOC(sl.d_yz, sizeof(float2) * np);
  ALLOC(sl.d_walk_yz, sizeof(float2) * np);
  ALLOC(sl.d_walk_lab, np);
  ALLOC(sl.d_walk_mi, sizeof(uint32_t) * mf);
  ALLOC(sl.d_walk_nrim, sizeof(uint32_t) * mf);
  ALLOC(sl.d_lab, np);
  ALLOC(sl.d_cls, np);
  ALLOC(sl.d_nlab, sizeof(uint32_t) * mf);
  ALLOC(sl.d_walk, sizeof(uint32_t) * mf);
  ALLOC(sl.d_counts, sizeof(uint32_t) * h->crop_chunks_cap);
  ALLOC(sl.d_counts_fin, sizeof(uint32_t) * h->crop_chunks_cap);
  ALLOC(sl.d_flags, sizeof(uint32_t) * mf);
  ALLOC(sl.d_nfinite, sizeof(uint32_t) * mf);
  ALLOC(sl.d_list, sizeof(uint32_t) * ((size_t)mf + 1));
  ALLOC(sl.d_list_frames, (size_t)64 * mf);
  ALLOC(sl.d_masks, sizeof(unsigned long long) * (size_t)h->crop_chunks_cap * (kCropChunk / 64));
  ALLOC(sl.d_parent, sizeof(uint32_t) * np);
  ALLOC(sl.d_count, sizeof(uint32_t) * np);
  ALLOC(sl.d_hash_head, sizeof(uint32_t) * (size_t)mf * kClusterHashSize);
  ALLOC(sl.d_hash_next, sizeof(uint32_t) * np);
  ALLOC(sl.d_partial, sizeof(GridPartial) * (size_t)mf * h->max_theta);
  ALLOC(sl.d_partial2, sizeof(GridPartial) * (size_t)mf * h->max_theta);
  ALLOC(sl.d_partial3, sizeof(GridPartial) * (size_t)mf * h->max_theta);
  ALLOC(sl.d_partial4, sizeof(GridPartial) * (size_t)mf * 4);
  ALLOC(sl.d_solverec, sizeof(SolveRec) * 2 * (size_t)mf);
  ALLOC(sl.d_bound, sizeof(uint32_t) * mf);
  ALLOC(sl.d_bound_sub, sizeof(uint32_t) * mf);
  ALLOC(sl.d_tie_count, sizeof(uint32_t) * mf);
  ALLOC(sl.d_tie_list, sizeof(GridPartial) * (size_t)mf * kTieCap);
  ALLOC(sl.d_iters, sizeof(unsigned long long) * kBatchWords);
#undef ALLOC
  HIP_TRY(h, hipHostMalloc((void**)&sl.h_res, sizeof(ilcc_result) * mf, hipHostMallocDefault));
  HIP_TRY(h, hipHostMalloc((void**)&sl.h_rec, sizeof(float) * (size_t)mf * (ILCC_RECORD_HEADER + 3 * ILCC_MAX_CORNERS), hipHostMallocDefault));
  HIP_TRY(h, hipHostMalloc((void**)&sl.h_iters, sizeof(unsigned long long) * kBatchWords, hipHostMallocDefault));
  // (the records and counters are written by kernels, not by D2H copies: the device's view of the two pinned buffers)
  HIP_TRY(h, hipHostGetDevicePointer((void**)&sl.h_rec_dev, sl.h_rec, 0));
  HIP_TRY(h, hipHostGetDevicePointer((void**)&sl.h_iters_dev, sl.h_iters, 0));
  HIP_TRY(h, hipHostMalloc((void**)&sl.h_off, sizeof(uint64_t) * (mf + 1), hipHostMallocDefault));
  HIP_TRY(h, hipHostMalloc((void**)&sl.h_online, sizeof(uint32_t) * 2 * (size_t)mf, hipHostMallocDefault));
  {
    const int32_t st = size_group_prepass(h, sl);
    if (st != ILCC_OK) return st;
  }
  sl.allocated = true;
  return ILCC_OK;
}

Ctx make_ctx(ilcc_handle* h, Slot& sl, const float4* d_xyzi, const float* d_clicks, uint32_t n_frames,
             uint32_t crop_chunks) {
  Ctx c{};
  c.xyzi = d_xyzi;
  c.off = sl.d_off;
  c.clicks = d_clicks;
  c.n_frames = n_frames;
  c.crop_chunks = crop_chunks;
  c.res = sl.d_res;
  c.roi = sl.d_roi;
  c.cluster = sl.d_cluster;
  c.board = sl.d_board;
  c.pca = sl.d_pca;
  c.optim = sl.d_optim;
  c.yz = sl.d_yz;
  c.lab = sl.d_lab;
  c.cls = sl.d_cls;
  c.n_lab = sl.d_nlab;
  c.walk_stride = sl.d_walk;
  c.walk_yz = sl.d_walk_yz;
  c.walk_lab = sl.d_walk_lab;
  c.walk_mi = sl.d_walk_mi;
  c.walk_nrim = sl.d_walk_nrim;
  c.crop_counts = sl.d_counts;
  c.crop_masks = sl.d_masks;
  c.uf_parent = sl.d_parent;
  c.uf_count = sl.d_count;
  c.uf_hash_head = sl.d_hash_head;
  c.uf_hash_next = sl.d_hash_next;
  c.cluster_lds_points = h->cluster_lds_points;
  c.cluster_cells_cap = h->cluster_cells_cap;
  c.wide = 0u;
  c.cluster_bits = cluster_bits_default();
  c.online_tier = 0u;
  c.online_window = 0.f;
  c.frame_flags = sl.d_flags;
  c.n_finite = sl.d_nfinite;
  c.crop_fin = sl.d_counts_fin;
  c.list_count = sl.d_list;
  c.list = sl.d_list + 1;
  c.list_frames = sl.d_list_frames;
  c.list_grid = h->list_grid;
  c.partial = sl.d_partial;
  c.solve_rec = sl.d_solverec;
  c.grid_blocks = (uint32_t)h->p.n_th;
  c.grid_lds_points = h->grid_lds_points;
  c.grid_bound = sl.d_bound;
  c.grid_bound_sub = sl.d_bound_sub;
  c.walk_limit = 0;
  c.tie_count = nullptr;   // only the full pass of K6 collects; K7a gets the pointers below
  c.tie_count_all = sl.d_tie_count;
  c.tie_list = sl.d_tie_list;
  c.grid_iters = sl.d_iters;
  c.seed_partial = nullptr;
  c.seed_blocks = 0;
  c.seed_n_ty = c.seed_n_tz = 1;
  c.seed_stride_t = 1;
  c.seed_stride_th = 1;
  c.seed_off_th = 0;
  c.refine_radius_th = 0;
  c.refine_step_th = 1;
  c.refine_window = 0;
  c.cth = h->d_cth;
  c.sth = h->d_sth;
  c.ay = h->d_ay;
  c.az = h->d_az;
  c.th_lattice = h->d_th_lattice;
  c.th_lat_lo = h->th_lat_lo;
  c.th_lat_hi = h->th_lat_hi;
  c.refine_hop_y = h->hop_y;
  c.refine_hop_z = h->hop_z;
  c.p = h->p;
  c.c_th = near_zero_index(h->p.th_min, h->p.th_step, h->p.n_th);
  c.c_ty = near_zero_index(h->p.ty_min, h->p.ty_step, h->p.n_ty);
  c.c_tz = near_zero_index(h->p.tz_min, h->p.tz_step, h->p.n_tz);
  return c;
}

int32_t check_offsets(ilcc_handle* h, const uint64_t* offsets, uint32_t n_frames) {
  if (!offsets || n_frames == 0) {
    h->err = "no frames";
    return ILCC_BAD_ARGUMENT;
  }
  if (n_frames > h->max_frames) {
    h->err = "n_frames exceeds the handle's max_frames";
    return ILCC_CAPACITY;
  }
  if (offsets[0] != 0) {
    h->err = "offsets[0] must be 0";
    return ILCC_BAD_ARGUMENT;
  }
  for (uint32_t f = 0; f < n_frames; ++f)
    if (offsets[f + 1] < offsets[f] || offsets[f + 1] - offsets[f] > 0x7FFFFFFFull) {
      h->err = "offsets must be non-decreasing, frames < 2^31 points";
      return ILCC_BAD_ARGUMENT;
    }
  if (offsets[n_frames] > h->max_points) {
    h->err = "total points exceed the handle's max_total_points";
    return ILCC_CAPACITY;
  }
  return ILCC_OK;
}

int32_t enqueue_impl(ilcc_handle* h, int si, const float4* d_xyzi, const uint64_t* offsets, uint32_t n_frames, const float* d_clicks,
                     bool front_only, bool no_crop);

// K2's workgroup keeps bitmap + per-cell arrays + (up to cluster_lds_points) sorted points in LDS: when the cell arrays have
// grown large (dense clouds), give up LDS points first -- frames of that size sort into HBM anyway
void fit_cluster_lds(ilcc_handle* h) {
  while (cluster_lds_bytes(h->cluster_lds_points, h->cluster_cells_cap, cluster_bits_online()) > 156u * 1024u && h->cluster_lds_points > 0)
    h->cluster_lds_points = h->cluster_lds_points > 512u ? h->cluster_lds_points - 512u : 0u;
}

uint32_t board_corners(const ilcc_params& p) { return (uint32_t)((p.board_w - 1) * (p.board_h - 1)); }
size_t trimmed_bytes(uint32_t n_corners) { return offsetof(ilcc_result, corners) + sizeof(float) * 3 * (size_t)n_corners; }

// per frame only the record's head and the board's corners (a pitched copy: rows of trimmed_bytes out of sizeof(ilcc_result))
hipError_t copy_results_trimmed(ilcc_result* dst, const ilcc_result* d_src, uint32_t n_frames, uint32_t n_corners, hipStream_t s) {
  if (n_frames == 0) return hipSuccess;
  return hipMemcpy2DAsync(dst, sizeof(ilcc_result), d_src, sizeof(ilcc_result), trimmed_bytes(n_corners), n_frames,
                          hipMemcpyDeviceToHost, s);
}

// enqueue the whole path for one batch on the slot's stream (no host synchronisation).  On a mid-pipeline failure the
// kernels already queued may still be running on the slot's buffers: wait for them before handing the error back, so
// that the next submit can reuse the slot.
int32_t enqueue(ilcc_handle* h, int si, const float4* d_xyzi, const uint64_t* offsets, uint32_t n_frames,
                const float* d_clicks, bool front_only = false, bool no_crop = false) {
  if (h->poisoned) {
    h->err = "handle unusable: a failed ilcc_set_params could not restore the device tables";
    return ILCC_HIP_ERROR;
  }
  const int32_t st = enqueue_impl(h, si, d_xyzi, offsets, n_frames, d_clicks, front_only, no_crop);
  if (st != ILCC_OK && h->slots[si].stream) {
    const std::string why = h->err;
    (void)hipStreamSynchronize(h->slots[si].stream);
    (void)hipGetLastError();
    h->slots[si].busy = false;
    h->err = why;
  }
  return st;
}

// Host-side constants of the grid search.  (A/B builds override the #defines with -D, tools/build_variant.sh; what each was
// measured at is recorded in DESIGN.md section 4 "K6" and profiles/README.md.)
#ifndef ILCC_SEED_POINTS
#define ILCC_SEED_POINTS 128   // seed and refinement walk a PREFIX of the frame's walk: a sixteenth of its labelled points (kSeedShift), at least this many
#endif
#ifndef ILCC_BOX_POINTS
#define ILCC_BOX_POINTS 48     // rim points a (frame, theta) workgroup's own box pre-pass looks at, at least
#endif
#ifndef ILCC_K6_CHAIN
#define ILCC_K6_CHAIN 1        // 0: the full passes of different batches may share the chip (measured slower every round)
#endif
constexpr int kAnchorRadius = 1;   // the anchor scores 2 * 1 + 1 thetas around the refinement's argmin, one 4 x 4 tile each, on ALL points
#ifndef ILCC_ANCHOR_ROUNDS
#define ILCC_ANCHOR_ROUNDS 2
#endif
constexpr int kAnchorRounds = ILCC_ANCHOR_ROUNDS;   // k6_anchor rounds, each re-centred on the previous one's argmin (config 5: 62.5 -> 70.6 k frames/s)

// where the full pass (and the common pre-pass) find the records of the launch that published the frame's bound: `blocks` records
// per frame from a launch over the FULL tables
void seeded_by_full_table_records(Ctx& full, const ilcc_handle* h, GridPartial* records, uint32_t blocks) {
  full.seed_partial = records;
  full.seed_blocks = blocks;
  full.seed_n_ty = h->p.n_ty;
  full.seed_n_tz = h->p.n_tz;
  full.seed_stride_t = 1;
  full.seed_stride_th = 1;
  full.seed_off_th = 0;
}

// Locating the minimum with three kinds of launches (batches too small for k6_locate's one workgroup per frame): seed over the
// decimated tables, refinement around its argmin (both on the walk's prefix, with a bound word of their own: their sums are not
// costs of complete candidates), then kAnchorRounds anchor rounds on every point, which publish the frame's real bound.
int32_t enqueue_locate_launches(ilcc_handle* h, Slot& sl, const Ctx& c, hipStream_t s, Ctx& full) {
  Ctx seed = c;
  seed.cth = h->d_cth2;
  seed.sth = h->d_sth2;
  seed.ay = h->d_ay2;
  seed.az = h->d_az2;
  seed.p.n_th = h->n_th2;
  seed.p.n_ty = h->n_ty2;
  seed.p.n_tz = h->n_tz2;
  seed.grid_blocks = (uint32_t)h->n_th2;
  seed.partial = sl.d_partial2;
  // the "nearest to zero" indices of the decimated tables (the kernel reads ay[c_ty] / az[c_tz] for its rim test and
  // uses all three for the tie-break distance: they must index THIS launch's tables, not the full ones)
  seed.c_th = std::min(std::max(c.c_th / std::max(1, h->seed_stride_th), 0), h->n_th2 - 1);
  seed.c_ty = std::min(c.c_ty / std::max(1, h->seed_stride_t), h->n_ty2 - 1);
  seed.c_tz = std::min(c.c_tz / std::max(1, h->seed_stride_t), h->n_tz2 - 1);
  seed.walk_limit = (uint32_t)ILCC_SEED_POINTS;
  seed.grid_bound = sl.d_bound_sub;
  launch_grid_cost(seed, s, /*use_oob=*/1, nullptr, true);
  HIP_TRY(h, hipEventRecord(sl.k6ev[1], s));
  // refinement: every kRefineThetaStride-th theta within a third of a seed stride of the seed's argmin, 8 x 8 translations
  Ctx refine = c;
  refine.seed_partial = sl.d_partial2;
  refine.seed_blocks = (uint32_t)h->n_th2;
  refine.seed_n_ty = h->n_ty2;
  refine.seed_n_tz = h->n_tz2;
  refine.seed_stride_t = h->seed_stride_t;
  refine.seed_stride_th = h->seed_stride_th;
  refine.seed_off_th = h->seed_stride_th / 2;
#ifndef ILCC_REFINE_WIDE
#define ILCC_REFINE_WIDE 1
#endif
  refine.refine_window = (ILCC_REFINE_WIDE && h->seed_stride_t > 2 * kTileA) ? 4 : 2;   // tiles per axis: the window spans the seed's stride
  refine.refine_radius_th = (std::max(1, h->seed_stride_th / ILCC_REFINE_RADIUS_DIV) / kRefineThetaStride) * kRefineThetaStride;
  refine.refine_step_th = kRefineThetaStride;
  refine.grid_blocks = std::min((uint32_t)(2 * (refine.refine_radius_th / kRefineThetaStride) + 1), h->max_theta);
  refine.partial = sl.d_partial3;
  refine.walk_limit = (uint32_t)ILCC_SEED_POINTS;
  refine.grid_bound = sl.d_bound_sub;
  launch_grid_cost(refine, s, /*use_oob=*/1, nullptr, true);
  HIP_TRY(h, hipEventRecord(sl.k6ev[2], s));
  // anchor rounds (k6_anchor): 3 thetas x one tile on ALL points; every further round re-centres the tile on the previous
  // round's argmin -- a greedy descent on complete costs towards the grid minimum, for a tighter bound in front of the full pass
  Ctx anchor = c;
  seeded_by_full_table_records(anchor, h, sl.d_partial3, refine.grid_blocks);
  anchor.refine_radius_th = kAnchorRadius;
  anchor.grid_blocks = 2 * kAnchorRadius + 1;
  GridPartial* const ping[2] = {sl.d_partial4, sl.d_partial2};
  for (int round = 0; round < kAnchorRounds; ++round) {
    anchor.partial = ping[round & 1];
    launch_anchor(anchor, s);
    anchor.seed_partial = ping[round & 1];
    anchor.seed_blocks = 2 * kAnchorRadius + 1;
  }
  seeded_by_full_table_records(full, h, ping[(kAnchorRounds - 1) & 1], 2 * kAnchorRadius + 1);
  return ILCC_OK;
}

// The grid search of one batch on its stream: K5w (walk layout), the launches that locate the minimum and publish the frame's bound
// (k6_locate, or seed + refinement + anchor rounds), the common pre-pass, the full pass -- with the slot's K6 events recorded in
// between.  chain: the full pass waits for the previous batch's (the pipeline; the diagnostic entry ilcc_grid_solve runs alone).
int32_t enqueue_grid_search(ilcc_handle* h, Slot& sl, int si, const Ctx& c, hipStream_t s, uint32_t n_frames, bool chain) {
  const bool prune = h->p.grid_prune != 0;
  launch_walk_order(c, s);   // K5w: the labelled points in K6's walk layout, once per frame
  HIP_TRY(h, hipEventRecord(sl.k6ev[0], s));
  Ctx full = c;
  // (grid_prune = 0 keeps the locate launches: they only initialise the frame's bound, which the cut-free full
  // pass still needs to recognise near ties; it never cuts a tile)
  if (h->n_th2 > 0 && h->p.n_th >= 8 && h->p.n_ty >= 8 && h->p.n_tz >= 8) {
    // These launches only have to LOCATE the minimum.  Seed and refinement therefore look at a prefix of the point walk (an
    // eighth of the frame's labelled points, at least ILCC_SEED_POINTS positions: a uniform sample of the board) and keep their
    // own bound word; the anchor evaluates what they found on EVERY point and publishes the frame's real bound.  (Round 2 ran
    // seed and refinement on all points: 16 % of the path's VALU instructions.)
    LocatePlan lp{};
    lp.cth2 = h->d_cth2;
    lp.sth2 = h->d_sth2;
    lp.ay2 = h->d_ay2;
    lp.az2 = h->d_az2;
    lp.n_th2 = h->n_th2;
    lp.n_ty2 = h->n_ty2;
    lp.n_tz2 = h->n_tz2;
    lp.c_th2 = std::min(std::max(c.c_th / std::max(1, h->seed_stride_th), 0), h->n_th2 - 1);
    lp.c_ty2 = std::min(c.c_ty / std::max(1, h->seed_stride_t), h->n_ty2 - 1);
    lp.c_tz2 = std::min(c.c_tz / std::max(1, h->seed_stride_t), h->n_tz2 - 1);
    lp.stride_th = h->seed_stride_th;
    lp.off_th = h->seed_stride_th / 2;
    lp.stride_t = h->seed_stride_t;
    lp.refine_radius = std::max(1, h->seed_stride_th / ILCC_REFINE_RADIUS_DIV);
    lp.sample_min = (uint32_t)ILCC_SEED_POINTS;
    lp.sample_cap = std::max((uint32_t)ILCC_SEED_POINTS, ((c.grid_lds_points >> 3) + 63u) & ~63u);
    lp.out = sl.d_partial4;
    // one launch, one workgroup per frame (k6_locate) when the batch has the frames to fill the chip that way
    const bool fused = n_frames >= (uint32_t)kLocateMinFrames && h->n_th2 <= 16 &&
                       locate_lds_bytes(lp.sample_cap, h->p.n_ty, h->p.n_tz, h->n_ty2, h->n_tz2) <= 60u * 1024u;
    if (fused) {
      launch_locate(c, s, lp);
      HIP_TRY(h, hipEventRecord(sl.k6ev[1], s));   // (the whole locate launch is accounted as the "seed" span)
      HIP_TRY(h, hipEventRecord(sl.k6ev[2], s));
      seeded_by_full_table_records(full, h, sl.d_partial4, 1);
    } else {
      const int32_t st = enqueue_locate_launches(h, sl, c, s, full);
      if (st != ILCC_OK) return st;
    }
  } else {
    HIP_TRY(h, hipEventRecord(sl.k6ev[1], s));
    HIP_TRY(h, hipEventRecord(sl.k6ev[2], s));
  }
  HIP_TRY(h, hipEventRecord(sl.k6ev[3], s));   // behind the anchor: the common pre-pass gets an event span of its own
  const GroupPrepassPlan gp = group_prepass_plan(h->p);
  full.box_points = gp.box ? (uint32_t)ILCC_BOX_POINTS : 0u;   // (eligibility: group_prepass_plan, which also sized the buffers)
  // k6_group_prepass: one box pre-pass for kThetaGroup consecutive thetas, launched HERE -- behind the anchor (it needs the
  // frame's bound), in front of the wait for the previous batch's full pass, so it runs beside that pass like the other small
  // launches.  Its buffers were sized for (max_frames, this grid) by alloc_slot / ilcc_set_params.
  if (full.box_points != 0u && gp.on && (size_t)n_frames * gp.groups <= sl.grp_alive_cap &&
      (size_t)n_frames * gp.groups * gp.words <= sl.grp_mask_cap) {
    full.grp_count = gp.groups;
    full.grp_words = gp.words;
    launch_group_prepass(full, s, sl.d_grp_alive, sl.d_grp_mask);
    full.grp_alive = sl.d_grp_alive;
    full.grp_mask = sl.d_grp_mask;
  } else if (full.box_points != 0u && gp.on) {
    // eligible grid, buffers too small for this batch (never expected: size_group_prepass sized them): slower, not wrong -- say so once
    if (h->group_prepass_skipped++ == 0) h->err = "note: k6_group_prepass skipped (mask buffers smaller than this batch needs): slower, results unaffected";
  }
  HIP_TRY(h, hipEventRecord(sl.ev[7], s));
  // The FULL passes of different slots are chained so that they never share the chip (two passes side by side both run at half
  // speed and every batch's front end waits longer for wave slots); the launches above are small and are left free to overlap
  // with another batch's full pass, like K2 / K3 / K7.
  if (chain && ILCC_K6_CHAIN && h->k6_last >= 0 && h->k6_last != si && h->slots[h->k6_last].busy)
    HIP_TRY(h, hipStreamWaitEvent(s, h->slots[h->k6_last].k6_done, 0));
  HIP_TRY(h, hipEventRecord(sl.ev[8], s));
  full.tie_count = sl.d_tie_count;
  launch_grid_cost(full, s, /*use_oob=*/1, nullptr, prune);
  HIP_TRY(h, hipEventRecord(sl.k6_done, s));
  if (chain) h->k6_last = si;
  return ILCC_OK;
}

int32_t enqueue_impl(ilcc_handle* h, int si, const float4* d_xyzi, const uint64_t* offsets, uint32_t n_frames,
                     const float* d_clicks, bool front_only, bool no_crop) {
  Slot& sl = h->slots[si];
  uint64_t max_n = 0;
  for (uint32_t f = 0; f < n_frames; ++f) max_n = std::max<uint64_t>(max_n, offsets[f + 1] - offsets[f]);
  uint32_t chunks = (uint32_t)((max_n + kCropChunk - 1) / kCropChunk);
  if (chunks == 0) chunks = 1;
  if ((uint64_t)chunks * n_frames > (uint64_t)h->crop_chunks_cap) {
    h->err = "crop chunk table too small for this (very ragged) batch";
    return ILCC_CAPACITY;
  }
  sl.off.assign(offsets, offsets + n_frames + 1);
  sl.n_frames = n_frames;
  hipStream_t s = sl.stream;
  std::memcpy(sl.h_off, sl.off.data(), sizeof(uint64_t) * (n_frames + 1));
  HIP_TRY(h, hipMemcpyAsync(sl.d_off, sl.h_off, sizeof(uint64_t) * (n_frames + 1), hipMemcpyHostToDevice, s));
  // (result records, component counters, K6 counters and near-tie counters are reset inside K1 / K2)
  Ctx c = make_ctx(h, sl, d_xyzi, d_clicks, n_frames, chunks);
  HIP_TRY(h, hipEventRecord(sl.ev[0], s));
  if (!no_crop) {
    launch_roi_crop(c, s, sl.ev[9]);   // ev[9]: between the count pass and the scatter
    HIP_TRY(h, hipEventRecord(sl.ev[1], s));
    launch_cluster(c, s);
  } else {
    // get_chessboard_by_point clusters the WHOLE cloud (no ROI; setClusterTolerance(0.1), LidarCornersEst.cpp:80 -- EuclideanCluster()
    // uses 0.12, :131) and keeps the cluster around the predicted point.  Two tiers, identical results:
    //   1. K1 crops a +-1.25 m window around the point, K2 clusters it on the LDS cell grid -- the ROI pipeline's fast path -- and
    //      verifies that the whole cloud gives the same cluster (fine_cluster_frame: nearest point, admissible, complete);
    //   2. the frames it could not vouch for: K1 with an unbounded box (only non-finite points go), K2 on hashed cells.
    c.p.cluster_tol = c.p.online_cluster_tol;
    c.wide = 1u;   // a synchronous call: latency is all that counts
    Ctx t1 = c;
    t1.online_tier = 1u;
    t1.online_window = 1.25f;
    t1.p.roi_half[0] = t1.p.roi_half[1] = t1.p.roi_half[2] = (double)t1.online_window;
    t1.cluster_bits = cluster_bits_online();
    launch_roi_crop(t1, s, sl.ev[9]);
    HIP_TRY(h, hipEventRecord(sl.ev[1], s));
    launch_cluster(t1, s);
    Ctx t2 = c;
    t2.online_tier = 2u;
    t2.p.roi_half[0] = t2.p.roi_half[1] = t2.p.roi_half[2] = (double)INFINITY;
    launch_roi_crop(t2, s, nullptr);
    launch_cluster(t2, s);
    HIP_TRY(h, hipMemcpyAsync(sl.h_online, sl.d_flags, sizeof(uint32_t) * n_frames, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(sl.h_online + h->max_frames, sl.d_nfinite, sizeof(uint32_t) * n_frames, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(h, hipEventRecord(sl.ev[2], s));
  launch_ransac_plane(c, s);
  HIP_TRY(h, hipEventRecord(sl.ev[3], s));
  launch_plane_frame_hist(c, s);
  sl.grid = !front_only && h->p.solver == ILCC_SOLVER_GRID;
  HIP_TRY(h, hipEventRecord(sl.ev[4], s));
  if (sl.grid) {
    const int32_t st = enqueue_grid_search(h, sl, si, c, s, n_frames, /*chain=*/true);
    if (st != ILCC_OK) return st;
  }
  HIP_TRY(h, hipEventRecord(sl.ev[5], s));
  if (!front_only) {
    if (sl.grid) {
      Ctx c7 = c;
      c7.tie_count = sl.d_tie_count;   // near ties of the full pass, re-ordered on fixed-point sums by K7r
      launch_pattern_refine_corners(c7, s);
    } else {
      launch_refine_corners(c, s);
    }
  }
  HIP_TRY(h, hipEventRecord(sl.ev[6], s));
  HIP_TRY(h, hipGetLastError());
  // the copy back: what the result mode asks for rides on the batch's stream (include/ilcc_hip.h, "Result traffic")
  sl.compact = h->result_mode == ILCC_RESULTS_COMPACT;
  sl.rec_corners = board_corners(h->p);
  // (compact records and the batch counters are STORED into the pinned staging by kernels: no D2H command in the SDMA queue that
  // carries the next batches' input copies -- launch_store_to_host)
  if (sl.compact) {
    launch_pack_records(sl.d_res, n_frames, sl.rec_corners, 0u, sl.h_rec_dev, s);   // K9 writes the records straight into the pinned staging
  } else {
    HIP_TRY(h, copy_results_trimmed(sl.h_res, sl.d_res, n_frames, sl.rec_corners, s));
  }
  launch_store_to_host(sl.d_iters, sl.h_iters_dev, sizeof(unsigned long long) * kBatchWords, s);
  HIP_TRY(h, hipGetLastError());
  sl.busy = true;
  sl.online = no_crop;
  return ILCC_OK;
}

// the slot's full records in its pinned staging (they are there already unless the batch ran in ILCC_RESULTS_COMPACT mode)
int32_t ensure_full(ilcc_handle* h, Slot& sl) {
  if (sl.h_res_valid) return ILCC_OK;
  HIP_TRY(h, copy_results_trimmed(sl.h_res, sl.d_res, sl.n_frames, sl.rec_corners, sl.stream));
  HIP_TRY(h, hipStreamSynchronize(sl.stream));
  sl.h_res_valid = true;
  return ILCC_OK;
}

// wait for the slot's batch, hand the records over (full and / or compact), account the timing
int32_t finish(ilcc_handle* h, int si, ilcc_result* out, float* out_compact = nullptr, float* d_records = nullptr,
               uint32_t n_corners = 0, uint32_t tag_base = 0) {
  Slot& sl = h->slots[si];
  if (d_records) launch_pack_records(sl.d_res, sl.n_frames, n_corners, tag_base, d_records, sl.stream);   // same stream: after K7
  const uint32_t n_frames = sl.n_frames;
  const size_t rec_w = (size_t)ILCC_RECORD_HEADER + 3 * (size_t)sl.rec_corners;
  if (out_compact && !sl.compact) {   // not enqueued with the batch: pack and copy now
    launch_pack_records(sl.d_res, n_frames, sl.rec_corners, 0u, sl.h_rec_dev, sl.stream);
  }
  HIP_TRY(h, hipStreamSynchronize(sl.stream));
  sl.busy = false;
  sl.h_res_valid = !sl.compact;
  h->last_slot = si;
  if (out) {
    const int32_t st = ensure_full(h, sl);
    if (st != ILCC_OK) return st;
    const size_t row = trimmed_bytes(sl.rec_corners);
    for (uint32_t f = 0; f < n_frames; ++f) std::memcpy(&out[f], &sl.h_res[f], row);
  }
  if (out_compact) std::memcpy(out_compact, sl.h_rec, sizeof(float) * n_frames * rec_w);
  float ms[6];
  for (int k = 0; k < 6; ++k) HIP_TRY(h, hipEventElapsedTime(&ms[k], sl.ev[k], sl.ev[k + 1]));
  float k6k[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // K5w, seed, refinement, anchor, common pre-pass, full pass
  if (sl.grid) {   // K6 = (seed + refinement passes) + (full pass); the wait for the previous batch's full pass in between is not K6 time
    float pre = 0.f, fullp = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&pre, sl.ev[4], sl.ev[7]));
    HIP_TRY(h, hipEventElapsedTime(&fullp, sl.ev[8], sl.ev[5]));
    ms[4] = pre + fullp;
    HIP_TRY(h, hipEventElapsedTime(&k6k[0], sl.ev[4], sl.k6ev[0]));
    HIP_TRY(h, hipEventElapsedTime(&k6k[1], sl.k6ev[0], sl.k6ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&k6k[2], sl.k6ev[1], sl.k6ev[2]));
    HIP_TRY(h, hipEventElapsedTime(&k6k[3], sl.k6ev[2], sl.k6ev[3]));
    HIP_TRY(h, hipEventElapsedTime(&k6k[4], sl.k6ev[3], sl.ev[7]));
    k6k[5] = fullp;
  }
  float tot = 0;
  HIP_TRY(h, hipEventElapsedTime(&tot, sl.ev[0], sl.ev[6]));
  if (h->tl_on && h->tl_ref && sl.grid) {
    const hipEvent_t evs[ILCC_TIMELINE_COLS - 1] = {sl.ev[0], sl.ev[9], sl.ev[1], sl.ev[2], sl.ev[3], sl.ev[4], sl.k6ev[0], sl.k6ev[1],
                                                    sl.k6ev[2], sl.k6ev[3], sl.ev[7], sl.ev[8], sl.ev[5], sl.ev[6]};
    h->tl_rows.push_back((double)si);
    for (hipEvent_t e : evs) {
      float t = 0.f;
      HIP_TRY(h, hipEventElapsedTime(&t, h->tl_ref, e));
      h->tl_rows.push_back((double)t);
    }
  }
  ilcc_timing& t = h->timing;
  t.roi_crop = ms[0];
  t.cluster = ms[1];
  t.ransac_plane = ms[2];
  t.plane_frame_hist = ms[3];
  t.grid_cost = ms[4];
  t.refine_corners = ms[5];
  t.total = tot;
  t.batches += 1;
  for (int k = 0; k < 6; ++k) t.stage_ms_sum[k] += ms[k];
  t.stage_ms_sum[6] += tot;
  {
    float cnt = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&cnt, sl.ev[0], sl.ev[9]));
    t.roi_count_ms_sum += cnt;
  }
  uint32_t max_lab = 0, max_roi = 0;
  uint64_t evals = 0;
  for (uint32_t f = 0; f < n_frames; ++f) {
    // (status, n_roi, labelled points) from whichever record came back with the batch
    int32_t status, n_roi;
    uint32_t m;
    if (sl.h_res_valid) {
      const ilcc_result& r = sl.h_res[f];
      status = r.status;
      n_roi = r.n_roi;
      m = (uint32_t)(r.n_black + r.n_white);
    } else {
      const float* r = sl.h_rec + (size_t)f * rec_w;
      status = (int32_t)r[0];
      n_roi = (int32_t)r[19];
      m = (uint32_t)r[13] + (uint32_t)r[14];
    }
    if (n_roi > 0 && (uint32_t)n_roi <= (uint32_t)kClusterLdsPointsMax) max_roi = std::max(max_roi, (uint32_t)n_roi);
    if (status != ILCC_OK && status != ILCC_AMBIGUOUS) continue;
    max_lab = std::max(max_lab, m);
    evals += (uint64_t)m * (uint64_t)h->p.n_th * h->p.n_ty * h->p.n_tz;
  }
  if (sl.grid) {
    t.grid_cost_launches += 1;
    t.grid_cost_ms_sum += ms[4];
    t.walk_order_ms_sum += k6k[0];
    t.grid_cost_kernel_ms_sum += (double)k6k[1] + k6k[2] + k6k[3] + k6k[4] + k6k[5];
    t.grid_cost_locate_ms_sum += (double)k6k[1] + k6k[2] + k6k[3];
    t.grid_cost_prepass_ms_sum += k6k[4];
    t.grid_cost_full_ms_sum += k6k[5];
    t.grid_cost_evals_nominal_sum += evals;
    // (both colour phases of a (point, candidate) pair = 1 evaluation)
    unsigned long long iters = 0;
    unsigned long long iters_in = 0;
    for (int k = 0; k < kIterSlots; ++k) {
      iters += sl.h_iters[k];
      iters_in += sl.h_iters[kIterSlots + k];
      t.grid_cost_box_evals_sum += sl.h_iters[2 * kIterSlots + k];
    }
    t.grid_cost_evals_sum += (uint64_t)iters * grid_cost_evals_per_count();
    t.grid_cost_evals_interior_sum += (uint64_t)iters_in * grid_cost_evals_per_count();
  }
  // adapt the K6 / K7 LDS staging size to the labelled-point counts actually seen (later calls): the largest count so far,
  // rounded up to 256 points.  (Not to a power of two: 1 781 points -- the largest of the bench frames -- need 21.8 KB per K6
  // workgroup at 1 792 and 24.9 KB at 2 048: seven instead of six workgroups per CU -- which, the kernel being VALU-bound,
  // measured no difference: 248.8 k vs 250 k frames/s.)
  // ... but never, on its own, past the capacity at which TWO workgroups of the K6 full pass still share a CU's 160 KB (12 bytes per
  // staged point + the (ty, tz) tables + ~3 KB of static LDS: 6 400 points for both BASELINE grids).  Round 6: a stream of 64-ring
  // frames whose closest boards hold 6 682 labelled points (the first 128 frames: 6 170) grew the staging to 6 912 points, ONE
  // workgroup per CU, and every frame's full pass took 1.15 ms instead of 0.74 ms per 128 frames.  The few frames above the capacity
  // walk their points through L2 (grid_cost_body<LDS_POINTS = false>); ilcc_reserve may still ask for more, explicitly.
  const uint32_t two_per_cu = (uint32_t)(((160u * 1024u / 2u) - 3072u - 4u * (uint32_t)(h->p.n_ty + h->p.n_tz)) / 12u) & ~255u;
  uint32_t want = std::min<uint32_t>((uint32_t)kGridLdsPointsMax, std::max<uint32_t>(1024u, (max_lab + 255u) & ~255u));
  want = std::min<uint32_t>(want, std::max<uint32_t>(1024u, two_per_cu));
  if (want > h->grid_lds_points) h->grid_lds_points = want;
  // the same for K2's one-workgroup LDS path (ROI points per frame, steps of 512): what it does not hold of a CU's 160 KiB
  // is room for K6 workgroups of other batches
  want = std::min<uint32_t>((uint32_t)kClusterLdsPointsMax, (max_roi + 511u) & ~511u);
  if (want > h->cluster_lds_points) h->cluster_lds_points = want;
  // ... and for its per-cell arrays: the most occupied cells a frame of this batch needed (+ 1/8), steps of 256.  Frames the LDS
  // cell grid cannot hold at any capacity (bounding grid too large: un-cropped clouds; more cells than the largest capacity) take
  // the hashed-cell path of their own workgroup (k2_cluster.hip).
  const uint64_t cells_needed = sl.h_iters[3 * kIterSlots];
  want = (uint32_t)std::min<uint64_t>((uint64_t)kClusterCellsMax, (cells_needed + cells_needed / 8 + 255u) & ~255ull);
  if (want > h->cluster_cells_cap) h->cluster_cells_cap = want;
  fit_cluster_lds(h);
  return ILCC_OK;
}

}  // namespace

namespace ilcc {
// text behind ilcc_last_error(NULL) for the handle-less entry points in other translation units
void set_global_error(const std::string& s) { g_err = s; }
}  // namespace ilcc

extern "C" {

int32_t ilcc_abi_version(void) { return ILCC_ABI_VERSION; }

const char* ilcc_strerror(int32_t status) {
  switch (status) {
    case ILCC_OK: return "ok";
    case ILCC_NO_ROI_POINTS: return "no points inside the ROI box around the click";
    case ILCC_NO_CLUSTER: return "no Euclidean cluster of admissible size";
    case ILCC_NO_PLANE: return "could not estimate a planar model";
    case ILCC_DEGENERATE_HIST: return "intensity histogram is degenerate (no bin on one side of the mean)";
    case ILCC_TOO_FEW_POINTS: return "too few points";
    case ILCC_BAD_ARGUMENT: return "bad argument";
    case ILCC_CAPACITY: return "handle capacity exceeded";
    case ILCC_HIP_ERROR: return "HIP runtime error";
    case ILCC_IO_ERROR: return "file I/O error";
    case ILCC_BOARD_NOT_FOUND: return "no chessboard plane of sufficient size around the given point";
    case ILCC_AMBIGUOUS: return "board position ambiguous: a basin one square away costs about the same";
    default: return "unknown status";
  }
}

const char* ilcc_last_error(const ilcc_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

void ilcc_default_params(ilcc_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->roi_half[0] = 1.0;
  p->roi_half[1] = 1.5;
  p->roi_half[2] = 2.0;
  p->cluster_tol = 0.12;
  p->cluster_min = 100;
  p->cluster_max = 25000;
  p->ransac_thresh = 0.03;
  p->ransac_hyp = 50;            // SACSegmentation: max_iterations_
  p->ransac_probability = 0.99;  // SACSegmentation: probability_
  p->ransac_seed = 12345u;
  p->hist_bins = 100;
  p->gray_rate = 2.5;
  p->huber_delta = 0.1;
  p->grid_length = 0.15;
  p->board_w = 6;
  p->board_h = 8;
  p->solver = ILCC_SOLVER_GRID;
  p->phase_mode = 2;
  p->max_iterations = 50;
  p->grid_prune = 1;
  const double kPi = 3.14159265358979323846;
  p->n_th = 61;
  p->th_step = 0.5 * kPi / 180.0;
  p->th_min = -15.0 * kPi / 180.0;
  p->n_ty = 40;
  p->ty_step = 0.15 / 20.0;
  p->ty_min = -0.15;
  p->n_tz = 40;
  p->tz_step = 0.15 / 20.0;
  p->tz_min = -0.15;
  p->refine_div = 16;
  p->refine_max_rounds = 64;
  p->refine_th_margin = 32;
  p->ambiguity_eps = 1.0;
  p->online_cluster_tol = 0.10;   // LidarCornersEst.cpp:80
  p->min_cell_coverage = 0.9;
}

// Minimal OpenCV-FileStorage YAML reader for the three scalar keys the path uses
// (cv::FileStorage is not available; /root/reference/ilcc2/config/pointgrey.yaml:17-19).
int32_t ilcc_set_chessboard_param(ilcc_params* p, const char* cam_yaml) {
  if (!p || !cam_yaml) return ILCC_BAD_ARGUMENT;
  std::ifstream in(cam_yaml);
  if (!in.is_open()) {
    g_err = std::string("can not open ") + cam_yaml;   // LidarCornersEst.cpp:27
    return ILCC_IO_ERROR;
  }
  double grid_length = -1;
  long cx = -1, cy = -1;
  std::string line;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    const size_t colon = line.find(':');
    if (colon == std::string::npos) continue;
    std::string key = line.substr(0, colon), val = line.substr(colon + 1);
    auto trim = [](std::string& s) {
      const size_t a = s.find_first_not_of(" \t\r\n");
      const size_t b = s.find_last_not_of(" \t\r\n");
      s = (a == std::string::npos) ? std::string() : s.substr(a, b - a + 1);
    };
    trim(key);
    trim(val);
    if (val.empty()) continue;
    char* endp = nullptr;
    if (key == "grid_length") grid_length = std::strtod(val.c_str(), &endp);
    else if (key == "corner_in_x") cx = (long)std::strtod(val.c_str(), &endp);
    else if (key == "corner_in_y") cy = (long)std::strtod(val.c_str(), &endp);
  }
  if (!(grid_length > 0) || cx < 1 || cy < 1) {
    g_err = "grid_length / corner_in_x / corner_in_y missing or invalid";
    return ILCC_BAD_ARGUMENT;
  }
  int32_t w = (int32_t)cx + 1, hh = (int32_t)cy + 1;   // :31-32
  if (w > hh) std::swap(w, hh);                        // :35-39
  // keep the default grid's meaning (one cell either way, g/20 steps) when the square size changes
  const double scale = grid_length / p->grid_length;
  p->grid_length = grid_length;
  p->board_w = w;
  p->board_h = hh;
  if (scale > 0 && scale != 1.0) {
    p->ty_min *= scale;
    p->ty_step *= scale;
    p->tz_min *= scale;
    p->tz_step *= scale;
  }
  return ILCC_OK;
}

ilcc_handle* ilcc_create(int32_t device, const ilcc_params* p, uint32_t max_frames, uint64_t max_total_points) {
  g_err.clear();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device: libilcc_hip has no CPU fallback";
    return nullptr;
  }
  if (!p || max_frames == 0 || max_total_points == 0) {
    g_err = "bad arguments to ilcc_create";
    return nullptr;
  }
  std::string why;
  if (!params_ok(*p, why)) {
    g_err = why;
    return nullptr;
  }
  ilcc_handle* h = new (std::nothrow) ilcc_handle();
  if (!h) return nullptr;
  h->p = *p;
  h->max_frames = max_frames;
  h->max_points = max_total_points;
  h->max_theta = 4096;
  h->crop_chunks_cap =
      (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, max_total_points / kCropChunk + (uint64_t)max_frames + 1);
  auto fail = [&](const std::string& what) {
    g_err = what;
    ilcc_destroy(h);
    return (ilcc_handle*)nullptr;
  };
  hipError_t e;
  if (device >= 0) {
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(std::string("hipSetDevice: ") + hipGetErrorString(e));
    h->device = device;
  } else {
    (void)hipGetDevice(&h->device);
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0)
      h->list_grid = (uint32_t)(4 * cus);
  }
  // dynamic-LDS limits are kept per (function, device): raise them for THIS device, and say so when that fails
  if ((e = set_kernel_attributes_k2()) != hipSuccess || (e = set_kernel_attributes_k6()) != hipSuccess ||
      (e = set_kernel_attributes_k7()) != hipSuccess)
    return fail(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
  float** tabs[] = {&h->d_cth, &h->d_sth, &h->d_ay, &h->d_az, &h->d_cth2, &h->d_sth2, &h->d_ay2, &h->d_az2};
  for (float** t : tabs)
    if ((e = hipMalloc((void**)t, sizeof(float) * h->max_theta)) != hipSuccess)
      return fail(std::string("hipMalloc tables: ") + hipGetErrorString(e));
  if ((e = hipMalloc((void**)&h->d_solve, sizeof(double) * 8)) != hipSuccess ||
      (e = hipMalloc((void**)&h->d_refine_io, sizeof(RefineOut))) != hipSuccess)
    return fail(std::string("hipMalloc: ") + hipGetErrorString(e));
  if (alloc_slot(h, h->slots[0]) != ILCC_OK) return fail(h->err);   // further slots on first asynchronous use
  if (upload_tables(h) != ILCC_OK) return fail(h->err);
  return h;
}

void ilcc_destroy(ilcc_handle* h) {
  if (!h) return;
  for (Slot& sl : h->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    free_slot(sl);
  }
  void* bufs[] = {h->d_cth, h->d_sth, h->d_ay, h->d_az, h->d_cth2, h->d_sth2, h->d_ay2, h->d_az2, h->d_solve, h->d_refine_io,
                  h->d_th_lattice};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (h->tl_ref) (void)hipEventDestroy(h->tl_ref);
  delete h;
}

int32_t ilcc_set_params(ilcc_handle* h, const ilcc_params* p) {
  if (!h || !p) return ILCC_BAD_ARGUMENT;
  std::string why;
  if (!params_ok(*p, why)) {
    h->err = why;
    return ILCC_BAD_ARGUMENT;
  }
  for (const Slot& sl : h->slots)
    if (sl.busy) {
      h->err = "ilcc_set_params with a batch in flight: ilcc_wait first";
      return ILCC_BAD_ARGUMENT;
    }
  int32_t st = sync_all(h);
  if (st != ILCC_OK) return st;
  const ilcc_params old = h->p;
  h->p = *p;
  st = upload_tables(h);
  for (Slot& sl : h->slots)   // the common pre-pass's buffers follow the grid (no batch is in flight here)
    if (st == ILCC_OK && sl.allocated) st = size_group_prepass(h, sl);
  if (st != ILCC_OK) {
    // the device tables may be partly overwritten: put the previous parameter set back on the device as well, and refuse
    // further work if even that fails
    const std::string why = h->err;
    h->p = old;
    if (upload_tables(h) != ILCC_OK) h->poisoned = true;
    h->err = why;
  }
  return st;
}

int32_t ilcc_reserve(ilcc_handle* h, uint32_t labelled_points_per_frame, uint32_t roi_points_per_frame) {
  if (!h) return ILCC_BAD_ARGUMENT;
  for (const Slot& sl : h->slots)
    if (sl.busy) {
      h->err = "ilcc_reserve with a batch in flight: ilcc_wait first";
      return ILCC_BAD_ARGUMENT;
    }
  const uint32_t lab = std::min<uint32_t>((uint32_t)kGridLdsPointsMax, (std::max(labelled_points_per_frame, 1u) + 255u) & ~255u);
  h->grid_lds_points = std::max(h->grid_lds_points, std::max(1024u, lab));
  const uint32_t roi = std::min<uint32_t>((uint32_t)kClusterLdsPointsMax, (std::max(roi_points_per_frame, 1u) + 511u) & ~511u);
  h->cluster_lds_points = std::max(h->cluster_lds_points, std::max((uint32_t)kClusterLdsPointsMin, roi));
  // occupied cells: a cell of side 0.57 tol holds ~4.5 points of a VLP-16 ROI and ~8 of a 64-ring one; a quarter of the
  // points is a safe capacity (a frame above it takes the slower point-level path once and the handle grows)
  const uint32_t cells = std::min<uint32_t>((uint32_t)kClusterCellsMax, (roi_points_per_frame / 4u + 255u) & ~255u);
  h->cluster_cells_cap = std::max(h->cluster_cells_cap, std::max((uint32_t)kClusterCellsMin, cells));
  fit_cluster_lds(h);
  return ILCC_OK;
}

int32_t ilcc_submit_batch_device(ilcc_handle* h, const float* d_xyzi, const uint64_t* offsets, uint32_t n_frames,
                                 const float* d_clicks, int32_t* ticket) {
  if (!h || !d_xyzi || !d_clicks || !ticket) return ILCC_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int32_t st = check_offsets(h, offsets, n_frames);
  if (st != ILCC_OK) return st;
  const int si = h->next_slot;
  Slot& sl = h->slots[si];
  if (sl.busy) {
    h->err = "every pipeline slot holds a batch: ilcc_wait for the oldest ticket first";
    return ILCC_CAPACITY;
  }
  st = alloc_slot(h, sl);
  if (st != ILCC_OK) return st;
  st = enqueue(h, si, reinterpret_cast<const float4*>(d_xyzi), offsets, n_frames, d_clicks);
  if (st != ILCC_OK) return st;
  *ticket = si;
  h->next_slot = (si + 1) % kSlots;
  return ILCC_OK;
}

int32_t ilcc_submit_batch(ilcc_handle* h, const float* xyzi, const uint64_t* offsets, uint32_t n_frames, const float* clicks,
                          int32_t* ticket) {
  if (!h || !xyzi || !clicks || !ticket) return ILCC_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int32_t st = check_offsets(h, offsets, n_frames);
  if (st != ILCC_OK) return st;
  const int si = h->next_slot;
  Slot& sl = h->slots[si];
  if (sl.busy) {
    h->err = "every pipeline slot holds a batch: ilcc_wait for the oldest ticket first";
    return ILCC_CAPACITY;
  }
  st = alloc_slot(h, sl);
  if (st != ILCC_OK) return st;
  // the copy rides on the slot's stream: ordered before this batch's K1, concurrent with every other slot's kernels
  if (offsets[n_frames] > 0)
    HIP_TRY(h, hipMemcpyAsync(sl.d_xyzi, xyzi, sizeof(float4) * offsets[n_frames], hipMemcpyHostToDevice, sl.stream));
  HIP_TRY(h, hipMemcpyAsync(sl.d_clicks, clicks, sizeof(float) * 3 * n_frames, hipMemcpyHostToDevice, sl.stream));
  st = enqueue(h, si, sl.d_xyzi, offsets, n_frames, sl.d_clicks);
  if (st != ILCC_OK) return st;
  *ticket = si;
  h->next_slot = (si + 1) % kSlots;
  return ILCC_OK;
}

// A ticket belongs to the call family that issued it: ilcc_submit_chessboard_by_point's batches skip the crop and
// the search, and only their own wait applies the by-point epilogue (plane >= min_plane_points, whole-cloud n_roi); the other way
// round that epilogue would read staging no copy of the batch has filled.  The batch stays in flight.
static int32_t wrong_ticket_kind(ilcc_handle* h, const char* fn, bool ticket_is_online) {
  h->err = std::string(fn) + (ticket_is_online ? ": this ticket came from ilcc_submit_chessboard_by_point: wait with ilcc_wait_chessboard_by_point"
                                               : ": this ticket came from ilcc_submit_batch[_device]: wait with ilcc_wait / ilcc_wait_compact / ilcc_wait_records_device");
  return ILCC_BAD_ARGUMENT;
}

int32_t ilcc_wait(ilcc_handle* h, int32_t ticket, ilcc_result* out) {
  if (!h || !out || ticket < 0 || ticket >= kSlots || !h->slots[ticket].busy) {
    if (h) h->err = "ilcc_wait: no batch in flight under this ticket";
    return ILCC_BAD_ARGUMENT;
  }
  if (h->slots[ticket].online) return wrong_ticket_kind(h, "ilcc_wait", true);
  HIP_TRY(h, hipSetDevice(h->device));
  return finish(h, ticket, out);
}

int32_t ilcc_wait_records_device(ilcc_handle* h, int32_t ticket, ilcc_result* out, void* d_records, uint32_t n_corners,
                                 uint32_t tag_base) {
  if (!h || !d_records || n_corners > ILCC_MAX_CORNERS || ticket < 0 || ticket >= kSlots || !h->slots[ticket].busy) {   // (out may be NULL: device records only)
    if (h) h->err = "ilcc_wait_records_device: bad argument or no batch in flight under this ticket";
    return ILCC_BAD_ARGUMENT;
  }
  if (h->slots[ticket].online) return wrong_ticket_kind(h, "ilcc_wait_records_device", true);
  HIP_TRY(h, hipSetDevice(h->device));
  return finish(h, ticket, out, nullptr, static_cast<float*>(d_records), n_corners, tag_base);
}

int32_t ilcc_set_result_mode(ilcc_handle* h, int32_t mode) {
  if (!h || (mode != ILCC_RESULTS_FULL && mode != ILCC_RESULTS_COMPACT)) return ILCC_BAD_ARGUMENT;
  for (const Slot& sl : h->slots)
    if (sl.busy) {
      h->err = "ilcc_set_result_mode with a batch in flight: ilcc_wait first";
      return ILCC_BAD_ARGUMENT;
    }
  h->result_mode = mode;
  return ILCC_OK;
}

uint32_t ilcc_record_floats(const ilcc_handle* h, int32_t ticket) {
  if (!h || ticket < 0 || ticket >= kSlots || !h->slots[ticket].busy) return 0u;
  return (uint32_t)ILCC_RECORD_HEADER + 3u * h->slots[ticket].rec_corners;
}

int32_t ilcc_wait_compact(ilcc_handle* h, int32_t ticket, float* records, uint64_t capacity_floats) {
  if (!h || !records || ticket < 0 || ticket >= kSlots || !h->slots[ticket].busy) {
    if (h) h->err = "ilcc_wait_compact: no batch in flight under this ticket";
    return ILCC_BAD_ARGUMENT;
  }
  if (h->slots[ticket].online) return wrong_ticket_kind(h, "ilcc_wait_compact", true);
  {
    const Slot& sl = h->slots[ticket];   // the record width was fixed when the batch was submitted
    const uint64_t need = (uint64_t)sl.n_frames * ((uint64_t)ILCC_RECORD_HEADER + 3ull * sl.rec_corners);
    if (capacity_floats < need) {
      h->err = "ilcc_wait_compact: the batch's records need " + std::to_string(need) + " floats, the buffer holds " +
               std::to_string(capacity_floats) + " (the batch stays in flight)";
      return ILCC_BAD_ARGUMENT;
    }
  }
  HIP_TRY(h, hipSetDevice(h->device));
  return finish(h, ticket, nullptr, records);
}

int32_t ilcc_fetch_results(ilcc_handle* h, uint32_t first, uint32_t n, ilcc_result* out) {
  if (!h || !out || h->last_slot < 0) return ILCC_BAD_ARGUMENT;
  Slot& sl = h->slots[h->last_slot];
  if (sl.busy || first > sl.n_frames || n > sl.n_frames - first) {
    h->err = "ilcc_fetch_results: no completed batch, or the range exceeds it";
    return ILCC_BAD_ARGUMENT;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  const int32_t st = ensure_full(h, sl);
  if (st != ILCC_OK) return st;
  const size_t row = trimmed_bytes(sl.rec_corners);
  for (uint32_t f = 0; f < n; ++f) std::memcpy(&out[f], &sl.h_res[first + f], row);
  return ILCC_OK;
}

int32_t ilcc_extract_batch_device(ilcc_handle* h, const float* d_xyzi, const uint64_t* offsets,
                                  uint32_t n_frames, const float* d_clicks, ilcc_result* out) {
  if (!h || !d_xyzi || !d_clicks || !out) return ILCC_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int32_t st = check_offsets(h, offsets, n_frames);
  if (st != ILCC_OK) return st;
  if (h->slots[0].busy) {
    h->err = "synchronous call while ticket 0 is in flight";
    return ILCC_BAD_ARGUMENT;
  }
  st = enqueue(h, 0, reinterpret_cast<const float4*>(d_xyzi), offsets, n_frames, d_clicks);
  if (st != ILCC_OK) return st;
  return finish(h, 0, out);
}

int32_t ilcc_extract_batch(ilcc_handle* h, const float* xyzi, const uint64_t* offsets, uint32_t n_frames,
                           const float* clicks, ilcc_result* out) {
  if (!h || !xyzi || !clicks || !out) return ILCC_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int32_t st = check_offsets(h, offsets, n_frames);
  if (st != ILCC_OK) return st;
  Slot& sl = h->slots[0];
  if (sl.busy) {
    h->err = "synchronous call while ticket 0 is in flight";
    return ILCC_BAD_ARGUMENT;
  }
  if (offsets[n_frames] > 0)
    HIP_TRY(h, hipMemcpyAsync(sl.d_xyzi, xyzi, sizeof(float4) * offsets[n_frames], hipMemcpyHostToDevice, sl.stream));
  HIP_TRY(h, hipMemcpyAsync(sl.d_clicks, clicks, sizeof(float) * 3 * n_frames, hipMemcpyHostToDevice, sl.stream));
  st = enqueue(h, 0, sl.d_xyzi, offsets, n_frames, sl.d_clicks);
  if (st != ILCC_OK) return st;
  return finish(h, 0, out);
}

int32_t ilcc_extract(ilcc_handle* h, const float* xyzi, uint32_t n, const float click[3], ilcc_result* out) {
  const uint64_t off[2] = {0, n};
  return ilcc_extract_batch(h, xyzi, off, 1, click, out);
}

namespace {
// what get_chessboard_by_point reports beyond the front half's record: n_roi of the WHOLE cloud, the second-tier count, and
// `if(outcloud->size() < 500 || find_board == false) return false;` (LidarCornersEst.cpp:111-112)
void finish_chessboard_by_point(ilcc_handle* h, Slot& sl, uint32_t n_frames, int32_t min_plane_points, ilcc_result* out) {
  // frames the first tier answered from a window of the cloud: n_roi is what the clustering of the WHOLE cloud runs on -- its
  // finite points (K1 counted them on the way); the handle's own copy keeps the window's count, which is what ILCC_CLOUD_ROI holds
  for (uint32_t f = 0; f < n_frames; ++f)
    if (sl.h_online[f] == 0u && out[f].status != ILCC_NO_ROI_POINTS) out[f].n_roi = (int32_t)sl.h_online[h->max_frames + f];
  for (uint32_t f = 0; f < n_frames; ++f) h->timing.online_second_tier_frames += sl.h_online[f] != 0u ? 1u : 0u;
  for (uint32_t f = 0; f < n_frames; ++f)
    if (out[f].status == ILCC_OK && (out[f].n_plane < min_plane_points || !out[f].found_board)) {
      out[f].status = ILCC_BOARD_NOT_FOUND;
      sl.h_res[f].status = ILCC_BOARD_NOT_FOUND;
    }
}
}  // namespace

int32_t ilcc_chessboard_by_point_batch(ilcc_handle* h, const float* xyzi, const uint64_t* offsets, uint32_t n_frames,
                                       const float* points, int32_t min_plane_points, ilcc_result* out) {
  if (!h || !xyzi || !points || !out) return ILCC_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int32_t st = check_offsets(h, offsets, n_frames);
  if (st != ILCC_OK) return st;
  Slot& sl = h->slots[0];
  if (sl.busy) {
    h->err = "synchronous call while ticket 0 is in flight";
    return ILCC_BAD_ARGUMENT;
  }
  if (offsets[n_frames] > 0)
    HIP_TRY(h, hipMemcpyAsync(sl.d_xyzi, xyzi, sizeof(float4) * offsets[n_frames], hipMemcpyHostToDevice, sl.stream));
  HIP_TRY(h, hipMemcpyAsync(sl.d_clicks, points, sizeof(float) * 3 * n_frames, hipMemcpyHostToDevice, sl.stream));
  st = enqueue(h, 0, sl.d_xyzi, offsets, n_frames, sl.d_clicks, /*front_only=*/true, /*no_crop=*/true);
  if (st != ILCC_OK) return st;
  st = finish(h, 0, out);
  if (st != ILCC_OK) return st;
  finish_chessboard_by_point(h, sl, n_frames, min_plane_points, out);
  return ILCC_OK;
}

// The same call in two halves, like ilcc_submit_batch / ilcc_wait: up to four calls in flight per handle, the H2D copy of one
// overlapping the kernels of the others (the two tiers need no host decision in between: the second tier's kernels skip the
// frames the first has answered).  A tracker that hands over one scan at a time gains nothing; a recorded sequence does.
int32_t ilcc_submit_chessboard_by_point(ilcc_handle* h, const float* xyzi, const uint64_t* offsets, uint32_t n_frames,
                                        const float* points, int32_t* ticket) {
  if (!h || !xyzi || !points || !ticket) return ILCC_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int32_t st = check_offsets(h, offsets, n_frames);
  if (st != ILCC_OK) return st;
  const int si = h->next_slot;
  Slot& sl = h->slots[si];
  if (sl.busy) {
    h->err = "every pipeline slot holds a batch: wait for the oldest ticket first";
    return ILCC_CAPACITY;
  }
  st = alloc_slot(h, sl);
  if (st != ILCC_OK) return st;
  if (offsets[n_frames] > 0)
    HIP_TRY(h, hipMemcpyAsync(sl.d_xyzi, xyzi, sizeof(float4) * offsets[n_frames], hipMemcpyHostToDevice, sl.stream));
  HIP_TRY(h, hipMemcpyAsync(sl.d_clicks, points, sizeof(float) * 3 * n_frames, hipMemcpyHostToDevice, sl.stream));
  st = enqueue(h, si, sl.d_xyzi, offsets, n_frames, sl.d_clicks, /*front_only=*/true, /*no_crop=*/true);
  if (st != ILCC_OK) return st;
  *ticket = si;
  h->next_slot = (si + 1) % kSlots;
  return ILCC_OK;
}

int32_t ilcc_wait_chessboard_by_point(ilcc_handle* h, int32_t ticket, int32_t min_plane_points, ilcc_result* out) {
  if (!h || !out || ticket < 0 || ticket >= kSlots || !h->slots[ticket].busy) {
    if (h) h->err = "ilcc_wait_chessboard_by_point: no batch in flight under this ticket";
    return ILCC_BAD_ARGUMENT;
  }
  if (!h->slots[ticket].online) return wrong_ticket_kind(h, "ilcc_wait_chessboard_by_point", false);
  HIP_TRY(h, hipSetDevice(h->device));
  Slot& sl = h->slots[ticket];
  const uint32_t n_frames = sl.n_frames;
  const int32_t st = finish(h, ticket, out);
  if (st != ILCC_OK) return st;
  finish_chessboard_by_point(h, sl, n_frames, min_plane_points, out);
  return ILCC_OK;
}

int64_t ilcc_fetch_classes(ilcc_handle* h, uint32_t frame, uint8_t* out_class, uint64_t cap_points) {
  if (!h || h->last_slot < 0) return -(int64_t)ILCC_BAD_ARGUMENT;
  Slot& sl = h->slots[h->last_slot];
  if (sl.busy || frame >= sl.n_frames) return -(int64_t)ILCC_BAD_ARGUMENT;
  if (ensure_full(h, sl) != ILCC_OK) return -(int64_t)ILCC_HIP_ERROR;
  const ilcc_result& r = sl.h_res[frame];
  const int64_t n = (r.status == ILCC_OK || r.status == ILCC_AMBIGUOUS || r.status == ILCC_BOARD_NOT_FOUND) ? r.n_plane : 0;
  const int64_t m = std::min<int64_t>(n, (int64_t)cap_points);
  if (m > 0 && out_class &&
      hipMemcpy(out_class, sl.d_cls + sl.off[frame], (size_t)m, hipMemcpyDeviceToHost) != hipSuccess)
    return -(int64_t)ILCC_HIP_ERROR;
  return n;
}

int64_t ilcc_fetch_cloud(ilcc_handle* h, uint32_t frame, int32_t which, float* out_xyzi, uint64_t cap_points) {
  if (!h || h->last_slot < 0) return -(int64_t)ILCC_BAD_ARGUMENT;
  Slot& sl = h->slots[h->last_slot];
  if (sl.busy || frame >= sl.n_frames) return -(int64_t)ILCC_BAD_ARGUMENT;
  if (ensure_full(h, sl) != ILCC_OK) return -(int64_t)ILCC_HIP_ERROR;
  const ilcc_result& r = sl.h_res[frame];
  const float4* src = nullptr;
  int64_t n = 0;
  switch (which) {
    case ILCC_CLOUD_ROI: src = sl.d_roi; n = r.n_roi; break;
    case ILCC_CLOUD_CLUSTER: src = sl.d_cluster; n = r.n_cluster; break;
    case ILCC_CLOUD_CHESSBOARD: src = sl.d_board; n = r.n_plane; break;   // also after ILCC_BOARD_NOT_FOUND
    case ILCC_CLOUD_PCA:
      src = sl.d_pca;
      n = (r.status == ILCC_OK || r.status == ILCC_AMBIGUOUS || r.status == ILCC_DEGENERATE_HIST || r.status == ILCC_BOARD_NOT_FOUND) ? r.n_plane : 0;
      break;
    case ILCC_CLOUD_OPTIM: src = sl.d_optim; n = (r.status == ILCC_OK || r.status == ILCC_AMBIGUOUS) ? r.n_plane : 0; break;
    default: return -(int64_t)ILCC_BAD_ARGUMENT;
  }
  const int64_t m = std::min<int64_t>(n, (int64_t)cap_points);
  if (m > 0 && out_xyzi &&
      hipMemcpy(out_xyzi, src + sl.off[frame], sizeof(float4) * (size_t)m, hipMemcpyDeviceToHost) != hipSuccess)
    return -(int64_t)ILCC_HIP_ERROR;
  return n;
}

int64_t ilcc_fetch_labelled(ilcc_handle* h, uint32_t frame, float* out_yz, uint8_t* out_label, uint64_t cap_points) {
  if (!h || h->last_slot < 0) return -(int64_t)ILCC_BAD_ARGUMENT;
  Slot& sl = h->slots[h->last_slot];
  if (sl.busy || frame >= sl.n_frames) return -(int64_t)ILCC_BAD_ARGUMENT;
  uint32_t n = 0;
  if (hipMemcpy(&n, sl.d_nlab + frame, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -(int64_t)ILCC_HIP_ERROR;
  const uint64_t m = std::min<uint64_t>(n, cap_points);
  if (m > 0) {
    if (out_yz && hipMemcpy(out_yz, sl.d_yz + sl.off[frame], sizeof(float2) * m, hipMemcpyDeviceToHost) != hipSuccess)
      return -(int64_t)ILCC_HIP_ERROR;
    if (out_label && hipMemcpy(out_label, sl.d_lab + sl.off[frame], m, hipMemcpyDeviceToHost) != hipSuccess)
      return -(int64_t)ILCC_HIP_ERROR;
  }
  return (int64_t)n;
}

int64_t ilcc_fetch_walk(ilcc_handle* h, uint32_t frame, float* out_yz, uint8_t* out_label, uint64_t cap_points, uint32_t counts[2]) {
  if (!h || h->last_slot < 0) return -(int64_t)ILCC_BAD_ARGUMENT;
  Slot& sl = h->slots[h->last_slot];
  if (sl.busy || frame >= sl.n_frames || !sl.grid) return -(int64_t)ILCC_BAD_ARGUMENT;
  uint32_t n = 0;
  if (hipMemcpy(&n, sl.d_nlab + frame, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -(int64_t)ILCC_HIP_ERROR;
  if (n > (uint32_t)kGridLdsPointsMax) n = 0;   // such a frame is walked through global memory: no layout
  if (counts) {
    if (hipMemcpy(&counts[0], sl.d_walk_mi + frame, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&counts[1], sl.d_walk_nrim + frame, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
      return -(int64_t)ILCC_HIP_ERROR;
  }
  const uint64_t m = std::min<uint64_t>(n, cap_points);
  if (m > 0) {
    if (out_yz && hipMemcpy(out_yz, sl.d_walk_yz + sl.off[frame], sizeof(float2) * m, hipMemcpyDeviceToHost) != hipSuccess)
      return -(int64_t)ILCC_HIP_ERROR;
    if (out_label && hipMemcpy(out_label, sl.d_walk_lab + sl.off[frame], m, hipMemcpyDeviceToHost) != hipSuccess)
      return -(int64_t)ILCC_HIP_ERROR;
  }
  return (int64_t)n;
}

// shared setup for the two single-kernel test entries: frame 0 of slot 0 = caller's labelled points
static int32_t stage_labelled(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m) {
  if (!h || (m > 0 && (!yz || !label))) return ILCC_BAD_ARGUMENT;
  if (m > h->max_points) {
    h->err = "more points than the handle holds";
    return ILCC_CAPACITY;
  }
  Slot& sl = h->slots[0];
  if (sl.busy) {
    h->err = "diagnostic entry while ticket 0 is in flight";
    return ILCC_BAD_ARGUMENT;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = sl.stream;
  const uint64_t off[2] = {0, m};
  ilcc_result r;
  std::memset(&r, 0, sizeof(r));
  r.status = ILCC_OK;
  HIP_TRY(h, hipMemcpyAsync(sl.d_off, off, sizeof(off), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(sl.d_res, &r, sizeof(r), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(sl.d_nlab, &m, sizeof(m), hipMemcpyHostToDevice, s));
  const uint32_t stride = walk_stride(m);
  HIP_TRY(h, hipMemcpyAsync(sl.d_walk, &stride, sizeof(stride), hipMemcpyHostToDevice, s));
  if (m > 0) {
    HIP_TRY(h, hipMemcpyAsync(sl.d_yz, yz, sizeof(float2) * m, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(sl.d_lab, label, m, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(h, hipStreamSynchronize(s));
  sl.n_frames = 0;   // the stage buffers no longer describe a batch
  return ILCC_OK;
}

int32_t ilcc_grid_cost(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m, int32_t use_oob,
                       float* cost_out, int32_t* best_index, float* best_cost) {
  int32_t st = stage_labelled(h, yz, label, m);
  if (st != ILCC_OK) return st;
  Slot& sl = h->slots[0];
  hipStream_t s = sl.stream;
  const size_t vol = (size_t)h->p.n_th * h->p.n_ty * h->p.n_tz * 2;
  float* d_vol = nullptr;
  if (cost_out) HIP_TRY(h, hipMalloc((void**)&d_vol, sizeof(float) * vol));
  struct VolGuard {   // every exit path below frees the volume
    float* p;
    ~VolGuard() { if (p) (void)hipFree(p); }
  } guard{d_vol};
  uint32_t lds = 1024;
  while (lds < m && lds < (uint32_t)kGridLdsPointsMax) lds <<= 1;
  const uint32_t saved = h->grid_lds_points;
  h->grid_lds_points = std::max(saved, lds);
  const Ctx c = make_ctx(h, sl, nullptr, nullptr, 1, 1);
  h->grid_lds_points = saved;
  const uint32_t inf_bits = 0x7f800000u;
  HIP_TRY(h, hipMemcpyAsync(sl.d_bound, &inf_bits, sizeof(inf_bits), hipMemcpyHostToDevice, s));
  launch_walk_order(c, s);
  // full evaluation when the volume is wanted, the pipeline's branch-and-bound variant otherwise
  launch_grid_cost(c, s, use_oob, d_vol, /*prune=*/d_vol == nullptr && h->p.grid_prune != 0);
  std::vector<GridPartial> part(c.grid_blocks);
  hipError_t e = hipGetLastError();   // a launch that asked for more LDS than the device grants fails HERE, not at the copy
  if (e == hipSuccess) e = hipMemcpyAsync(part.data(), sl.d_partial, sizeof(GridPartial) * c.grid_blocks, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && cost_out) e = hipMemcpyAsync(cost_out, d_vol, sizeof(float) * vol, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    h->err = std::string("ilcc_grid_cost: ") + hipGetErrorString(e);
    return ILCC_HIP_ERROR;
  }
  GridPartial b = part[0];
  for (uint32_t k = 1; k < c.grid_blocks; ++k) {
    const GridPartial& t = part[k];
    if (t.cost < b.cost || (t.cost == b.cost && (t.d2 < b.d2 || (t.d2 == b.d2 && t.flat < b.flat)))) b = t;
  }
  if (best_index) *best_index = (int32_t)b.flat;
  if (best_cost) *best_cost = b.cost;
  return ILCC_OK;
}

int32_t ilcc_grid_solve(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m, int32_t* grid_index, float* grid_cost,
                        int32_t lat[3], int32_t* phase, int64_t* cost_q, int64_t* alt_cost_q, int32_t* rounds, int32_t* hops,
                        int32_t* flags, int32_t* ties) {
  if (!h || h->p.solver != ILCC_SOLVER_GRID) {
    if (h) h->err = "ilcc_grid_solve needs ILCC_SOLVER_GRID";
    return ILCC_BAD_ARGUMENT;
  }
  int32_t st = stage_labelled(h, yz, label, m);
  if (st != ILCC_OK) return st;
  Slot& sl = h->slots[0];
  hipStream_t s = sl.stream;
  uint32_t lds = 1024;
  while (lds < m && lds < (uint32_t)kGridLdsPointsMax) lds <<= 1;
  const uint32_t saved = h->grid_lds_points;
  h->grid_lds_points = std::max(saved, lds);
  Ctx c = make_ctx(h, sl, nullptr, nullptr, 1, 1);
  h->grid_lds_points = saved;
  // what K1 resets per frame and per batch
  const uint32_t inf_bits = 0x7f800000u, zero = 0u;
  HIP_TRY(h, hipMemcpyAsync(sl.d_bound, &inf_bits, sizeof(inf_bits), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(sl.d_bound_sub, &inf_bits, sizeof(inf_bits), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(sl.d_tie_count, &zero, sizeof(zero), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemsetAsync(sl.d_iters, 0, sizeof(unsigned long long) * kBatchWords, s));
  st = enqueue_grid_search(h, sl, 0, c, s, 1, /*chain=*/false);
  if (st != ILCC_OK) return st;
  Ctx c7 = c;
  c7.tie_count = sl.d_tie_count;
  launch_pattern_refine_corners(c7, s);
  HIP_TRY(h, hipGetLastError());
  SolveRec rec{};
  ilcc_result r;
  HIP_TRY(h, hipMemcpyAsync(&rec, sl.d_solverec, sizeof(rec), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(&r, sl.d_res, offsetof(ilcc_result, corners), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  if (!rec.valid) {
    h->err = "ilcc_grid_solve: the grid search produced no candidate";
    return ILCC_TOO_FEW_POINTS;
  }
  const double div = (double)(h->p.refine_div > 0 ? h->p.refine_div : 1);
  if (grid_index) *grid_index = r.grid_index;
  if (grid_cost) *grid_cost = r.grid_cost;
  if (lat) {
    lat[0] = (int32_t)std::lround((rec.x[0] - h->p.th_min) / (h->p.th_step / div));
    lat[1] = (int32_t)std::lround((rec.x[1] - h->p.ty_min) / (h->p.ty_step / div));
    lat[2] = (int32_t)std::lround((rec.x[2] - h->p.tz_min) / (h->p.tz_step / div));
  }
  if (phase) *phase = rec.phase;
  if (cost_q) *cost_q = (int64_t)std::llround(rec.sel * 1099511627776.0);      // exact: the record holds cost_q / 2^40 in a double
  if (alt_cost_q) *alt_cost_q = (int64_t)std::llround(rec.cost_b * 1099511627776.0);
  if (rounds) *rounds = rec.iters_a;
  if (hops) *hops = rec.iters_b;
  if (flags) *flags = rec.flags;
  if (ties) *ties = rec.ties;
  return ILCC_OK;
}

int32_t ilcc_get_theta_t(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m, int32_t topleft_white,
                         int32_t use_oob, double theta_t[3], double* cost, int32_t* iterations) {
  if (!theta_t) return ILCC_BAD_ARGUMENT;
  int32_t st = stage_labelled(h, yz, label, m);
  if (st != ILCC_OK) return st;
  Slot& sl = h->slots[0];
  hipStream_t s = sl.stream;
  HIP_TRY(h, hipMemcpyAsync(h->d_solve, theta_t, sizeof(double) * 3, hipMemcpyHostToDevice, s));
  const Ctx c = make_ctx(h, sl, nullptr, nullptr, 1, 1);
  launch_local_solve(c, s, topleft_white, use_oob, h->d_solve, h->d_solve + 3);
  HIP_TRY(h, hipGetLastError());
  double back[5];
  HIP_TRY(h, hipMemcpyAsync(back, h->d_solve, sizeof(back), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  theta_t[0] = back[0];
  theta_t[1] = back[1];
  theta_t[2] = back[2];
  if (cost) *cost = back[3];
  if (iterations) *iterations = (int32_t)back[4];
  return ILCC_OK;
}

int32_t ilcc_pattern_refine(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m, int32_t lat[3],
                            int32_t* phase, int64_t* cost_q, int64_t* alt_cost_q, int32_t* rounds, int32_t* hops) {
  if (!h || !lat || !phase) return ILCC_BAD_ARGUMENT;
  if (lat[0] < h->th_lat_lo || lat[0] > h->th_lat_hi || (*phase != 0 && *phase != 1)) {
    // the kernel indexes its cos/sin table with lat[0] (the basin check reads it unguarded)
    h->err = "ilcc_pattern_refine: theta lattice coordinate outside [th_min - refine_th_margin, th_max + refine_th_margin] steps";
    return ILCC_BAD_ARGUMENT;
  }
  int32_t st = stage_labelled(h, yz, label, m);
  if (st != ILCC_OK) return st;
  Slot& sl = h->slots[0];
  hipStream_t s = sl.stream;
  RefineOut io{};
  io.lat[0] = lat[0];
  io.lat[1] = lat[1];
  io.lat[2] = lat[2];
  io.phase = *phase;
  HIP_TRY(h, hipMemcpyAsync(h->d_refine_io, &io, sizeof(io), hipMemcpyHostToDevice, s));
  const Ctx c = make_ctx(h, sl, nullptr, nullptr, 1, 1);
  launch_pattern_refine_test(c, s, h->d_refine_io);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(&io, h->d_refine_io, sizeof(io), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  lat[0] = io.lat[0];
  lat[1] = io.lat[1];
  lat[2] = io.lat[2];
  *phase = io.phase;
  if (cost_q) *cost_q = io.cost_q;
  if (alt_cost_q) *alt_cost_q = io.alt_q;
  if (rounds) *rounds = io.rounds;
  if (hops) *hops = io.hops;
  return ILCC_OK;
}

int32_t ilcc_debug_timeline_enable(ilcc_handle* h, int32_t on) {
  if (!h) return ILCC_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  h->tl_rows.clear();
  h->tl_on = on != 0;
  if (h->tl_on) {
    if (!h->tl_ref) HIP_TRY(h, hipEventCreate(&h->tl_ref));
    HIP_TRY(h, hipEventRecord(h->tl_ref, h->slots[0].stream));
    HIP_TRY(h, hipEventSynchronize(h->tl_ref));
  }
  return ILCC_OK;
}

int32_t ilcc_debug_timeline_fetch(ilcc_handle* h, double* rows, uint32_t cap_rows) {
  if (!h || (!rows && cap_rows)) return -ILCC_BAD_ARGUMENT;
  const size_t have = h->tl_rows.size() / ILCC_TIMELINE_COLS;
  const size_t n = std::min<size_t>(have, cap_rows);
  if (n) std::memcpy(rows, h->tl_rows.data(), sizeof(double) * n * ILCC_TIMELINE_COLS);
  h->tl_rows.clear();
  return (int32_t)n;
}

void ilcc_get_timing(const ilcc_handle* h, ilcc_timing* t) {
  if (h && t) *t = h->timing;
}
void ilcc_reset_timing(ilcc_handle* h) {
  if (h) h->timing = ilcc_timing{};
}

// get_lidar_corners.cpp:27-36 -- ofstream(trunc), `x << " " << y << " " << z << endl` with the
// stream's default float formatting (precision 6).
int32_t ilcc_save_corners2txt(const float* corners_xyz, uint32_t n_corners, const char* filename) {
  if (!corners_xyz || !filename) return ILCC_BAD_ARGUMENT;
  std::ofstream outfile(filename, std::ios_base::trunc);
  if (!outfile.is_open()) return ILCC_IO_ERROR;
  for (uint32_t i = 0; i < n_corners; ++i)
    outfile << corners_xyz[3 * i] << " " << corners_xyz[3 * i + 1] << " " << corners_xyz[3 * i + 2] << std::endl;
  outfile.close();
  return outfile.fail() ? ILCC_IO_ERROR : ILCC_OK;
}

// ImageCornersEst.cpp:281-299 -- `float x,y,z; infile >> x >> y >> z` until eof or num corners
int32_t ilcc_read_lidar_corners(const char* filename, uint32_t num, double* out_xyz) {
  if (!filename || !out_xyz) return -ILCC_BAD_ARGUMENT;
  std::ifstream infile(filename);
  if (!infile.is_open()) return -ILCC_IO_ERROR;
  uint32_t counter = 0;
  while (!infile.eof() && counter < num) {
    float x = 0, y = 0, z = 0;
    infile >> x >> y >> z;
    if (infile.fail()) break;
    out_xyz[3 * counter] = x;
    out_xyz[3 * counter + 1] = y;
    out_xyz[3 * counter + 2] = z;
    ++counter;
  }
  return (int32_t)counter;
}

}  // extern "C"

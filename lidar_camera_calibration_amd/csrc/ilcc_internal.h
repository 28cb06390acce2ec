// ilcc_internal.h -- device-side context, block/wave utilities shared by the stage kernels.
// gfx950 only: wavefront = 64 lanes, 4 SIMD-32 per CU, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ilcc_hip.h"

#define ILCC_WAVE 64

namespace ilcc {

// block sizes (multiples of the 64-lane wavefront)
constexpr int kCropThreads = 256;      // K1
constexpr int kCropChunk = 4096;       // points per K1 block
#ifndef ILCC_K2_THREADS
#define ILCC_K2_THREADS 256    // round 4: the clustering runs on cells (k2_cluster.hip); rounds 1-3: 1024 threads and 100 KB of LDS per frame
#endif
constexpr int kFrameThreads = ILCC_K2_THREADS;    // K2: one workgroup per frame (1024 threads in batches of <= kSmallBatchFrames frames)
#ifndef ILCC_K3_THREADS
#define ILCC_K3_THREADS 256    // measured in the pipelined bench: 1024: 229.6 k, 512: 230.4 k, 256: 235.2 k frames/s
#endif
#ifndef ILCC_K45_THREADS
#define ILCC_K45_THREADS 256   // measured in the pipelined bench (box pre-pass in place): 1024: 519 k, 512: 530 k, 256: 543 k frames/s -- a 1024-thread workgroup needs 16 free wave slots on ONE CU while another batch's K6 full pass keeps refilling them
#endif
constexpr int kPlaneThreads = ILCC_K3_THREADS;    // K3: one workgroup per frame
constexpr int kPlaneThreadsSmallBatch = 1024;     // K3 in batches of <= kSmallBatchFrames frames (latency, not CU footprint, matters)
#ifndef ILCC_SMALL_BATCH
#define ILCC_SMALL_BATCH 64
#endif
constexpr int kSmallBatchFrames = ILCC_SMALL_BATCH;   // batches this small cannot fill 256 CUs with one workgroup per frame: the per-frame kernels go wide
constexpr int kHistThreads = ILCC_K45_THREADS;    // K4/K5: one workgroup per frame
#ifndef ILCC_K6_THREADS
#define ILCC_K6_THREADS 256
#endif
constexpr int kGridThreads = ILCC_K6_THREADS;      // K6: wavefronts x 64 per workgroup (measured: see DESIGN.md)
#ifndef ILCC_K6_THREADS_LARGE
#define ILCC_K6_THREADS_LARGE 512
#endif
constexpr int kGridThreadsLarge = ILCC_K6_THREADS_LARGE;   // K6 on frames staged above kGridLargeFrom points per workgroup
#ifndef ILCC_K6_LARGE_FROM
#define ILCC_K6_LARGE_FROM 2048
#endif
constexpr int kGridLargeFrom = ILCC_K6_LARGE_FROM;
constexpr int kTileA = 4;              // K6 candidate tile: ty values per wavefront pass
#ifndef ILCC_TILE_B
#define ILCC_TILE_B 4
#endif
constexpr int kTileB = ILCC_TILE_B;    // K6 candidate tile: tz values per wavefront pass (4 or 8)
#ifndef ILCC_K6_GROUP
#define ILCC_K6_GROUP 7   // round 4 (config 2 / config 5, k frames/s): 2: 821 / 50.9, 3: 837 / 54.5, 4: 837 / 55.1, 5: 842 / 56.2, 7: 841 / 56.3; round 6, the per-theta pre-pass skipped behind a valid mask: 3: 1 235 / 82.9, 4: 1 250 / 83.9, 5: 1 276 / 89.7, 7: 1 280 / 91.5, 9: 1 283 / 92.6, 11: 1 271 / 92.2, 15: 1 177 / 90.8
#endif
constexpr int kThetaGroup = ILCC_K6_GROUP;   // K6: consecutive thetas that share one common box pre-pass (k6_group_prepass)
constexpr int kGridLdsPointsMax = 8192;   // K6 LDS staging upper bound (12 B per point -> 96 KiB)
constexpr int kGridTableMax = 8192;       // K6: n_ty + n_tz bound (their tables sit in LDS behind the points: 32 KiB)
#ifndef ILCC_K7_THREADS
#define ILCC_K7_THREADS 256
#endif
constexpr int kSolveThreads = ILCC_K7_THREADS;     // K7: wavefronts x 64 per (frame, phase)
#ifndef ILCC_K7R_THREADS
#define ILCC_K7R_THREADS 192   // 3 wavefronts, one per theta of the stencil: measured 192: 230 k, 384-768: 225 k, 1024: 219 k frames/s
                               // (the kernel's latency is set by the few frames that walk 30+ rounds; small workgroups leave the CUs to K6)
#endif
constexpr int kRefineThreads = ILCC_K7R_THREADS;   // K7r: one workgroup per frame
constexpr int kRefineThreadsSmallBatch = 768;      // K7r in batches of <= kSmallBatchFrames frames: 4 wavefronts per theta of the stencil
constexpr int kRefineList = 32;        // K7r: candidates evaluated per sweep over the points
constexpr int kTieCap = 256;           // K6 -> K7a: near-tie candidates kept per frame for the fp64 recount
constexpr float kTieEps = 2e-5f;       // relative cost window of a near-tie (fp32 sums of ~1e3 terms agree to ~1e-6)
constexpr int kCoverageCellsMax = 1024;   // K7b: board squares tracked by the coverage mask (board_w x board_h)
constexpr int kIterSlots = 64;         // K6 executed-iteration counters (spread to avoid one hot atomic); [0,64): all points, [64,128): interior-class points, [128,192): (point, tile) evaluations of the box pre-pass
constexpr int kBatchWords = 3 * kIterSlots + 2;   // Ctx::grid_iters: the K6 counters, then K2's: [192] most occupied cells a frame's LDS cell grid needed ([193] spare)
#ifndef ILCC_K2_ALLPAIRS_MAX
#define ILCC_K2_ALLPAIRS_MAX 256
#endif
constexpr int kClusterAllPairsMax = ILCC_K2_ALLPAIRS_MAX;   // K2: above this many points the spatial hash finds neighbours
constexpr int kClusterHashSize = 1 << 17;    // K2: hash buckets per frame (global memory)
constexpr int kClusterLdsPointsMax = 4096;   // K2: largest LDS capacity for the cell-sorted points (ROI points per frame); larger frames keep them in HBM/L2
constexpr int kClusterLdsPointsMin = 1024;   // K2: smallest (the handle grows it in steps of 512 with the ROI sizes it sees)
constexpr int kClusterCellsMin = 512;        // K2: occupied cells per frame the LDS arrays hold: smallest ...
constexpr int kClusterCellsMax = 6144;       // ... and largest capacity (5 words per cell: 120 KiB) -- the handle grows it in steps of 256

struct GridPartial {   // per K6 workgroup best candidate
  float cost;
  uint32_t d2;         // squared index distance to the candidate nearest (0,0,0)
  uint32_t flat;       // ((k*n_ty+a)*n_tz+b)*2+phase
  uint32_t pad;
};

struct SolveRec {      // K7a / K7r result for one (frame, phase slot)
  double x[3];
  double cost_a, cost_b, sel;
  double margin;       // K7r: (cheapest neighbouring basin - cost) / cost
  int32_t iters_a, iters_b, phase, valid;
  int32_t flags, ties;
};

struct RefineOut {     // K7r diagnostic entry (ilcc_pattern_refine)
  long long cost_q, alt_q;
  int32_t lat[3];
  int32_t phase, rounds, hops;
};

// everything a kernel needs, passed by value
struct Ctx {
  // inputs
  const float4* xyzi;        // all frames, packed
  const uint64_t* off;       // n_frames+1 point offsets (device copy)
  const float* clicks;       // n_frames x 3
  uint32_t n_frames;
  uint32_t crop_chunks;      // K1 chunks per frame (max over frames)
  // per-frame records
  ilcc_result* res;
  // stage buffers, all indexed with the input offsets (capacity of a frame = its input size)
  float4* roi;
  float4* cluster;
  float4* board;             // m_cloud_chessboard
  float4* pca;               // m_cloud_PCA
  float4* optim;             // m_cloud_optim
  float2* yz;                // labelled (non-gray) points, plane-frame y,z
  uint8_t* lab;              // 0 black, 1 white
  uint32_t* n_lab;           // per frame
  uint32_t* walk_stride;     // per frame: K6's point-walk stride, walk_stride(n_lab) -- written with n_lab (K4/K5, stage_labelled)
  // K5w: the labelled points in K6's walk layout [interior | rim | other border] (frames of at most kGridLdsPointsMax points)
  float2* walk_yz;
  uint8_t* walk_lab;
  uint32_t *walk_mi, *walk_nrim;   // per frame: interior-class points, rim points
  uint8_t* cls;              // color_by_gray_zone class of every plane point: 0 black, 1 gray, 2 white
  uint32_t* crop_counts;     // n_frames x crop_chunks
  unsigned long long* crop_masks;  // n_frames x crop_chunks x (kCropChunk / 64) keep-bits of the count pass
  uint32_t* uf_parent;       // K2 scratch (global fallback / labels)
  uint32_t* uf_count;        // K2 component sizes
  uint32_t* uf_hash_head;    // K2 spatial hash: n_frames x kClusterHashSize bucket heads
  uint32_t* uf_hash_next;    // K2 spatial hash: chain links, one per point
  uint32_t cluster_lds_points;   // K2: ROI points per frame whose cell-sorted copy fits the workgroup's LDS; larger frames sort into HBM
  uint32_t cluster_cells_cap;    // K2: occupied cells per frame the workgroup's LDS arrays hold; frames with more take the point-level path
  uint32_t wide;             // per-frame kernels at their small-batch (latency) widths whatever the batch size: synchronous calls nothing overlaps with (the online caller)
  uint32_t cluster_bits;     // K2: cells of the padded bounding grid the LDS bitmap holds (a multiple of 64)
  // The online caller (get_chessboard_by_point on un-cropped clouds) in two tiers.  online_tier 1: K1 crops a window of
  // +-online_window around the predicted point, K2 clusters it on the LDS cell grid and VERIFIES that the result is the one the
  // whole cloud gives (see fine_cluster_frame); frames it cannot vouch for get frame_flags[f] = 1.  online_tier 2: K1 (unbounded
  // box) and K2 (hashed cells) run on the flagged frames only.  0: the ROI pipeline.
  uint32_t online_tier;
  float online_window;
  uint32_t* frame_flags;     // n_frames
  uint32_t* n_finite;        // n_frames: finite points of the frame (tier 1; what an unbounded crop would keep)
  uint32_t* crop_fin;        // n_frames x crop_chunks: finite points per K1 chunk (tier 1)
  uint32_t* list_count;      // tier 2: number of listed (flagged) frames ...
  uint32_t* list;            // ... and their indices
  void* list_frames;         // tier 2: one 64-byte ListedFrame (k2_cluster.hip) per frame
  uint32_t list_grid;        // tier 2: workgroups of the kernels over (listed frame, chunk) items
  GridPartial* partial;      // n_frames x grid_blocks
  SolveRec* solve_rec;       // n_frames x 2
  uint32_t grid_blocks;      // K6 workgroups per frame
  uint32_t grid_lds_points;  // K6 points staged in LDS per workgroup (multiple of 64)
  uint32_t* grid_bound;      // per frame: float bits of the best complete candidate cost so far (K6 pruning)
  uint32_t* grid_bound_sub;  // the same for the subsampled seed / refinement launches (costs over a prefix of the walk: never a valid bound for complete costs)
  uint32_t walk_limit;       // K6: 0 = every labelled point; else only the first walk_limit positions of the walk (a uniform sample of the board)
  // near ties: candidates of the full pass whose fp32 cost is within kTieEps of the bound at the time they
  // complete; K7r recounts them on the oracle's fixed-point cost so that the argmin is the oracle's even when fp32 cannot order them
  uint32_t* tie_count;       // per frame (nullptr: this launch does not collect)
  uint32_t* tie_count_all;   // the same array, always set: K1 resets it
  GridPartial* tie_list;     // n_frames x kTieCap: cost (fp32), d2, flat
  unsigned long long* grid_iters;  // executed K6 work in counts of grid_cost_evals_per_count() evaluations, for the VALU rate
  uint32_t box_points;             // K6 full pass: border-class walk positions the box pre-pass looks at per tile (0: no pre-pass)
  // K6 full pass behind k6_group_prepass (nullptr: no common pre-pass): per (frame, group of kThetaGroup thetas) a state word and a bit mask
  const uint32_t* grp_alive;
  const uint32_t* grp_mask;
  uint32_t grp_count, grp_words;   // theta groups per frame = ceil(n_th / kThetaGroup); mask words per group = ceil(tiles / 32)
  // seeding pass of the branch-and-bound (a decimated subset of the same grid, evaluated first)
  const GridPartial* seed_partial; // n_frames x seed_blocks, nullptr when this launch is the seed pass / unused
  uint32_t seed_blocks;
  int32_t seed_n_ty, seed_n_tz, seed_stride_t;   // seed (a2,b2) -> grid (a2*stride, b2*stride)
  int32_t seed_stride_th, seed_off_th;           // seed k2 -> grid theta index seed_off_th + k2*seed_stride_th
  // refinement pass (between seed and full pass): workgroup j evaluates the 16 x 16 (ty, tz) window around
  // the seed argmin at theta index (seed theta) + j - refine_radius_th; 0 = this launch is not a refinement
  int32_t refine_window;     // != 0: this launch (the refinement) evaluates only a window of refine_window x refine_window tiles (2: 8 x 8 translations; 4: 16 x 16) around the seed argmin, every refine_step_th-th theta within +-refine_radius_th
  int32_t refine_radius_th;
  int32_t refine_step_th;    // theta step between the workgroups of a refinement / anchor launch (kRefineThetaStride / 1)
  // candidate tables (device)
  const float* cth;          // cos(theta_k)/g
  const float* sth;          // sin(theta_k)/g
  const float* ay;           // (ty_a + W g/2)/g
  const float* az;           // (tz_b + H g/2)/g
  // K7r: cos/sin of every theta lattice point, index (lattice theta) - th_lat_lo
  const double2* th_lattice;
  int32_t th_lat_lo, th_lat_hi;
  int32_t refine_hop_y, refine_hop_z;   // one board square along y / z in lattice units
  // parameters
  ilcc_params p;
  int32_t c_th, c_ty, c_tz;  // index of the candidate nearest zero on each axis
};

// K6 walks a frame's M labelled points in the order slot s <- point (s * S) mod M with S ~ 0.618 M coprime to M (a
// golden-ratio permutation: every prefix of the walk is a sample spread over the whole board).  Computed ONCE per frame
// where n_lab is written -- the Euclid loop costs a few hundred instructions and every one of K6's ~150 workgroups per
// frame used to repeat it.
__host__ __device__ inline uint32_t walk_stride(uint32_t M) {
  if (M <= 2) return 1u;
  uint32_t S = ((uint32_t)((float)M * 0.6180339f)) | 1u;
  for (;; S += 2u) {
    uint32_t a = S, b = M;
    while (b) {
      const uint32_t t = a % b;
      a = b;
      b = t;
    }
    if (a == 1u) break;
  }
  return S >= M ? 1u : S;
}

// ---------------------------------------------------------------- wave / block helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & (ILCC_WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, ILCC_WAVE);
  return v;  // valid in lane 0
}

// sum over the whole workgroup; result broadcast to every thread. scratch: >= 17 T in LDS.
// Fixed combination order (lane tree, then wavefronts in index order) -> deterministic.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  v = wave_sum(v);
  const int nw = (blockDim.x + ILCC_WAVE - 1) / ILCC_WAVE;
  __syncthreads();
  if (lane_id() == 0) scratch[wave_id()] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    T s = scratch[0];
    for (int w = 1; w < nw; ++w) s += scratch[w];
    scratch[16] = s;
  }
  __syncthreads();
  return scratch[16];
}

// N block sums with ONE pair of barriers; the same combination order as block_sum (lane tree, then wavefronts in
// index order), so every total is bit-identical to N separate block_sum calls
template <int NV>
__device__ __forceinline__ void block_sum_n(double (&v)[NV], double* scratch /* >= 16 * NV + NV */) {
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  const int kW = (blockDim.x + ILCC_WAVE - 1) / ILCC_WAVE;
  __syncthreads();
  if (lane_id() == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) scratch[wave_id() * NV + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t = scratch[k];
    for (int w = 1; w < kW; ++w) t += scratch[w * NV + k];
    v[k] = t;
  }
}

// exclusive scan of 0/1 flags over the workgroup (thread order); returns rank, total via ref.
// scratch: >= 17 uint32 in LDS.
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* scratch, uint32_t& total) {
  const unsigned long long m = __ballot(flag);
  const int lane = lane_id();
  const uint32_t in_wave = __popcll(m & ((1ull << lane) - 1ull));
  const int nw = (blockDim.x + ILCC_WAVE - 1) / ILCC_WAVE;
  __syncthreads();
  if (lane == 0) scratch[wave_id()] = __popcll(m);
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const uint32_t c = scratch[w];
    if (w < wave_id()) base += c;
    tot += c;
  }
  total = tot;
  return base + in_wave;
}

// k6_locate (k6_grid_cost.hip): seed + refinement + anchor of the grid search in one launch, one workgroup per frame
struct LocatePlan {
  const float *cth2, *sth2, *ay2, *az2;   // the seed's decimated tables (subsets of the full ones)
  int32_t n_th2, n_ty2, n_tz2;
  int32_t c_th2, c_ty2, c_tz2;            // their entries nearest zero (tie-break distance)
  int32_t stride_th, off_th, stride_t;    // seed (k2, a2, b2) -> full tables (off_th + k2 stride_th, a2 stride_t, b2 stride_t)
  int32_t refine_radius;                  // refinement: theta within +- this many steps of the seed's
  uint32_t sample_min, sample_cap;        // the sample: max(sample_min, M >> kSeedShift) walk positions, at most sample_cap (LDS)
  GridPartial* out;                       // one record per frame: the anchor's best candidate (full-table flat index, (a << 16) | b)
};

size_t locate_lds_bytes(uint32_t sample_cap, int n_ty, int n_tz, int n_ty2, int n_tz2);
void launch_locate(const Ctx& c, hipStream_t s, const LocatePlan& lp);
void launch_anchor(const Ctx& c, hipStream_t s);   // one anchor round of the separate locate launches (c.seed_partial -> c.partial, c.grid_blocks thetas)
constexpr int kRefineThetaStride = 2;  // the refinement scores every other theta of its range (the anchor covers the ones in between)
constexpr int kLocateMinFrames = 512;   // smaller batches keep the three launches: a frame's workgroups per theta are what fills the chip there (128 frames alone: 0.27 ms in three launches, 0.30 ms in one)

// ---------------------------------------------------------------- launchers (one per stage TU)
void launch_roi_crop(const Ctx& c, hipStream_t s, hipEvent_t after_count = nullptr);   // after_count: recorded between the count pass and the scatter
void launch_cluster(const Ctx& c, hipStream_t s);
void launch_ransac_plane(const Ctx& c, hipStream_t s);
void launch_plane_frame_hist(const Ctx& c, hipStream_t s);
void launch_walk_order(const Ctx& c, hipStream_t s);   // K5w (k6_grid_cost.hip): before any launch_grid_cost on the frames
void launch_grid_cost(const Ctx& c, hipStream_t s, int32_t use_oob, float* cost_volume /*nullable*/,
                      bool prune);
void launch_group_prepass(const Ctx& c, hipStream_t s, uint32_t* grp_alive, uint32_t* grp_mask);   // in front of the full pass (c.grp_count, c.grp_words set)
uint32_t grid_cost_evals_per_count();   // (point, candidate) evaluations behind one count of Ctx::grid_iters
void launch_refine_corners(const Ctx& c, hipStream_t s);
void launch_pack_records(const ilcc_result* d_res, uint32_t n_frames, uint32_t n_corners, uint32_t tag_base, float* d_out,
                         hipStream_t s);
// device memory -> pinned (mapped) host memory with the GPU's own stores, on stream s: no SDMA command (k7_refine_corners.hip)
void launch_store_to_host(const void* d_src, void* h_dst, size_t bytes, hipStream_t s);
// K7r on every frame of the batch (GRID mode), then K7b
void launch_pattern_refine_corners(const Ctx& c, hipStream_t s);
// stand-alone K7r on the labelled points of frame 0 (test entry)
void launch_pattern_refine_test(const Ctx& c, hipStream_t s, RefineOut* d_io);
// once per (process, device): raise the dynamic-LDS limits of the kernels that need more than 64 KiB
hipError_t set_kernel_attributes_k2();
size_t cluster_lds_bytes(uint32_t pts_cap, uint32_t cells_cap, uint32_t bits);   // K2's dynamic LDS for these capacities (bits: cells the bitmap holds)
uint32_t cluster_bits_default();
uint32_t cluster_bits_online();
hipError_t set_kernel_attributes_k6();
hipError_t set_kernel_attributes_k7();
// stand-alone local solve on the labelled points of frame 0 (test entry)
void launch_local_solve(const Ctx& c, hipStream_t s, int32_t tlw, int32_t use_oob, double* theta_t,
                        double* cost_iters /*[2]: cost, iterations*/);

}  // namespace ilcc

/*
 * ilcc_hip.h -- C-ABI of libilcc_hip.so: MI355X (gfx950) implementation of ilcc2's LiDAR
 * chessboard-corner extraction path.
 *
 * The reference has no FFI; its boundary for this path is the class LidarCornersEst as driven
 * by one ROS node (all citations relative to /root/reference/):
 *
 *   reference call (ilcc2/test/get_lidar_corners.cpp)                     C-ABI replacement
 *   ------------------------------------------------------------------   ----------------------------
 *   :112 new LidarCornersEst                                              ilcc_create
 *   :114 set_chessboard_param(yaml)   (src/LidarCornersEst.cpp:20-46)     ilcc_set_chessboard_param
 *   :183 setROI(cloud, click)         (src/LidarCornersEst.cpp:48-70)     \
 *   :188 EuclideanCluster()           (:124-186, getPlane :190-221)        |
 *   :191 PCA()                        (:366-372, :330-364, :224-328)       > ilcc_extract / _batch /
 *   :194 get_corners(lidar_corner)    (:374-450, Optimization.cpp:94-160,  |   _batch_device
 *                                      Optimization.h:31-107, :501-556)   /
 *   :192-193,200-201 m_cloud_chessboard/_PCA/_optim/_corners members       ilcc_fetch_cloud
 *   :197-198 save_corners2txt(m_cloud_corners, path) (:27-36)             ilcc_save_corners2txt
 *   consumer: ImageCornersEst::read_lidar_corners (src/ImageCornersEst.cpp:281-299)
 *                                                                         ilcc_read_lidar_corners
 *
 * Rules of the boundary: plain pointers and sizes only; the caller owns every input/output
 * buffer; the library owns device memory inside the handle; no exception crosses the ABI; a
 * batch call never aborts on a bad frame (per-frame status instead of the reference's bool
 * returns / uncaught std::out_of_range).  A handle is not thread-safe: one handle per
 * (thread, device); every call is synchronous on return unless stated otherwise.
 * There is NO CPU fallback: without a HIP device ilcc_create fails.
 */
#ifndef ILCC_HIP_H_
#define ILCC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILCC_MAX_CORNERS 256
#define ILCC_ABI_VERSION 5

/* per-frame / per-call status */
enum {
  ILCC_OK = 0,
  ILCC_NO_ROI_POINTS = 1,   /* setROI left nothing */
  ILCC_NO_CLUSTER = 2,      /* reference: cluster_indices.at(0) throws (LidarCornersEst.cpp:167) */
  ILCC_NO_PLANE = 3,        /* reference: PCL_ERROR (LidarCornersEst.cpp:206-209) */
  ILCC_DEGENERATE_HIST = 4, /* reference: UB in calHist (LidarCornersEst.cpp:235,266-282) */
  ILCC_TOO_FEW_POINTS = 5,
  ILCC_BAD_ARGUMENT = 6,
  ILCC_CAPACITY = 7,        /* more frames / points than the handle was created for */
  ILCC_HIP_ERROR = 8,
  ILCC_IO_ERROR = 9,
  ILCC_BOARD_NOT_FOUND = 10, /* get_chessboard_by_point returned false (plane < 500 points or no cluster
                                around the predicted centre), LidarCornersEst.cpp:111-112 */
  ILCC_AMBIGUOUS = 11        /* ILCC_SOLVER_GRID: a basin one square away costs (almost) the same as the chosen one
                                (basin_margin < ambiguity_eps); corners ARE written.  The reference leaves this call to
                                the human at the viewer (keys 'd' / 'r', LidarCornersEst.cpp:415-437) */
};

/* ilcc_result.flags */
enum {
  ILCC_FLAG_TIE_OVERFLOW = 1, /* more than 256 grid candidates within 2e-5 of the minimum: the fixed-point recount of
                                 near ties was skipped and the fp32 argmin (same tie-break) was used */
  ILCC_FLAG_REFINE_CAPPED = 2, /* a pattern search stopped at refine_max_rounds before its stride reached the finest lattice:
                                 theta_t is a valid (cheaper-than-start) point but not a lattice minimum */
  ILCC_FLAG_BORDER_RISK = 8,  /* ABI 5: a labelled point lies within fp32 rounding (4e-6 square) of a cell border under the grid argmin or one
                                 of its 26 grid neighbours.  The grid pass ranks candidates on fp32 sums; such a point may fall into the
                                 other cell there, its term differ by a whole residual from the exact (fp64) one, and the ranking of
                                 those candidates from the exact one.  The refinement then does NOT take its shortcut (the first round is
                                 evaluated on exact costs, not inferred from the grid pass), so everything downstream of the grid argmin
                                 is exact; the argmin itself stays the fp32 one.  Measure-zero on real data (7-14 % of synthetic frames
                                 have a point inside the window somewhere in the 27-neighbourhood; none of 16 384 swept frames differed) */
  ILCC_FLAG_LOW_COVERAGE = 4  /* fewer than min_cell_coverage of the board's squares hold a labelled point under the final
                                 pose (cells_hit < min_cell_coverage * board_w * board_h): the pattern is under-sampled
                                 (far board, few rings) and a one-square slip can fit the data BETTER than the truth.
                                 status stays ILCC_OK; see "accepting a frame" below */
};

/* how (theta, ty, tz) is found */
enum {
  /* the reference's own trajectory: Ceres-style local solve from (0,0,0), pass A (out-of-board
   * term on) then pass B (off) -- LidarCornersEst.cpp:398-409 */
  ILCC_SOLVER_REFERENCE_LOCAL = 0,
  /* exhaustive (theta,ty,tz) x colour-phase grid on the pass-A cost (out-of-board term on), one
   * wavefront per candidate tile; then a monotone pattern search on the same cost (27-point stencil on a lattice
   * of step / refine_div, fixed-point sums) and a check of the eight neighbouring one-square-shifted basins.
   * No Ceres-style iteration in this mode: the reference's two solves, restated faithfully, stop at their
   * 50-iteration caps and can leave the cost higher than where they started. */
  ILCC_SOLVER_GRID = 1
};

/* which clouds ilcc_fetch_cloud can return (topics of get_lidar_corners.cpp:120-124) */
enum {
  ILCC_CLOUD_ROI = 0,        /* m_cloud_ROI */
  ILCC_CLOUD_CLUSTER = 1,    /* chosen Euclidean cluster (temp_cloud before getPlane) */
  ILCC_CLOUD_CHESSBOARD = 2, /* m_cloud_chessboard  -> /ChessBoard */
  ILCC_CLOUD_PCA = 3,        /* m_cloud_PCA         -> /pca_cloud */
  ILCC_CLOUD_OPTIM = 4       /* m_cloud_optim       -> /Optim_cloud */
};

typedef struct ilcc_params {
  /* setROI half extents x,y,z (LidarCornersEst.cpp:59,64,54) */
  double roi_half[3];
  /* EuclideanCluster (LidarCornersEst.cpp:131-133) */
  double cluster_tol;
  int32_t cluster_min;
  int32_t cluster_max;
  /* getPlane (LidarCornersEst.cpp:201); ransac_hyp = SACSegmentation::max_iterations_ (PCL default 50) of the counter-based
   * sampler's hypotheses (with ransac_probability <= 0: their fixed number, as in ABI <= 4) */
  double ransac_thresh;
  int32_t ransac_hyp;
  uint32_t ransac_seed;
  /* calHist / get_gray_zone (LidarCornersEst.cpp:226,371) */
  int32_t hist_bins;
  double gray_rate;
  /* get_theta_t (Optimization.cpp:137) */
  double huber_delta;
  /* board: squares per side, board_w <= board_h (LidarCornersEst.cpp:30-39), side length */
  double grid_length;
  int32_t board_w;
  int32_t board_h;
  /* solver */
  int32_t solver;      /* ILCC_SOLVER_* */
  int32_t phase_mode;  /* REFERENCE_LOCAL only: 0 topleftWhite=false (reference's first turn),
                          1 true, 2 both from zero, keep lower with-OOB cost (replaces key 'd') */
  int32_t max_iterations; /* trust-region iterations per pass (Ceres default 50) */
  int32_t grid_prune;     /* 1 (default): K6 abandons a candidate tile as soon as its partial costs exceed
                             the best complete cost known for the frame (exact: the argmin cannot
                             change); 0: every candidate is summed over every point */
  /* exhaustive grid: theta_k = th_min + k th_step (rad), ty_a, tz_b likewise (m) */
  int32_t n_th, n_ty, n_tz;
  double th_min, th_step;
  double ty_min, ty_step;
  double tz_min, tz_step;
  /* ILCC_SOLVER_GRID refinement */
  int32_t refine_div;        /* finest lattice = grid step / refine_div; power of two <= 64 (default 16), 0: keep the grid argmin */
  int32_t refine_max_rounds; /* bound on the 27-candidate rounds of one pattern search (default 64) */
  int32_t refine_th_margin;  /* the search may leave the grid's theta range by this many grid steps (default 32) */
  int32_t reserved0;
  double ambiguity_eps;      /* status ILCC_AMBIGUOUS when basin_margin < ambiguity_eps (default 1.0: the best alternative must cost at least twice as much; <= 0: never) */
  /* get_chessboard_by_point hard-codes its own tolerance: setClusterTolerance(0.1), LidarCornersEst.cpp:80 */
  double online_cluster_tol;
  double min_cell_coverage;  /* ILCC_FLAG_LOW_COVERAGE below this fraction of occupied board squares (default 0.9; <= 0: never) */
  double ransac_probability; /* ABI 5: SACSegmentation::probability_ (PCL default 0.99): the plane RANSAC stops once its iterations reach
                                log(1 - p) / log(1 - w^3), w = the best hypothesis' inlier share -- pcl::RandomSampleConsensus's rule,
                                3-5 hypotheses on a board cluster; <= 0: ransac_hyp hypotheses, no early stop */
} ilcc_params;

/* Accepting a frame.  The reference leaves the decision to the operator, who looks at the virtual board drawn over the
 * points and presses 'o' or 'r' (LidarCornersEst.cpp:415-441).  The automatic stand-in has two signals:
 *   status == ILCC_AMBIGUOUS   a basin one square away costs about the same (basin_margin < ambiguity_eps)
 *   flags & ILCC_FLAG_LOW_COVERAGE   the labelled points leave more than 10 % of the squares empty (cells_hit)
 * Recommended rule (the DEFAULT of the class mirrors' get_corners; each signal has its own switch there,
 * accept_ambiguous / accept_low_coverage, because the reference leaves the decision to the operator): accept iff
 * status == ILCC_OK and the flag is clear.
 * On 2 x 1024 synthetic VLP-16 frames at 2-3.5 m (profiles/r03_confidence_study.json) that rule accepts 1916 of the 1959
 * ILCC_OK frames, rejects all 8 whose corners are a full square (150 mm) off, and the worst accepted frame is 16.7 mm
 * off; status alone accepts those 8. */

typedef struct ilcc_result {
  int32_t status;
  int32_t n_points;            /* input points of the frame */
  int32_t n_roi, n_cluster, n_plane;
  int32_t n_black, n_gray, n_white;
  int32_t n_corners;
  int32_t phase;               /* topleftWhite chosen (0/1) */
  int32_t iters_a, iters_b;    /* REFERENCE_LOCAL: trust-region iterations of pass A / B; GRID: refinement rounds / basin hops */
  int32_t grid_index;          /* ((k*n_ty+a)*n_tz+b)*2+phase of the grid argmin, -1 if unused */
  int32_t found_board;         /* 1: the cluster holding the click's nearest point was admissible
                                  (find_board of get_chessboard_by_point, LidarCornersEst.cpp:91-102) */
  float grid_cost;
  float plane[4];              /* refit plane nx,ny,nz,d of getPlane */
  float pca[16];               /* row-major 4x4 pca_matrix (lidar -> plane frame) */
  double gray_zone[2];
  double theta_t[3];
  double cost_a, cost_b;       /* REFERENCE_LOCAL: final cost of pass A / pass B; GRID: cost_a = sel_cost, cost_b = cost of
                                  the cheapest neighbouring basin */
  double sel_cost;             /* with-OOB cost at theta_t */
  double basin_margin;         /* GRID: (cost_b - sel_cost) / sel_cost -- how much worse the best one-square-shifted
                                  alternative is; 0 = the data cannot tell the two apart */
  int32_t flags;               /* ILCC_FLAG_* */
  int32_t grid_ties;           /* GRID: candidates the grid pass listed within 2e-5 of its minimum */
  int32_t cells_hit;           /* board squares (of board_w x board_h) that hold >= 1 labelled point under the final pose */
  int32_t n_oob;               /* labelled points outside the board under the final pose */
  float corners[ILCC_MAX_CORNERS * 3]; /* x y z, outer loop short board axis, inner long axis */
} ilcc_result;

typedef struct ilcc_handle ilcc_handle;

/* per-stage device time of the last batch call, ms (HIP events on the handle's stream) */
typedef struct ilcc_timing {
  float roi_crop, cluster, ransac_plane, plane_frame_hist, grid_cost, refine_corners, total;
  uint32_t grid_cost_launches;   /* kernel launches accumulated since ilcc_reset_timing */
  double grid_cost_ms_sum;       /* their summed HIP-event duration, ms */
  uint64_t grid_cost_evals_sum;  /* point x candidate evaluations they performed (both phases = 1) */
  uint64_t grid_cost_evals_nominal_sum; /* evaluations an unpruned exhaustive pass needs */
  uint64_t grid_cost_evals_interior_sum; /* part of grid_cost_evals_sum spent on points that cannot leave the board under any
                                            translation of the grid (cheaper term: no out-of-board logic) */
  uint64_t grid_cost_box_evals_sum; /* (point, 16-candidate tile) evaluations of the box pre-pass: a lower bound for a whole
                                       tile at once (not part of grid_cost_evals_sum) */
  /* ABI 4/5: the K6 stage launch by launch.  grid_cost_ms_sum above is the SPAN of the stage between two HIP events on the
   * batch's stream.  The stage's launches are also bracketed one by one -- locate launches (seed, refinement, anchor),
   * common pre-pass, full pass -- by HIP events on the same stream: EVENT SPANS per launch, i.e. a kernel's duration plus,
   * with other batches in flight, whatever it waited for a CU behind the event in front of it.  With one batch in flight
   * (nothing else on the chip) they are what a rocprofv3 kernel trace of the same run adds up to. */
  double grid_cost_kernel_ms_sum;   /* sum over batches of the event spans of seed + refinement + anchor + common pre-pass + full pass, ms */
  double grid_cost_full_ms_sum;     /* the full pass alone */
  double walk_order_ms_sum;         /* K5w (once per frame, in front of the K6 launches) */
  double grid_cost_prepass_ms_sum;  /* ABI 5: the common pre-pass (k6_group_prepass) alone; it is part of grid_cost_kernel_ms_sum */
  double grid_cost_locate_ms_sum;   /* ABI 5: seed + refinement + anchor (the launches that only locate the minimum and publish the bound) */
  /* ABI 5: every stage accumulated over the batches since ilcc_reset_timing (the float fields above are the LAST batch only) */
  uint64_t batches;                 /* batches accounted */
  double stage_ms_sum[7];           /* roi_crop, cluster, ransac_plane, plane_frame_hist, grid_cost, refine_corners, total */
  double roi_count_ms_sum;          /* K1's count pass alone -- the kernel that reads every input point once (the HBM-bound one) */
  uint64_t online_second_tier_frames; /* ilcc_chessboard_by_point_batch: frames whose answer needed the whole cloud clustered (the first
                                         tier answers from a window around the predicted point and verifies it) */
} ilcc_timing;

int32_t ilcc_abi_version(void);
const char* ilcc_strerror(int32_t status);
const char* ilcc_last_error(const ilcc_handle* h);

/* reference constants (every one is hard-coded in the reference, see field comments) */
void ilcc_default_params(ilcc_params* p);

/* LidarCornersEst::set_chessboard_param (LidarCornersEst.cpp:20-46): reads grid_length,
 * corner_in_x, corner_in_y from an OpenCV-YAML file, squares = corners+1, sorted ascending. */
int32_t ilcc_set_chessboard_param(ilcc_params* p, const char* cam_yaml);

/* device < 0: current device.  max_frames / max_total_points bound one batch call.  A fresh handle sizes its on-chip
 * staging from the batches it sees: call ilcc_reserve right after ilcc_create when the FIRST batch's latency matters. */
ilcc_handle* ilcc_create(int32_t device, const ilcc_params* p, uint32_t max_frames,
                         uint64_t max_total_points);
void ilcc_destroy(ilcc_handle* h);
int32_t ilcc_set_params(ilcc_handle* h, const ilcc_params* p);
/* Optional sizing hint (ABI 3).  Two on-chip staging capacities follow the data: the labelled (black/white) points per
 * frame the grid search keeps in LDS, and the ROI points per frame the clustering workgroup keeps in LDS.  A fresh handle
 * starts small (1024 / 2048) and grows them after each batch to what that batch needed; a frame above the current
 * capacity takes a slower path with IDENTICAL results (points walked through L2; clustering by one workgroup in global
 * memory), so only the first such batch is slower.  ilcc_reserve sets the capacities up front so that the first batch
 * runs like the hundredth: pass the largest counts expected (e.g. 2000 / 2500 for VLP-16 at 2-3.5 m; 4500 / 25000 for a
 * 64-ring sensor).  roi_points_per_frame > 4096 also arms the multi-workgroup clustering kernels. */
int32_t ilcc_reserve(ilcc_handle* h, uint32_t labelled_points_per_frame, uint32_t roi_points_per_frame);

/* one frame, host buffers: xyzi = n x {x,y,z,intensity} float32 (pcl::PointXYZI payload) */
int32_t ilcc_extract(ilcc_handle* h, const float* xyzi, uint32_t n, const float click[3],
                     ilcc_result* out);

/* batch, host buffers.  offsets[f]..offsets[f+1] = point range of frame f (n_frames+1 entries,
 * offsets[0] = 0).  clicks = n_frames x 3.  out = n_frames records. */
int32_t ilcc_extract_batch(ilcc_handle* h, const float* xyzi, const uint64_t* offsets,
                           uint32_t n_frames, const float* clicks, ilcc_result* out);

/* batch, inputs already resident in HBM (d_xyzi, d_clicks device pointers; offsets on host).
 * out is a host buffer. */
int32_t ilcc_extract_batch_device(ilcc_handle* h, const float* d_xyzi, const uint64_t* offsets,
                                  uint32_t n_frames, const float* d_clicks, ilcc_result* out);

/* Asynchronous form of ilcc_extract_batch_device: enqueue the whole path for one batch and return;
 * up to 4 batches may be in flight per handle (each in its own buffers and stream; run the process with
 * GPU_MAX_HW_QUEUES >= 5, HIP's default of 4 makes two of the streams share a hardware queue), so the short
 * latency-bound stages of one batch overlap with the grid search of another.  *ticket identifies the
 * batch; ilcc_wait blocks until it is complete and copies its records to out (n_frames entries).
 * (The library never touches the environment: the HOST exports GPU_MAX_HW_QUEUES=8 before the HIP runtime starts --
 * the Python package does it on import, the C++ mains in host/ do it first thing in main().  The first use of the
 * fourth slot prints a one-time note when the variable is absent or below 5.)
 * Tickets must be waited for in submission order once all slots are taken (ILCC_CAPACITY otherwise).
 * The inputs must stay valid and unchanged until the matching ilcc_wait returns. */
int32_t ilcc_submit_batch_device(ilcc_handle* h, const float* d_xyzi, const uint64_t* offsets,
                                 uint32_t n_frames, const float* d_clicks, int32_t* ticket);
/* The same with HOST inputs (SURVEY.md 8d counts this copy): the batch's H2D copy is enqueued on the slot's own
 * stream in front of its kernels, so it overlaps with the kernels of the other batches in flight and the host never
 * blocks on it.  xyzi should be page-locked (hipHostMalloc / hipHostRegister / torch pin_memory): the copy of a
 * pageable buffer is staged by the runtime and serialises.  xyzi and clicks must stay valid until ilcc_wait. */
int32_t ilcc_submit_batch(ilcc_handle* h, const float* xyzi, const uint64_t* offsets, uint32_t n_frames,
                          const float* clicks, int32_t* ticket);
int32_t ilcc_wait(ilcc_handle* h, int32_t ticket, ilcc_result* out);
/* ilcc_wait that also leaves, in device memory, the fixed-size records the multi-GPU gather ships
 * (one collective of corner records per step): d_records[n_frames][ILCC_RECORD_HEADER + 3*n_corners]
 * floats = status, n_corners, phase, grid_index, iters_a, iters_b, cost_a, cost_b, sel_cost, theta,
 * ty, tz, n_plane, n_black, n_white, basin_margin, tag (= tag_base + frame index: lets the receiver check WHOSE
 * record sits where), check (24-bit xor-fold of the corner bits and the tag: lets it check the contents), flags, n_roi,
 * then x y z per corner (zero beyond the frame's n_corners).
 * Complete on return, so the caller can hand the buffer to RCCL on any stream.  out may be NULL (ABI 4): nothing but the
 * device-side records is produced then. */
#define ILCC_RECORD_HEADER 20
int32_t ilcc_wait_records_device(ilcc_handle* h, int32_t ticket, ilcc_result* out, void* d_records, uint32_t n_corners,
                                 uint32_t tag_base);

/* Result traffic (ABI 4).  SURVEY.md 8(d) counts 12 * n_corners + 64 bytes of result per frame; ilcc_result is a
 * 3.3 KB record sized for ILCC_MAX_CORNERS.  Two things keep the copy back near the algorithmic size:
 *   - ilcc_wait copies, per frame, only the record's head and the corners of the handle's board
 *     (offsetof(ilcc_result, corners) + 12 * (board_w - 1) * (board_h - 1) bytes: 652 B for the 7 x 5 board); corners beyond
 *     the board's count are left as the caller passed them;
 *   - ILCC_RESULTS_COMPACT: the batch's own stream packs the gather records described above (K9: ILCC_RECORD_HEADER
 *     floats + 3 per corner = 500 B for the 7 x 5 board; tag = frame index within the batch; slot 19 = n_roi) and ONLY
 *     those cross PCIe; ilcc_wait_compact hands them over.  The full records stay in HBM: ilcc_fetch_results reads
 *     them for the last completed batch.  (Round 6: K9 STORES the records into the handle's pinned, mapped staging itself -- no
 *     device-to-host copy command is queued behind the batch's kernels, so the copy engine that carries the next batches' input
 *     copies never waits on one: tools/dev_h2d_probe.py.)
 * Both waits work in both modes -- the mode decides which copy is enqueued with the batch (the other one is then a
 * synchronous copy inside the wait).  Default: ILCC_RESULTS_FULL. */
enum { ILCC_RESULTS_FULL = 0, ILCC_RESULTS_COMPACT = 1 };
int32_t ilcc_set_result_mode(ilcc_handle* h, int32_t mode);   /* no batch may be in flight */
/* records: host memory for n_frames x (ILCC_RECORD_HEADER + 3 * n_corners) floats, n_corners = (board_w - 1) * (board_h - 1) of
 * the parameters the batch was SUBMITTED with.  capacity_floats (ABI 5) = the floats `records` can hold: the call returns
 * ILCC_BAD_ARGUMENT (nothing copied, the batch stays waitable) when that is less than the batch needs. */
int32_t ilcc_wait_compact(ilcc_handle* h, int32_t ticket, float* records, uint64_t capacity_floats);
/* floats per record of the batch in flight under `ticket` (0: no such batch) */
uint32_t ilcc_record_floats(const ilcc_handle* h, int32_t ticket);
/* full records [first, first + n) of the last completed batch, copied from HBM (synchronous) */
int32_t ilcc_fetch_results(ilcc_handle* h, uint32_t first, uint32_t n, ilcc_result* out);

/* LidarCornersEst::get_chessboard_by_point (LidarCornersEst.cpp:72-115), the front half of the path as
 * the online node uses it (ilcc2/test/lidar_chessboard_online.cpp:91-101): NO ROI crop, Euclidean
 * clustering of the whole cloud with params.online_cluster_tol (0.10, LidarCornersEst.cpp:80), the cluster
 * around `points[f]`, getPlane, then get_gray_zone(rate = params.gray_rate) for colouring.
 * status per frame: ILCC_OK (reference: true) or ILCC_BOARD_NOT_FOUND / another failure (false).
 * The plane cloud (`outcloud`) is ILCC_CLOUD_CHESSBOARD; ilcc_fetch_classes gives the
 * color_by_gray_zone class of each of its points.  Host buffers; batched like ilcc_extract_batch. */
int32_t ilcc_chessboard_by_point_batch(ilcc_handle* h, const float* xyzi, const uint64_t* offsets,
                                       uint32_t n_frames, const float* points, int32_t min_plane_points,
                                       ilcc_result* out);
/* The same call in two halves (like ilcc_submit_batch / ilcc_wait: up to four calls in flight per handle, the H2D copy of one
 * overlapping the kernels of the others; `xyzi` and `points` must stay valid -- and should be pinned -- until the wait returns).
 * ilcc_fetch_cloud / ilcc_fetch_classes then refer to the call last waited for.  For callers that hold more than one scan at a
 * time (a recorded sequence); the online node (lidar_chessboard_online.cpp:91-101) has one and uses the call above. */
int32_t ilcc_submit_chessboard_by_point(ilcc_handle* h, const float* xyzi, const uint64_t* offsets, uint32_t n_frames,
                                        const float* points, int32_t* ticket);
int32_t ilcc_wait_chessboard_by_point(ilcc_handle* h, int32_t ticket, int32_t min_plane_points, ilcc_result* out);
/* color_by_gray_zone classes (LidarCornersEst.cpp:452-499) of the last batch's plane cloud:
 * 0 black (I < gray_zone[0]), 1 gray, 2 white (I > gray_zone[1]).  Returns the point count. */
int64_t ilcc_fetch_classes(ilcc_handle* h, uint32_t frame, uint8_t* out_class, uint64_t cap_points);

/* copy one of the last completed batch's intermediate clouds of a frame (n x 4 float32) to the host.
 * returns the point count (<= cap_points written), or a negative status. */
int64_t ilcc_fetch_cloud(ilcc_handle* h, uint32_t frame, int32_t which, float* out_xyzi,
                         uint64_t cap_points);
/* non-gray points handed to the cost: y,z (plane frame) and label (0 black, 1 white) */
int64_t ilcc_fetch_labelled(ilcc_handle* h, uint32_t frame, float* out_yz, uint8_t* out_label,
                            uint64_t cap_points);
/* the same points in the grid search's WALK layout (GRID solver, frames of at most 8192 labelled points):
 * [interior class | rim | other border-class points], each part in golden-ratio walk order; counts[0] = interior-class
 * points (in the board under every rotation and translation of the grid), counts[1] = rim points.  Test/diagnostic entry. */
int64_t ilcc_fetch_walk(ilcc_handle* h, uint32_t frame, float* out_yz, uint8_t* out_label, uint64_t cap_points,
                        uint32_t counts[2]);

/* the grid-cost kernel on caller-supplied labelled points (host buffers): full cost volume
 * [n_th][n_ty][n_tz][2 phases] (cost_out may be NULL) and the argmin exactly as the pipeline
 * selects it.  Test/diagnostic entry of the hot kernel. */
int32_t ilcc_grid_cost(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m,
                       int32_t use_oob, float* cost_out, int32_t* best_index, float* best_cost);

/* The GRID solver as the pipeline runs it on a SMALL batch -- walk layout, the three launches that locate the minimum (seed,
 * refinement, anchor rounds; batches of >= 512 frames fuse them into k6_locate, whose equality with the three launches is a test
 * of its own: test_fused_locate_equals_the_three_launches), common pre-pass, full pass with its near-tie list, then the refinement
 * with the near-tie recount and its first-round shortcut -- on caller-supplied labelled points (host buffers).  Out: the grid argmin and its fp32 cost, the refined lattice point (units of step / refine_div from the
 * grid's minima), phase, fixed-point costs (units of 2^-40) of the result and of the cheapest neighbouring basin, rounds, hops,
 * ILCC_FLAG_* and the near-tie count.  Test/diagnostic entry (adversarial inputs for the fp32 ranking). */
int32_t ilcc_grid_solve(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m, int32_t* grid_index, float* grid_cost,
                        int32_t lat[3], int32_t* phase, int64_t* cost_q, int64_t* alt_cost_q, int32_t* rounds, int32_t* hops,
                        int32_t* flags, int32_t* ties);

/* the GRID-mode refinement kernel on caller-supplied labelled points: lat[3] in/out are lattice coordinates
 * (theta, ty, tz) in units of step / refine_div from (th_min, ty_min, tz_min), *phase in/out; out: fixed-point
 * costs (units of 2^-40) of the result and of the cheapest neighbouring basin, rounds and hops executed.
 * Test/diagnostic entry. */
int32_t ilcc_pattern_refine(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m, int32_t lat[3],
                            int32_t* phase, int64_t* cost_q, int64_t* alt_cost_q, int32_t* rounds, int32_t* hops);

/* the local solver (Optimization::get_theta_t) on caller-supplied labelled points */
int32_t ilcc_get_theta_t(ilcc_handle* h, const float* yz, const uint8_t* label, uint32_t m,
                         int32_t topleft_white, int32_t use_oob, double theta_t[3], double* cost,
                         int32_t* iterations);

/* Diagnostic (ABI 5): the real timeline of the batches, without a profiler.  While enabled, every completed batch appends one row
 * of ILCC_TIMELINE_COLS doubles: slot, then the times in ms -- relative to a reference event recorded when the timeline was
 * enabled -- of the batch's HIP events: start, after K1's count pass, after K1, K2, K3, K4/K5, K5w, seed, refinement, anchor,
 * common pre-pass (= ready for the full pass), full pass start (behind the wait for the previous batch's full pass), full pass
 * end, end of K7.  ilcc_debug_timeline_fetch copies up to cap_rows rows (oldest first) and clears the log; returns the rows. */
#define ILCC_TIMELINE_COLS 15
int32_t ilcc_debug_timeline_enable(ilcc_handle* h, int32_t on);
int32_t ilcc_debug_timeline_fetch(ilcc_handle* h, double* rows, uint32_t cap_rows);

void ilcc_get_timing(const ilcc_handle* h, ilcc_timing* t);
void ilcc_reset_timing(ilcc_handle* h);

/* save_corners2txt (get_lidar_corners.cpp:27-36): "x y z\n" per corner, ostream default float
 * formatting (6 significant digits), file truncated. */
int32_t ilcc_save_corners2txt(const float* corners_xyz, uint32_t n_corners, const char* filename);
/* ImageCornersEst::read_lidar_corners (ImageCornersEst.cpp:281-299): returns corners read */
int32_t ilcc_read_lidar_corners(const char* filename, uint32_t num, double* out_xyz);

#ifdef __cplusplus
}
#endif
#endif

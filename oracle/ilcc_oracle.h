/*
 * ilcc_oracle.h -- CPU restatement (plain C) of ilcc2's LiDAR chessboard-corner
 * extraction path.  TEST INFRASTRUCTURE ONLY.
 *
 *   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 *   load this library.  The product (libilcc_hip.so) never links, includes or
 *   calls anything in oracle/.
 *
 * PARITY STATUS: **parity unpinned** for whole-path input->output.  The reference
 * cannot be compiled here (needs PCL/Eigen/Ceres/OpenCV/ROS, all absent), ships no
 * tests and its six input bags are stripped.  What IS pinned, and what
 * tests/test_oracle_golden.py checks this file against:
 *   - the Ceres trust-region restatement against an OUTPUT OF THE REFERENCE: run on the
 *     shipped corner-file pairs (ilcc2/process_data/pointgrey*.txt) it reproduces the
 *     shipped extrinsic ilcc2/config/pointgrey.bin to 3e-16 (orc_solve_pose_3d2d),
 *   - the cost functor's hand-derived known-answer table (SURVEY.md App. C,
 *     derived from ilcc2/include/ilcc2/Optimization.h:31-107),
 *   - the six bundled output files ilcc2/process_data/pointgrey_lidar_{1..6}.txt
 *     (format, count, ordering, exact 0.15 m planar lattice),
 *   - the writer format of ilcc2/test/get_lidar_corners.cpp:27-36.
 * Third-party arithmetic (PCL 1.7/1.8, Eigen 3.3, Ceres 1.14 -- none pinned by the
 * reference, ilcc2/CMakeLists.txt:18-37) is restated from the published
 * algorithms; every function cites the reference call site it follows.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef ILCC_ORACLE_H_
#define ILCC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_CORNERS 256

/* status codes (shared meaning with include/ilcc_hip.h, but defined independently) */
enum {
  ORC_OK = 0,
  ORC_NO_ROI_POINTS = 1,
  ORC_NO_CLUSTER = 2,        /* reference: cluster_indices.at(0) throws, LidarCornersEst.cpp:167 */
  ORC_NO_PLANE = 3,          /* reference: PCL_ERROR, LidarCornersEst.cpp:206-209 */
  ORC_DEGENERATE_HIST = 4,   /* reference: UB (iter++ past rend), LidarCornersEst.cpp:266-282 */
  ORC_TOO_FEW_POINTS = 5,
  ORC_BAD_ARGUMENT = 6,
  ORC_BOARD_NOT_FOUND = 10,  /* get_chessboard_by_point returned false, LidarCornersEst.cpp:111-112 */
  ORC_AMBIGUOUS = 11         /* ORC_SOLVER_GRID: a neighbouring one-square-shifted basin costs (almost) the same; corners are
                                still written (the reference has no such signal: a human looks at the viewer and presses 'd') */
};

/* solver selection for orc_extract */
enum {
  ORC_SOLVER_REFERENCE_LOCAL = 0, /* get_corners default trajectory: topleftWhite=false, pass A then B from 0 */
  ORC_SOLVER_GRID = 1             /* exhaustive (theta,ty,tz) x phase grid on the pass-A cost, then a monotone pattern search on the
                                     same cost (lattice of step/refine_div) and a check of the eight neighbouring basins */
};

typedef struct orc_params {
  /* setROI, LidarCornersEst.cpp:54,59,64 : half extents x,y,z */
  double roi_half[3];
  /* EuclideanCluster, LidarCornersEst.cpp:131-133 */
  double cluster_tol;
  int32_t cluster_min;
  int32_t cluster_max;
  /* getPlane, LidarCornersEst.cpp:201 */
  double ransac_thresh;
  int32_t ransac_hyp;       /* SACSegmentation::max_iterations_ (PCL default 50): bound on the 3-point hypotheses that count as iterations;
                               with ransac_probability <= 0: their fixed number (rounds 1-4) */
  uint32_t ransac_seed;
  /* calHist / get_gray_zone, LidarCornersEst.cpp:226,371 */
  int32_t hist_bins;
  double gray_rate;
  /* get_theta_t, Optimization.cpp:137 */
  double huber_delta;
  /* board, LidarCornersEst.cpp:30-39 : squares, sorted so board_w <= board_h */
  double grid_length;
  int32_t board_w;
  int32_t board_h;
  /* solver */
  int32_t solver;
  int32_t accum_float;      /* 1: centroid/covariance accumulate in float like PCL; 0: double */
  int32_t phase_mode;       /* REFERENCE_LOCAL: 0 topleftWhite=false (reference's first turn), 1 true,
                               2 try both from zero, keep lower with-OOB cost (stands in for key 'd') */
  /* grid (ORC_SOLVER_GRID): theta in [th_min, th_min+(n_th-1)*th_step], ty,tz likewise */
  int32_t n_th, n_ty, n_tz;
  double th_min, th_step;
  double ty_min, ty_step;
  double tz_min, tz_step;
  /* ORC_SOLVER_GRID refinement (orc_pattern_refine) */
  int32_t refine_div;        /* finest lattice = grid step / refine_div (power of two, default 16; 0: keep the grid argmin) */
  int32_t refine_max_rounds; /* bound on the 27-candidate rounds of one pattern search (default 64) */
  int32_t refine_th_margin;  /* the search may leave the grid's theta range by this many grid steps (default 32) */
  int32_t refine_pad_;
  double ambiguity_eps;      /* ORC_AMBIGUOUS when (best neighbouring basin - cost) / cost < eps (default 1.0: an alternative must cost at least twice as much; <= 0: never) */
  double min_cell_coverage;  /* ORC_FLAG_LOW_COVERAGE when fewer than this fraction of the board's squares hold a labelled point under the final pose (default 0.9; <= 0: never) */
  double ransac_probability; /* SACSegmentation::probability_ (PCL default 0.99): RANSAC stops once iterations >= log(1 - p) / log(1 - w^3),
                                w = best inlier share so far (pcl::RandomSampleConsensus::computeModel); <= 0: ransac_hyp hypotheses, no early stop */
} orc_params;

typedef struct orc_result {
  int32_t status;
  int32_t n_roi, n_cluster, n_plane;
  int32_t n_black, n_gray, n_white;
  int32_t n_corners;
  int32_t phase;               /* topleftWhite used (0/1) */
  int32_t iters_a, iters_b;    /* trust-region iterations of pass A / B */
  int32_t grid_index;          /* winning candidate (ORC_SOLVER_GRID), else -1 */
  double gray_zone[2];
  double theta_t[3];
  double cost_a, cost_b;       /* final cost of pass A (OOB on) and B (OOB off) */
  double sel_cost;             /* with-OOB cost at the final theta_t (phase selection metric) */
  double grid_cost;
  double basin_margin;         /* ORC_SOLVER_GRID: (cost of the best one-square-shifted basin - sel_cost) / sel_cost */
  int32_t flags;               /* ORC_FLAG_* (the HIP library's ILCC_FLAG_* without the fp32-specific tie overflow) */
  int32_t cells_hit;           /* board squares that hold >= 1 labelled point under the final (theta, ty, tz) */
  int32_t n_oob;               /* labelled points outside the board under the final pose */
  int32_t pad_;
  float pca[16];               /* row-major 4x4, lidar -> plane frame (pca_matrix) */
  float corners[ORC_MAX_CORNERS * 3];
} orc_result;

#define ORC_FLAG_REFINE_CAPPED 2
#define ORC_FLAG_LOW_COVERAGE 4

void orc_default_params(orc_params* p);

/* board squares holding >= 1 labelled point / labelled points outside the board under theta_t: the functor's own
 * coordinates (Optimization.h:37-49), double.  The operator's eye at the viewer (LidarCornersEst.cpp:415-441) checks
 * exactly this -- does the virtual board sit ON the points -- before pressing 'o'. */
void orc_coverage(const double theta_t[3], const float* y, const float* z, int32_t m, const orc_params* p, int32_t* cells_hit,
                  int32_t* n_oob);

/* ---- stage functions (each usable on its own from tests) ---- */

/* a1 setROI: returns m, writes kept original indices (ascending) */
int32_t orc_roi_crop(const float* xyzi, int32_t n, const float click[3], const orc_params* p,
                     int32_t* out_idx);

/* a2 EuclideanCluster (without the UI): in = roi points (m x 4 floats).
 * returns size of the chosen cluster (0 if none), indices (into roi, ascending) in out_idx.
 * labels_out (optional, m ints): component id per point (= smallest member index). */
int32_t orc_cluster(const float* roi, int32_t m, const float click[3], const orc_params* p,
                    int32_t* out_idx, int32_t* labels_out);

int32_t orc_cluster2(const float* roi, int32_t m, const float click[3], const orc_params* p,
                     int32_t* out_idx, int32_t* labels_out, int32_t* found_out);

/* a3 getPlane: in = cluster points (m x 4). returns inlier count, indices ascending.
 * plane_out (optional): refit plane nx,ny,nz,d */
int32_t orc_ransac_plane(const float* pts, int32_t m, const orc_params* p, int32_t* out_idx,
                         float plane_out[4]);

/* a4 transformbyPCA: in = plane points (m x 4); out pca[16] row-major, pts_pca (m x 4, intensity carried) */
int32_t orc_plane_frame(const float* pts, int32_t m, const orc_params* p, float pca[16],
                        float* pts_pca);

/* a5 calHist + get_gray_zone */
int32_t orc_gray_zone(const float* intensity, int32_t m, const orc_params* p, double rlrh[2],
                      double gray_zone[2]);

/* a6 VirtualboardError::operator() -- raw residual, optional raw jacobian d r / d(theta,ty,tz) */
double orc_residual(const double theta_t[3], double y, double z, int32_t board_w, int32_t board_h,
                    double g, int32_t topleft_white, int32_t laser_white, int32_t use_oob,
                    double jac[3]);

/* half * sum rho(r^2) over labelled points. label: 0 black, 1 white, anything else skipped */
double orc_cost(const double theta_t[3], const float* y, const float* z, const int8_t* label,
                int32_t m, const orc_params* p, int32_t topleft_white, int32_t use_oob);

/* a7 get_theta_t : classify by gray zone and minimise (Ceres-1.14-like TRUST_REGION/SUBSPACE_DOGLEG).
 * pts_pca: m x 4 (x,y,z,intensity). theta_t in/out. returns iterations used; final cost in *cost */
int32_t orc_get_theta_t(const float* pts_pca, int32_t m, const double gray_zone[2],
                        const orc_params* p, int32_t topleft_white, int32_t use_oob,
                        double theta_t[3], double* cost);

/* exhaustive grid (spec of the GPU search): both phases, chosen OOB flag; returns winning flat index
 * ((k*n_ty + a)*n_tz + b)*2 + phase ; cost_out (optional) full volume [n_th*n_ty*n_tz*2].
 * Candidates are ordered by the fixed-point cost (orc_cost_q); the doubles returned are cost_q * 2^-40. */
int32_t orc_grid_search(const float* y, const float* z, const int8_t* label, int32_t m,
                        const orc_params* p, int32_t use_oob, double* best_cost,
                        double* cost_out);

/* ORC_SOLVER_GRID works on a FIXED-POINT cost: sum over the labelled points of rint(1/2 rho(r^2) * 2^40).
 * Integer sums do not depend on the summation order, so a parallel reduction (the GPU) and this serial loop
 * agree bit for bit, and ties are exact ties.  One quantum (2^-40 ~ 9e-13) is far below the fp64 noise of
 * a ~1e3-term sum of terms <= 1.  The per-point residual is the functor's (a6) evaluated with a reciprocal
 * scaling instead of the division by g and a square-root-free Huber (see term_q in the .c file): within a few
 * quanta of orc_cost's terms, and cheap enough for the GPU to take thousands of them per candidate stencil. */
#define ORC_COST_Q_ONE 1099511627776.0 /* 2^40 */
int64_t orc_cost_q(const double theta_t[3], const float* y, const float* z, const int8_t* label,
                   int32_t m, const orc_params* p, int32_t topleft_white, int32_t use_oob);

/* refinement of ORC_SOLVER_GRID.  lat[3] in/out: lattice coordinates (theta, ty, tz) in units of
 * step / refine_div relative to (th_min, ty_min, tz_min); phase in/out.  Pattern search: evaluate the 26
 * neighbours at the current stride (refine_div, refine_div/2, ..., 1 lattice units), move to the cheapest if it
 * is strictly cheaper than the centre (ties: smaller index distance, then index order), else halve the stride.
 * Then the eight neighbouring basins (one square along y and/or z; an odd shift flips the colour phase) are
 * evaluated once; a cheaper one is adopted and refined again (at most two hops).  Returns the final cost,
 * *alt_cost = cheapest neighbouring basin, rounds / hops executed. */
int64_t orc_pattern_refine2(const float* y, const float* z, const int8_t* label, int32_t m, const orc_params* p, int32_t lat[3],
                            int32_t* phase, int64_t* alt_cost, int32_t* rounds, int32_t* hops, int32_t* capped);
int64_t orc_pattern_refine(const float* y, const float* z, const int8_t* label, int32_t m, const orc_params* p,
                           int32_t lat[3], int32_t* phase, int64_t* alt_cost, int32_t* rounds, int32_t* hops);
/* lattice -> (theta, ty, tz) */
void orc_lattice_point(const orc_params* p, const int32_t lat[3], double theta_t[3]);

/* a9 getPCDcorners (inverse=false) */
int32_t orc_corners(const float pca[16], const double theta_t[3], const orc_params* p,
                    float* corners);

/* whole path for one frame. debug clouds optional (NULL ok): each n x 4 floats, capacity n. */
int32_t orc_extract(const float* xyzi, int32_t n, const float click[3], const orc_params* p,
                    orc_result* out, float* cloud_chessboard, float* cloud_pca);

/* f2 get_chessboard_by_point + color_by_gray_zone classes (0 black, 1 gray, 2 white); out->phase
 * carries find_board.  cloud_chessboard: n x 4 floats, classes: n bytes (NULL ok). */
int32_t orc_chessboard_by_point(const float* xyzi, int32_t n, const float point[3], const orc_params* p,
                                int32_t min_plane, orc_result* out, float* cloud_chessboard,
                                uint8_t* classes);

/* a11 save_corners2txt formatting of one float (ostream default, precision 6) into buf */
int32_t orc_format_float(float v, char* buf, int32_t cap);

/* Pose3d2dError + HuberLoss(0.1) through the SAME trust-region code as the path's board fit
 * (src/Optimization.cpp:13-91).  Not on the hot path: it is here because the reference ships the
 * inputs and the output of this solve, which pins the solver restatement against a real Ceres run. */
int32_t orc_solve_pose_3d2d(const double* pts3d, const double* pts2d, int32_t n, const double camera[4],
                            double r[3], double t[3], double* final_cost);

/* f4 projection (after calibration): spaceToPlane + HSVtoRGB + the per-point loops of
 * test/pcd2image.cpp:56-82 and test/rgblidar.cpp:50-74 (no drawing). */
typedef struct orc_camera_model {
  double R[9], t[3];
  double fx, cx, fy, cy;
  int32_t width, height;
} orc_camera_model;
typedef struct orc_pixel_hit {
  int32_t x, y;
  uint8_t r, g, b, pad;
  uint32_t index;
} orc_pixel_hit;
void orc_hsv_to_rgb(int32_t h, int32_t s, int32_t v, uint8_t rgb[3]);
int32_t orc_project_intensity(const float* xyzi, int32_t n, const orc_camera_model* cam, double dis, double lo,
                              double hi, orc_pixel_hit* hits);
int32_t orc_colourise(const float* xyzi, int32_t n, const orc_camera_model* cam, double dis, const uint8_t* image_bgr,
                      uint32_t image_step, float* xyzrgb);

#ifdef __cplusplus
}
#endif
#endif

/*
 * ilcc_oracle.c -- CPU restatement of ilcc2's LiDAR chessboard-corner extraction.
 * TEST INFRASTRUCTURE ONLY (see ilcc_oracle.h).  PARITY UNPINNED for whole-path input->output:
 * the reference cannot be built here and ships neither tests nor inputs.  The solver part is
 * pinned against the reference's shipped Ceres output (config/pointgrey.bin), see ilcc_oracle.h.
 *
 * Compile with -ffp-contract=off and without -ffast-math: the float stages are written
 * as plain (unfused) float expressions, which is what an -O3 x86-64 build of PCL computes.
 *
 * Citations are file:line under /root/reference/.
 */
#include "ilcc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* params                                                                    */
/* ------------------------------------------------------------------------- */

void orc_default_params(orc_params* p) {
  memset(p, 0, sizeof(*p));
  p->roi_half[0] = 1.0; /* LidarCornersEst.cpp:59 x */
  p->roi_half[1] = 1.5; /* :64 y */
  p->roi_half[2] = 2.0; /* :54 z */
  p->cluster_tol = 0.12; /* :131 */
  p->cluster_min = 100;  /* :132 */
  p->cluster_max = 25000; /* :133 */
  p->ransac_thresh = 0.03; /* :201 */
  p->ransac_hyp = 50;           /* SACSegmentation: max_iterations_ (50) */
  p->ransac_probability = 0.99; /* SACSegmentation: probability_ (0.99) */
  p->ransac_seed = 12345u; /* PCL seeds its sampler with 12345 when random=false */
  p->hist_bins = 100;      /* :226 */
  p->gray_rate = 2.5;      /* :371 */
  p->huber_delta = 0.1;    /* Optimization.cpp:137 */
  p->grid_length = 0.15;   /* config/pointgrey.yaml:17 */
  p->board_w = 6;          /* corner_in_y+1, sorted ascending, LidarCornersEst.cpp:31-39 */
  p->board_h = 8;
  p->solver = ORC_SOLVER_REFERENCE_LOCAL;
  p->accum_float = 0;
  p->phase_mode = 2;
  /* default coarse grid: ty,tz in [-g, g) step g/20, theta in [-15,15] deg step 0.5 deg */
  p->n_th = 61;
  p->th_step = 0.5 * M_PI / 180.0;
  p->th_min = -15.0 * M_PI / 180.0;
  p->n_ty = 40;
  p->ty_step = 0.15 / 20.0;
  p->ty_min = -0.15;
  p->n_tz = 40;
  p->tz_step = 0.15 / 20.0;
  p->tz_min = -0.15;
  p->refine_div = 16;
  p->refine_max_rounds = 64;
  p->min_cell_coverage = 0.9;
  p->refine_th_margin = 32;
  p->ambiguity_eps = 1.0;
}

/* ------------------------------------------------------------------------- */
/* a1  setROI  (LidarCornersEst.cpp:48-70, three pcl::PassThrough in z,x,y)    */
/* ------------------------------------------------------------------------- */
/* pcl::PassThrough::setFilterLimits takes floats; the reference passes
 * point.z-2.0 (float - double -> double) which is then narrowed to float.
 * A point is removed when x,y or z is not finite, or value < min || value > max. */
int32_t orc_roi_crop(const float* xyzi, int32_t n, const float click[3], const orc_params* p,
                     int32_t* out_idx) {
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = (float)((double)click[a] - p->roi_half[a]);
    hi[a] = (float)((double)click[a] + p->roi_half[a]);
  }
  int32_t m = 0;
  for (int32_t i = 0; i < n; ++i) {
    const float x = xyzi[4 * i + 0], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2];
    if (!isfinite(x) || !isfinite(y) || !isfinite(z)) continue;
    if (z < lo[2] || z > hi[2]) continue; /* :53-55 */
    if (x < lo[0] || x > hi[0]) continue; /* :58-60 */
    if (y < lo[1] || y > hi[1]) continue; /* :63-65 */
    out_idx[m++] = i;
  }
  return m;
}

/* ------------------------------------------------------------------------- */
/* a2  EuclideanCluster (LidarCornersEst.cpp:124-153)                          */
/* ------------------------------------------------------------------------- */
/* pcl::EuclideanClusterExtraction: for each unprocessed seed in index order, BFS with
 * radiusSearch(tol); a component is kept iff min <= size <= max; afterwards clusters are
 * sorted by size, largest first.  FLANN's radius search keeps neighbours with
 * squared L2 distance (accumulated in float, dx*dx + dy*dy + dz*dz) < (float)(tol*tol).
 * Then nearestKSearch(click,1) and the first cluster (in sorted order) that contains
 * that index is chosen; if none does, index 0 (the largest) stays selected (:144-153,158-167).
 *
 * Deviations (documented in DESIGN.md): members are returned in ascending index order
 * (PCL: BFS discovery order, which depends on FLANN's tree); equal-size clusters are
 * ordered by their smallest member (PCL: std::sort, unspecified).  Neither changes which
 * points form the board. */

typedef struct {
  uint64_t key;
  int32_t idx;
} cell_ent;

static int cmp_cell(const void* a, const void* b) {
  const cell_ent* x = (const cell_ent*)a;
  const cell_ent* y = (const cell_ent*)b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

static uint64_t cell_key(int64_t cx, int64_t cy, int64_t cz) {
  return ((uint64_t)(cx & 0x1FFFFF) << 42) | ((uint64_t)(cy & 0x1FFFFF) << 21) |
         (uint64_t)(cz & 0x1FFFFF);
}

static int32_t lower_bound_cell(const cell_ent* e, int32_t m, uint64_t key) {
  int32_t lo = 0, hi = m;
  while (lo < hi) {
    int32_t mid = (lo + hi) >> 1;
    if (e[mid].key < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

int32_t orc_cluster(const float* roi, int32_t m, const float click[3], const orc_params* p,
                    int32_t* out_idx, int32_t* labels_out) {
  return orc_cluster2(roi, m, click, p, out_idx, labels_out, NULL);
}

/* found_out: 1 when the cluster holding the click's nearest point is admissible (find_board of
 * get_chessboard_by_point, LidarCornersEst.cpp:91-102) */
int32_t orc_cluster2(const float* roi, int32_t m, const float click[3], const orc_params* p,
                     int32_t* out_idx, int32_t* labels_out, int32_t* found_out) {
  if (found_out) *found_out = 0;
  if (m <= 0) return 0;
  const float tol2 = (float)(p->cluster_tol * p->cluster_tol);
  const double cell = p->cluster_tol;

  /* uniform grid (cell = tol) only to accelerate the exact radius query */
  double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX};
  for (int32_t i = 0; i < m; ++i)
    for (int a = 0; a < 3; ++a)
      if (roi[4 * i + a] < mn[a]) mn[a] = roi[4 * i + a];
  cell_ent* ents = (cell_ent*)malloc(sizeof(cell_ent) * (size_t)m);
  int64_t* cc = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)m);
  for (int32_t i = 0; i < m; ++i) {
    for (int a = 0; a < 3; ++a) cc[3 * i + a] = (int64_t)floor((roi[4 * i + a] - mn[a]) / cell);
    ents[i].key = cell_key(cc[3 * i], cc[3 * i + 1], cc[3 * i + 2]);
    ents[i].idx = i;
  }
  qsort(ents, (size_t)m, sizeof(cell_ent), cmp_cell);

  int32_t* label = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
  int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
  for (int32_t i = 0; i < m; ++i) label[i] = -1;

  /* clusters: (seed=label id, size) */
  int32_t n_clusters = 0;
  int32_t* cl_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
  int32_t* cl_size = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);

  for (int32_t s = 0; s < m; ++s) {
    if (label[s] >= 0) continue;
    int32_t qh = 0, qt = 0;
    queue[qt++] = s;
    label[s] = s;
    while (qh < qt) {
      const int32_t i = queue[qh++];
      const float xi = roi[4 * i], yi = roi[4 * i + 1], zi = roi[4 * i + 2];
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dz = -1; dz <= 1; ++dz) {
            const int64_t cx = cc[3 * i] + dx, cy = cc[3 * i + 1] + dy, cz = cc[3 * i + 2] + dz;
            if (cx < 0 || cy < 0 || cz < 0) continue;
            const uint64_t key = cell_key(cx, cy, cz);
            for (int32_t e = lower_bound_cell(ents, m, key); e < m && ents[e].key == key; ++e) {
              const int32_t j = ents[e].idx;
              if (label[j] >= 0) continue;
              const float ddx = roi[4 * j] - xi, ddy = roi[4 * j + 1] - yi,
                          ddz = roi[4 * j + 2] - zi;
              float d2 = ddx * ddx;
              d2 = d2 + ddy * ddy;
              d2 = d2 + ddz * ddz;
              if (d2 < tol2) {
                label[j] = s;
                queue[qt++] = j;
              }
            }
          }
    }
    if (qt >= p->cluster_min && qt <= p->cluster_max) {
      cl_id[n_clusters] = s;
      cl_size[n_clusters] = qt;
      ++n_clusters;
    }
  }

  if (labels_out) memcpy(labels_out, label, sizeof(int32_t) * (size_t)m);

  /* nearestKSearch(click, 1): exact 1-NN, float squared distance, ties -> lowest index */
  int32_t nn = 0;
  float best = FLT_MAX;
  for (int32_t i = 0; i < m; ++i) {
    const float ddx = roi[4 * i] - click[0], ddy = roi[4 * i + 1] - click[1],
                ddz = roi[4 * i + 2] - click[2];
    float d2 = ddx * ddx;
    d2 = d2 + ddy * ddy;
    d2 = d2 + ddz * ddz;
    if (d2 < best) {
      best = d2;
      nn = i;
    }
  }

  int32_t chosen = -1;
  if (n_clusters > 0) {
    /* largest first; ties by smallest seed (stable) */
    int32_t largest = 0;
    for (int32_t c = 1; c < n_clusters; ++c)
      if (cl_size[c] > cl_size[largest]) largest = c;
    chosen = cl_id[largest]; /* plane_index = 0 default */
    for (int32_t c = 0; c < n_clusters; ++c)
      if (cl_id[c] == label[nn]) {
        chosen = cl_id[c]; /* cluster containing the NN of the click */
        if (found_out) *found_out = 1;
      }
  }

  int32_t k = 0;
  if (chosen >= 0)
    for (int32_t i = 0; i < m; ++i)
      if (label[i] == chosen) out_idx[k++] = i;

  free(ents);
  free(cc);
  free(label);
  free(queue);
  free(cl_id);
  free(cl_size);
  return k;
}

/* ------------------------------------------------------------------------- */
/* 3x3 symmetric eigen solver (cyclic Jacobi, double), ascending eigenvalues  */
/* ------------------------------------------------------------------------- */
/* stands in for Eigen::SelfAdjointEigenSolver (LidarCornersEst.cpp:337-339) and
 * pcl::eigen33 (inside SACSegmentation's optimizeModelCoefficients). Eigenvector signs
 * are arbitrary there; callers apply their own sign convention. v[c] = c-th eigenvector. */
static void eig3_sym(const double a_in[9], double w[3], double v[3][3]) {
  double a[3][3], q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = a_in[3 * i + j];
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int pp = 0; pp < 2; ++pp)
      for (int qq = pp + 1; qq < 3; ++qq) {
        if (a[pp][qq] == 0.0) continue;
        const double theta = (a[qq][qq] - a[pp][pp]) / (2.0 * a[pp][qq]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* A <- A * G */
          const double akp = a[k][pp], akq = a[k][qq];
          a[k][pp] = c * akp - s * akq;
          a[k][qq] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- G^T * A */
          const double apk = a[pp][k], aqk = a[qq][k];
          a[pp][k] = c * apk - s * aqk;
          a[qq][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double qkp = q[k][pp], qkq = q[k][qq];
          q[k][pp] = c * qkp - s * qkq;
          q[k][qq] = s * qkp + c * qkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (d[order[j]] > d[order[j + 1]]) {
        int t = order[j];
        order[j] = order[j + 1];
        order[j + 1] = t;
      }
  for (int c = 0; c < 3; ++c) {
    w[c] = d[order[c]];
    double nrm = 0;
    for (int k = 0; k < 3; ++k) nrm += q[k][order[c]] * q[k][order[c]];
    nrm = sqrt(nrm);
    for (int k = 0; k < 3; ++k) v[c][k] = q[k][order[c]] / nrm;
  }
}

/* ------------------------------------------------------------------------- */
/* a3  getPlane (LidarCornersEst.cpp:190-221)                                  */
/* ------------------------------------------------------------------------- */
/* pcl::SACSegmentation, SACMODEL_PLANE, SAC_RANSAC, threshold 0.03, optimize coefficients:
 *   hypotheses from 3 sampled points, inlier iff |n.p + d| < thr (strict, float),
 *   best = most inliers; refit = PCA plane of the inliers; inliers re-selected with the
 *   refit plane (SACSegmentation::segment).
 * PCL's sampler (boost mt19937, seed 12345) cannot be reproduced without PCL: the samples come from a
 * counter-based hash sampler (same function in the HIP kernel).  The ITERATION RULE is PCL's
 * (pcl::RandomSampleConsensus::computeModel, PCL 1.8 ransac.hpp): k starts at 1; a hypothesis with more
 * inliers than the best so far (strict: ties keep the earlier one) becomes the model and sets
 * k = log(1 - probability) / log(1 - w^3), w = its inlier share, 1 - w^3 clamped to [eps, 1 - eps]; a
 * degenerate sample is skipped without counting (at most 10 x max_iterations of them); the loop ends when
 * iterations >= k or iterations > max_iterations.  On a board cluster (w ~ 0.9) that is 3-5 hypotheses.
 * ransac_probability <= 0 keeps rounds 1-4's fixed number of hypotheses (ransac_hyp, no early stop). */
static uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
static uint32_t sample_index(uint32_t seed, uint32_t h, uint32_t k, uint32_t m) {
  const uint32_t r = hash_u32(seed ^ hash_u32(h * 3u + k + 0x9E3779B9u));
  return (uint32_t)(((uint64_t)r * (uint64_t)m) >> 32);
}

static int plane_from_3(const float* pts, uint32_t i0, uint32_t i1, uint32_t i2, float pl[4]) {
  if (i0 == i1 || i0 == i2 || i1 == i2) return 0;
  const float* p0 = pts + 4 * i0;
  const float* p1 = pts + 4 * i1;
  const float* p2 = pts + 4 * i2;
  const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
  const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
  float nx = ay * bz - az * by;
  float ny = az * bx - ax * bz;
  float nz = ax * by - ay * bx;
  float n2 = nx * nx;
  n2 = n2 + ny * ny;
  n2 = n2 + nz * nz;
  if (!(n2 > 1e-12f)) return 0; /* collinear / duplicate sample */
  const float nrm = sqrtf(n2);
  nx = nx / nrm;
  ny = ny / nrm;
  nz = nz / nrm;
  float d = nx * p0[0];
  d = d + ny * p0[1];
  d = d + nz * p0[2];
  pl[0] = nx;
  pl[1] = ny;
  pl[2] = nz;
  pl[3] = -d;
  return 1;
}

static inline float plane_dist(const float pl[4], const float* q) {
  float s = pl[0] * q[0];
  s = s + pl[1] * q[1];
  s = s + pl[2] * q[2];
  s = s + pl[3];
  return fabsf(s);
}

int32_t orc_ransac_plane(const float* pts, int32_t m, const orc_params* p, int32_t* out_idx,
                         float plane_out[4]) {
  if (m < 3) return 0;
  const float thr = (float)p->ransac_thresh;
  int32_t best_cnt = 0;
  float best_pl[4] = {0, 0, 0, 0};
  const int adaptive = p->ransac_probability > 0.0;
  const double log_probability = adaptive ? log(1.0 - p->ransac_probability) : 0.0;
  const double one_over_indices = 1.0 / (double)m;
  double k_iter = 1.0;
  int32_t iterations = 0;
  uint32_t skipped = 0;
  const uint32_t max_skip = (uint32_t)p->ransac_hyp * 10u;
  for (uint32_t h = 0;; ++h) {   /* h: samples drawn; iterations: the ones that gave a model */
    if (adaptive ? !((double)iterations < k_iter && skipped < max_skip) : !(h < (uint32_t)p->ransac_hyp)) break;
    float pl[4];
    const uint32_t i0 = sample_index(p->ransac_seed, h, 0, (uint32_t)m);
    const uint32_t i1 = sample_index(p->ransac_seed, h, 1, (uint32_t)m);
    const uint32_t i2 = sample_index(p->ransac_seed, h, 2, (uint32_t)m);
    if (!plane_from_3(pts, i0, i1, i2, pl)) {
      ++skipped;
      continue;
    }
    int32_t cnt = 0;
    for (int32_t i = 0; i < m; ++i)
      if (plane_dist(pl, pts + 4 * i) < thr) ++cnt;
    if (cnt > best_cnt) {
      best_cnt = cnt;
      memcpy(best_pl, pl, sizeof(pl));
      if (adaptive) {
        const double w = (double)best_cnt * one_over_indices;
        double p_no_outliers = 1.0 - w * w * w; /* (PCL: pow(w, 3)) */
        if (p_no_outliers < DBL_EPSILON) p_no_outliers = DBL_EPSILON;
        if (p_no_outliers > 1.0 - DBL_EPSILON) p_no_outliers = 1.0 - DBL_EPSILON;
        k_iter = log_probability / log(p_no_outliers);
      }
    }
    ++iterations;
    if (adaptive && iterations > p->ransac_hyp) break;
  }
  if (best_cnt == 0) return 0;

  /* optimizeModelCoefficients: needs > 3 inliers, else keeps the sample plane */
  float pl[4];
  memcpy(pl, best_pl, sizeof(pl));
  if (best_cnt > 3) {
    double c[3] = {0, 0, 0};
    for (int32_t i = 0; i < m; ++i)
      if (plane_dist(best_pl, pts + 4 * i) < thr)
        for (int a = 0; a < 3; ++a) c[a] += pts[4 * i + a];
    for (int a = 0; a < 3; ++a) c[a] /= best_cnt;
    double cov[9] = {0};
    for (int32_t i = 0; i < m; ++i)
      if (plane_dist(best_pl, pts + 4 * i) < thr) {
        const double dx = pts[4 * i] - c[0], dy = pts[4 * i + 1] - c[1], dz = pts[4 * i + 2] - c[2];
        cov[0] += dx * dx;
        cov[1] += dx * dy;
        cov[2] += dx * dz;
        cov[4] += dy * dy;
        cov[5] += dy * dz;
        cov[8] += dz * dz;
      }
    cov[3] = cov[1];
    cov[6] = cov[2];
    cov[7] = cov[5];
    for (int k = 0; k < 9; ++k) cov[k] /= best_cnt;
    double w[3], v[3][3];
    eig3_sym(cov, w, v);
    /* smallest eigenvector = normal; sign: keep the sample plane's orientation */
    double n[3] = {v[0][0], v[0][1], v[0][2]};
    if (n[0] * best_pl[0] + n[1] * best_pl[1] + n[2] * best_pl[2] < 0)
      for (int a = 0; a < 3; ++a) n[a] = -n[a];
    pl[0] = (float)n[0];
    pl[1] = (float)n[1];
    pl[2] = (float)n[2];
    pl[3] = (float)(-(n[0] * c[0] + n[1] * c[1] + n[2] * c[2]));
  }
  int32_t k = 0;
  for (int32_t i = 0; i < m; ++i)
    if (plane_dist(pl, pts + 4 * i) < thr) out_idx[k++] = i;
  if (plane_out) memcpy(plane_out, pl, sizeof(pl));
  return k;
}

/* ------------------------------------------------------------------------- */
/* a4  transformbyPCA (LidarCornersEst.cpp:330-364)                            */
/* ------------------------------------------------------------------------- */
/* compute3DCentroid, computeCovarianceMatrixNormalized (/N), SelfAdjointEigenSolver
 * (ascending), col(2) = col(0) x col(1), T = [E^T | -E^T c], transformPointCloud.
 * Eigen leaves eigenvector signs unspecified; convention used here AND in the HIP path:
 *   e0 (normal) points from the board towards the sensor origin (e0 . c < 0),
 *   e1's largest-magnitude component is positive (ties -> lowest axis). */
static void plane_frame_signs(double e0[3], double e1[3], const double c[3]) {
  if (e0[0] * c[0] + e0[1] * c[1] + e0[2] * c[2] > 0)
    for (int a = 0; a < 3; ++a) e0[a] = -e0[a];
  int big = 0;
  for (int a = 1; a < 3; ++a)
    if (fabs(e1[a]) > fabs(e1[big])) big = a;
  if (e1[big] < 0)
    for (int a = 0; a < 3; ++a) e1[a] = -e1[a];
}

int32_t orc_plane_frame(const float* pts, int32_t m, const orc_params* p, float pca[16],
                        float* pts_pca) {
  if (m < 3) return ORC_TOO_FEW_POINTS;
  double c[3], cov[9];
  if (p->accum_float) {
    /* pcl::compute3DCentroid / computeCovarianceMatrix accumulate in float for PointXYZI */
    float cf[3] = {0, 0, 0};
    for (int32_t i = 0; i < m; ++i)
      for (int a = 0; a < 3; ++a) cf[a] = cf[a] + pts[4 * i + a];
    for (int a = 0; a < 3; ++a) cf[a] = cf[a] / (float)m;
    float cv[6] = {0, 0, 0, 0, 0, 0};
    for (int32_t i = 0; i < m; ++i) {
      const float dx = pts[4 * i] - cf[0], dy = pts[4 * i + 1] - cf[1], dz = pts[4 * i + 2] - cf[2];
      cv[0] = cv[0] + dx * dx;
      cv[1] = cv[1] + dx * dy;
      cv[2] = cv[2] + dx * dz;
      cv[3] = cv[3] + dy * dy;
      cv[4] = cv[4] + dy * dz;
      cv[5] = cv[5] + dz * dz;
    }
    for (int k = 0; k < 6; ++k) cv[k] = cv[k] / (float)m;
    for (int a = 0; a < 3; ++a) c[a] = cf[a];
    cov[0] = cv[0];
    cov[1] = cov[3] = cv[1];
    cov[2] = cov[6] = cv[2];
    cov[4] = cv[3];
    cov[5] = cov[7] = cv[4];
    cov[8] = cv[5];
  } else {
    c[0] = c[1] = c[2] = 0;
    for (int32_t i = 0; i < m; ++i)
      for (int a = 0; a < 3; ++a) c[a] += pts[4 * i + a];
    for (int a = 0; a < 3; ++a) c[a] /= m;
    /* the product narrows the centroid to float before subtracting, as PCL's Vector4f does */
    for (int a = 0; a < 3; ++a) c[a] = (double)(float)c[a];
    memset(cov, 0, sizeof(cov));
    for (int32_t i = 0; i < m; ++i) {
      const double dx = pts[4 * i] - c[0], dy = pts[4 * i + 1] - c[1], dz = pts[4 * i + 2] - c[2];
      cov[0] += dx * dx;
      cov[1] += dx * dy;
      cov[2] += dx * dz;
      cov[4] += dy * dy;
      cov[5] += dy * dz;
      cov[8] += dz * dz;
    }
    cov[3] = cov[1];
    cov[6] = cov[2];
    cov[7] = cov[5];
    for (int k = 0; k < 9; ++k) cov[k] /= m;
  }
  double w[3], v[3][3];
  eig3_sym(cov, w, v);
  double e0[3] = {v[0][0], v[0][1], v[0][2]};
  double e1[3] = {v[1][0], v[1][1], v[1][2]};
  plane_frame_signs(e0, e1, c);
  /* Matrix3f eigenvectors: narrow to float, then col(2) = col(0).cross(col(1)) in float (:343) */
  float f0[3], f1[3], f2[3];
  for (int a = 0; a < 3; ++a) {
    f0[a] = (float)e0[a];
    f1[a] = (float)e1[a];
  }
  f2[0] = f0[1] * f1[2] - f0[2] * f1[1];
  f2[1] = f0[2] * f1[0] - f0[0] * f1[2];
  f2[2] = f0[0] * f1[1] - f0[1] * f1[0];
  const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
  const float* rows[3] = {f0, f1, f2};
  for (int r = 0; r < 3; ++r) {
    for (int a = 0; a < 3; ++a) pca[4 * r + a] = rows[r][a];
    float t = rows[r][0] * cf[0];
    t = t + rows[r][1] * cf[1];
    t = t + rows[r][2] * cf[2];
    pca[4 * r + 3] = -1.0f * t; /* :349 */
  }
  pca[12] = pca[13] = pca[14] = 0.0f;
  pca[15] = 1.0f;
  if (pts_pca)
    for (int32_t i = 0; i < m; ++i) {
      const float x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
      for (int r = 0; r < 3; ++r) {
        float s = pca[4 * r] * x;
        s = s + pca[4 * r + 1] * y;
        s = s + pca[4 * r + 2] * z;
        s = s + pca[4 * r + 3];
        pts_pca[4 * i + r] = s;
      }
      pts_pca[4 * i + 3] = pts[4 * i + 3];
    }
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a5  calHist + get_gray_zone (LidarCornersEst.cpp:224-328)                   */
/* ------------------------------------------------------------------------- */
static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

int32_t orc_gray_zone(const float* intensity, int32_t m, const orc_params* p, double rlrh[2],
                      double gray_zone[2]) {
  const int HL = p->hist_bins; /* HISTO_LENGTH :226 */
  if (m <= 0 || HL <= 0 || HL > 4096) return ORC_DEGENERATE_HIST;
  double* datas = (double*)malloc(sizeof(double) * (size_t)m);
  for (int32_t i = 0; i < m; ++i) datas[i] = intensity[i];
  qsort(datas, (size_t)m, sizeof(double), cmp_double); /* :232 */
  const double mn = datas[0], mx = datas[m - 1];
  if (!(mx > mn)) {
    free(datas);
    return ORC_DEGENERATE_HIST; /* reference divides by zero here */
  }
  const double factor = HL / (mx - mn); /* :235 */
  int* hist = (int*)calloc((size_t)HL + 1, sizeof(int));
  for (int32_t i = 0; i < m; ++i) {
    const double sample = datas[i] - mn;
    const int bin = (int)round(sample * factor); /* :239 ; == HL for the maximum -> UB write in the
                                                    reference; counted in a spare slot and ignored */
    hist[bin < 0 ? 0 : (bin > HL ? HL : bin)]++;
  }
  double sum = 0.0;
  for (int32_t i = 0; i < m; ++i) sum += datas[i]; /* ascending order, :245-247 */
  const double mean = sum / m;
  free(datas);
  const double bin_width = (mx - mn) / HL; /* :258 */

  /* std::map<double,int> keyed by the bin COUNT: insert() keeps the first (lowest) bin for a
   * repeated count (:261-263); walked from the largest count downwards (:264-282). */
  int low_found = 0, high_found = 0;
  double low = -1, high = -1;
  int prev_count = INT32_MAX;
  for (;;) {
    /* next distinct count below prev_count */
    int best = -1;
    for (int i = 0; i < HL; ++i)
      if (hist[i] < prev_count && hist[i] > best) best = hist[i];
    if (best < 0) break; /* map exhausted: reference runs past rend() -> UB */
    int index = 0;
    while (hist[index] != best) ++index; /* lowest bin with that count */
    const double bin_edge = bin_width * (double)index + mn; /* :269 */
    if (bin_edge > mean && !high_found) {
      high_found = 1;
      high = bin_edge;
    }
    if (bin_edge < mean && !low_found) {
      low_found = 1;
      low = bin_edge;
    }
    if (low_found && high_found) break;
    prev_count = best;
  }
  free(hist);
  if (!low_found || !high_found) return ORC_DEGENERATE_HIST;
  rlrh[0] = low;
  rlrh[1] = high;
  const double rate = p->gray_rate;
  gray_zone[0] = ((rate - 1) * low + high) / rate; /* :322 */
  gray_zone[1] = (low + (rate - 1) * high) / rate; /* :323 */
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a6  VirtualboardError::operator() (include/ilcc2/Optimization.h:31-107)     */
/* ------------------------------------------------------------------------- */
/* Jet semantics restated: floor/ceil drop derivatives, comparisons act on the scalar part,
 * abs(f) = f.a < 0 ? -f : f, matched cells and disabled OOB return the constant 0. */
/* c, s = cos / sin of theta_t[0]: callers that visit many points under one theta hand them in (the values are
 * the same doubles either way) */
static double residual_cs(double c, double s, const double theta_t[3], double y, double z, int32_t board_w,
                          int32_t board_h, double g, int32_t topleft_white, int32_t laser_white, int32_t use_oob,
                          double jac[3]) {
  /* ceres::AngleAxisRotatePoint with axis (theta,0,0) applied to (0,y,z) == Rx(theta) (:37-40) */
  const double ry = c * y - s * z;
  const double rz = s * y + c * z;
  const double r1 = ry + theta_t[1]; /* :41 */
  const double r2 = rz + theta_t[2]; /* :42 */
  const double W = (double)board_w, H = (double)board_h;
  const double i = (r1 + W * g / 2.0) / g; /* :45 */
  const double j = (r2 + H * g / 2.0) / g; /* :46 */
  /* d i / d(theta,ty,tz), d j / d(...) */
  const double di[3] = {(-s * y - c * z) / g, 1.0 / g, 0.0};
  const double dj[3] = {(c * y - s * z) / g, 0.0, 1.0 / g};
  double si = 0, sj = 0, res = 0;
  if (i > 0 && i < W && j > 0 && j < H) { /* :48-49 strict */
    const double ifl = floor(i), jfl = floor(j);
    const double ii = floor(ifl / 2.0) * 2.0, jj = floor(jfl / 2.0) * 2.0;
    int white = !topleft_white;
    if (ifl == ii && jfl == jj) white = topleft_white; /* both even :57-58 */
    if (ifl != ii && jfl != jj) white = topleft_white; /* both odd  :59-60 */
    if ((laser_white != 0) == (white != 0)) {
      res = 0; /* :64-65 */
    } else {
      double ie, je;
      if (i - ifl > 0.5) { ie = ceil(i) - i; si = -1; } else { ie = i - ifl; si = 1; } /* :70-73 */
      if (j - jfl > 0.5) { je = ceil(j) - j; sj = -1; } else { je = j - jfl; sj = 1; } /* :75-78 */
      res = ie + je; /* :80 */
    }
  } else if (use_oob) { /* :85-101 */
    double ie, je;
    if (fabs(i) < fabs(i - W)) { ie = fabs(i); si = (i < 0) ? -1 : 1; }
    else { ie = fabs(i - W); si = (i - W < 0) ? -1 : 1; }
    if (fabs(j) < fabs(j - H)) { je = fabs(j); sj = (j < 0) ? -1 : 1; }
    else { je = fabs(j - H); sj = (j - H < 0) ? -1 : 1; }
    res = ie + je;
  } else {
    res = 0; /* :102-104 */
  }
  if (jac)
    for (int k = 0; k < 3; ++k) jac[k] = si * di[k] + sj * dj[k];
  return res;
}

double orc_residual(const double theta_t[3], double y, double z, int32_t board_w, int32_t board_h,
                    double g, int32_t topleft_white, int32_t laser_white, int32_t use_oob,
                    double jac[3]) {
  const double th = theta_t[0];
  return residual_cs(cos(th), sin(th), theta_t, y, z, board_w, board_h, g, topleft_white, laser_white, use_oob, jac);
}

/* HuberLoss(a): rho(s) = s (s <= a^2) else 2 a sqrt(s) - a^2 ; rho' = 1 or a/sqrt(s) */
static inline void huber(double a, double s, double* rho0, double* rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    *rho0 = 2.0 * a * r - b;
    *rho1 = a / r;
    if (*rho1 < DBL_MIN) *rho1 = DBL_MIN;
  } else {
    *rho0 = s;
    *rho1 = 1.0;
  }
}

double orc_cost(const double theta_t[3], const float* y, const float* z, const int8_t* label,
                int32_t m, const orc_params* p, int32_t topleft_white, int32_t use_oob) {
  double cost = 0;
  const double c = cos(theta_t[0]), s = sin(theta_t[0]);
  for (int32_t k = 0; k < m; ++k) {
    if (label[k] != 0 && label[k] != 1) continue;
    const double r = residual_cs(c, s, theta_t, (double)y[k], (double)z[k], p->board_w, p->board_h,
                                 p->grid_length, topleft_white, label[k], use_oob, NULL);
    double r0, r1;
    huber(p->huber_delta, r * r, &r0, &r1);
    cost += 0.5 * r0;
  }
  return cost;
}

/* Fixed-point form of the same sum (ORC_SOLVER_GRID): every term is rounded to a multiple of 2^-40 and the
 * terms are added as integers, so the total does not depend on the order of summation. */
/* One point's term of the fixed-point cost.  The residual is VirtualboardError's (Optimization.h:31-107, the
 * logic of residual_cs above, out-of-board branch included) with two changes of ARITHMETIC, not of meaning, made
 * so that a GPU lane can evaluate a 3 x 3 stencil of translations from shared per-axis terms without fp64
 * divisions or square roots: the grid coordinate is scaled by the reciprocal 1/g computed once (the functor
 * divides by g: <= 1 ulp apart), and Huber's rho is taken on r directly -- r > delta ? 2 delta r - delta^2 : r^2 --
 * instead of through sqrt(r^2).  Every operation below is a plain IEEE double operation (no fused multiply-add),
 * in this order, on both sides. */
typedef struct {
  double in_dist;  /* min(frac, 1 - frac), :70-78 */
  double out_dist; /* min(|v|, |v - n|), :86-97 */
  int inside;      /* 0 < v < n, strict (:48-49) */
  int odd;         /* floor(v) odd */
} axis_terms_t;

static inline axis_terms_t axis_terms(double v, double n) {
  axis_terms_t t;
  t.inside = v > 0 && v < n;
  const double fl = floor(v);
  t.odd = fl != floor(fl / 2.0) * 2.0;
  const double fr = v - fl;
  t.in_dist = (fr > 0.5) ? (fl + 1.0) - v : fr;
  t.out_dist = (fabs(v) < fabs(v - n)) ? fabs(v) : fabs(v - n);
  return t;
}

static inline int64_t term_q(double c, double s, const double theta_t[3], float yf, float zf, int8_t label,
                             const orc_params* p, int32_t topleft_white, int32_t use_oob) {
  const double y = (double)yf, z = (double)zf, g = p->grid_length;
  const double W = (double)p->board_w, H = (double)p->board_h;
  const double inv_g = 1.0 / g;
  const double ry = c * y - s * z;
  const double rz = s * y + c * z;
  const axis_terms_t ai = axis_terms(((ry + theta_t[1]) + W * g / 2.0) * inv_g, W);
  const axis_terms_t aj = axis_terms(((rz + theta_t[2]) + H * g / 2.0) * inv_g, H);
  double res = 0.0;
  if (ai.inside && aj.inside) {
    const int white = (ai.odd == aj.odd) ? (topleft_white != 0) : !(topleft_white != 0); /* :53-61 */
    if ((label != 0) != white) res = ai.in_dist + aj.in_dist;                             /* :64-82 */
  } else if (use_oob) {
    res = ai.out_dist + aj.out_dist;                                                      /* :85-101 */
  }
  const double d = p->huber_delta;
  const double r0 = (res > d) ? 2.0 * d * res - d * d : res * res;
  return (int64_t)rint(r0 * (0.5 * ORC_COST_Q_ONE)); /* 1/2 rho, in units of 2^-40 */
}

int64_t orc_cost_q(const double theta_t[3], const float* y, const float* z, const int8_t* label,
                   int32_t m, const orc_params* p, int32_t topleft_white, int32_t use_oob) {
  int64_t cost = 0;
  const double c = cos(theta_t[0]), s = sin(theta_t[0]);
  for (int32_t k = 0; k < m; ++k) {
    if (label[k] != 0 && label[k] != 1) continue;
    cost += term_q(c, s, theta_t, y[k], z[k], label[k], p, topleft_white, use_oob);
  }
  return cost;
}

/* The same sum, abandoned (INT64_MAX returned) once it exceeds `limit`: the terms are non-negative integers, so
 * a partial sum above the best complete sum cannot win -- and cannot tie.  The points are visited with a stride
 * coprime to m so that every prefix samples the whole board (integer sums do not care about the order).  Only
 * an exact shortcut for orc_grid_search: the answer is the exhaustive one. */
static int64_t cost_q_bounded(const double theta_t[3], double c, double s, const float* y, const float* z,
                              const int8_t* label, int32_t m, const orc_params* p, int32_t topleft_white,
                              int32_t use_oob, int64_t limit, int32_t stride) {
  int64_t cost = 0;
  int32_t k = 0;
  for (int32_t n = 0; n < m; ++n) {
    if (label[k] == 0 || label[k] == 1) cost += term_q(c, s, theta_t, y[k], z[k], label[k], p, topleft_white, use_oob);
    k += stride;
    if (k >= m) k -= m;
    if ((n & 31) == 31 && cost > limit) return INT64_MAX;
  }
  return cost;
}

/* ------------------------------------------------------------------------- */
/* a7  Optimization::get_theta_t (src/Optimization.cpp:94-160)                 */
/* ------------------------------------------------------------------------- */
/* Ceres is not in the tree (find_package(Ceres REQUIRED), no version).  This restates the
 * published Ceres 1.14 TrustRegionMinimizer + DoglegStrategy(SUBSPACE_DOGLEG) +
 * DENSE_NORMAL_CHOLESKY for a 3-parameter problem with the option values the reference
 * leaves at their defaults: max_num_iterations 50, function_tolerance 1e-6,
 * gradient_tolerance 1e-10, parameter_tolerance 1e-8, initial_trust_region_radius 1e4,
 * max_trust_region_radius 1e16, min_trust_region_radius 1e-32, min_relative_decrease 1e-3,
 * min_lm_diagonal 1e-6, max_lm_diagonal 1e32, jacobi_scaling on, monotonic steps,
 * max_num_consecutive_invalid_steps 5; dogleg: mu 1e-8..1 (x10), thresholds .25/.75.
 * Robust loss enters through Ceres' Corrector (rho'' <= 0 for Huber, so residual and
 * Jacobian rows are scaled by sqrt(rho')). */

#define ORC_MAXP 6 /* parameters: 3 (board theta, ty, tz) or 6 (angle-axis + translation) */

typedef struct {
  int32_t kind;    /* 0: VirtualboardError blocks of the path;  1: Pose3d2dError blocks (orc_solve_pose_3d2d) */
  int32_t n;       /* labelled points / 3-D-2-D pairs */
  const float* y;  /* compacted */
  const float* z;
  const int8_t* lab;
  const orc_params* p;
  int32_t tlw, oob;
  const double* p3; /* kind 1 */
  const double* p2;
  double cam[4];    /* fx cx fy cy */
} lsq_problem;

static inline int lsq_np(const lsq_problem* q) { return q->kind ? 6 : 3; }
static inline int32_t lsq_nres(const lsq_problem* q) { return q->kind ? 2 * q->n : q->n; }

static double pose_eval(const lsq_problem* q, const double* x, double* r, double* J);

/* cost and (optionally) corrected residuals r[n] and jacobian J[n*3] at x */
static double lsq_eval(const lsq_problem* q, const double* x, double* r, double* J) {
  if (q->kind) return pose_eval(q, x, r, J);
  double cost = 0;
  const double cth = cos(x[0]), sth = sin(x[0]);
  for (int32_t k = 0; k < q->n; ++k) {
    double jac[3];
    const double res = residual_cs(cth, sth, x, (double)q->y[k], (double)q->z[k], q->p->board_w,
                                   q->p->board_h, q->p->grid_length, q->tlw, q->lab[k], q->oob,
                                   J ? jac : NULL);
    double r0, r1;
    huber(q->p->huber_delta, res * res, &r0, &r1);
    cost += 0.5 * r0;
    if (r) {
      const double sr = sqrt(r1);
      r[k] = sr * res;
      if (J)
        for (int c = 0; c < 3; ++c) J[3 * k + c] = sr * jac[c];
    }
  }
  return cost;
}

/* dense Cholesky solve (lower), returns 0 on non-positive pivot (Eigen LLT NumericalIssue) */
static int chol_solve(int np, const double* A, const double* b, double* x) {
  double L[ORC_MAXP][ORC_MAXP] = {{0}};
  for (int i = 0; i < np; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[np * i + j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return 0;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double yv[ORC_MAXP];
  for (int i = 0; i < np; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * yv[k];
    yv[i] = s / L[i][i];
  }
  for (int i = np - 1; i >= 0; --i) {
    double s = yv[i];
    for (int k = i + 1; k < np; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  for (int i = 0; i < np; ++i)
    if (!isfinite(x[i])) return 0;
  return 1;
}

/* minimise 0.5 y'By + g'y on |y| = radius (2-D, B symmetric positive semi-definite).
 * Ceres solves a quartic (DoglegStrategy::FindMinimumOnTrustRegionBoundary); the minimiser is
 * the same point whichever exact method finds it.  Here: eigen-decomposition of B and Newton
 * on the secular equation 1/|y(lam)| = 1/radius (More-Sorensen), y(lam) = -(B + lam I)^-1 g.
 * The HIP kernel uses the same routine. */
static void min_on_circle(const double B[4], const double g[2], double radius, double y[2]) {
  const double d = 0.5 * (B[0] - B[3]), e = B[1];
  const double h = sqrt(d * d + e * e), mean = 0.5 * (B[0] + B[3]);
  const double l1 = mean - h, l2 = mean + h;
  /* unit eigenvector of l2 */
  double v2x, v2y;
  if (h == 0.0) {
    v2x = 1.0;
    v2y = 0.0;
  } else if (d >= 0.0) {
    v2x = d + h;
    v2y = e;
  } else {
    v2x = e;
    v2y = h - d;
  }
  {
    const double nv = sqrt(v2x * v2x + v2y * v2y);
    if (nv > 0.0) {
      v2x /= nv;
      v2y /= nv;
    } else {
      v2x = 1.0;
      v2y = 0.0;
    }
  }
  const double v1x = -v2y, v1y = v2x;
  const double g1 = v1x * g[0] + v1y * g[1], g2 = v2x * g[0] + v2y * g[1];
  const double gn = sqrt(g1 * g1 + g2 * g2);
  double lo = fmax(0.0, -l1);
  lo = fmax(lo, gn / radius - l2);
  const double hi = gn / radius - l1;
  double lam = lo;
  if (!(l1 + lam > 0.0)) lam = lo + 1e-12 * fmax(1.0, fabs(hi));
  for (int it = 0; it < 60; ++it) {
    const double a1 = l1 + lam, a2 = l2 + lam;
    const double y1 = -g1 / a1, y2 = -g2 / a2;
    const double ny = sqrt(y1 * y1 + y2 * y2);
    const double qq = g1 * g1 / (a1 * a1 * a1) + g2 * g2 / (a2 * a2 * a2);
    if (!(qq > 0.0) || !isfinite(ny)) break;
    const double dl = (ny * ny / qq) * ((ny - radius) / radius);
    double nl = lam + dl;
    if (!(l1 + nl > 0.0)) nl = 0.5 * (lam + fmax(0.0, -l1)); /* safeguard: stay right of the pole */
    if (fabs(nl - lam) <= 1e-15 * fmax(1.0, fabs(nl))) {
      lam = nl;
      break;
    }
    lam = nl;
  }
  {
    const double a1 = l1 + lam, a2 = l2 + lam;
    double y1 = (a1 > 0.0) ? -g1 / a1 : 0.0, y2 = (a2 > 0.0) ? -g2 / a2 : 0.0;
    double ny = sqrt(y1 * y1 + y2 * y2);
    if (ny < radius * (1.0 - 1e-9) && !(a1 > 1e-300 * fmax(1.0, l2))) {
      /* hard case: g has no component along the soft eigenvector; move along it to the boundary */
      y1 = sqrt(fmax(0.0, radius * radius - y2 * y2));
      ny = radius;
    }
    if (ny > 0.0) { /* land exactly on the circle */
      y1 *= radius / ny;
      y2 *= radius / ny;
    }
    y[0] = v1x * y1 + v2x * y2;
    y[1] = v1y * y1 + v2y * y2;
  }
}

typedef struct {
  int np;
  double radius, mu;
  int reuse;
  double diagonal[ORC_MAXP], gradient[ORC_MAXP], gn[ORC_MAXP];
  double alpha, step_norm;
  /* subspace model */
  int one_dim;
  double basis[ORC_MAXP][2], sg[2], sB[4];
} dogleg_state;

static inline double vnorm(const double* v, int np) {
  double s = 0;
  for (int c = 0; c < np; ++c) s += v[c] * v[c];
  return sqrt(s);
}

/* traditional dogleg step in scaled space (fallback only) */
static void dogleg_traditional(dogleg_state* s, double* step) {
  const int np = s->np;
  const double gnn = vnorm(s->gn, np);
  const double gn_ = vnorm(s->gradient, np);
  if (gnn <= s->radius) {
    for (int c = 0; c < np; ++c) step[c] = s->gn[c] / s->diagonal[c];
    s->step_norm = gnn;
    return;
  }
  if (gn_ * s->alpha >= s->radius) {
    for (int c = 0; c < np; ++c) step[c] = -(s->radius / gn_) * s->gradient[c] / s->diagonal[c];
    s->step_norm = s->radius;
    return;
  }
  /* intersect the segment Cauchy -> GN with the boundary */
  double bdota = 0, a2 = 0, bma2 = 0;
  for (int c = 0; c < np; ++c) {
    const double a = -s->alpha * s->gradient[c];
    bdota += a * s->gn[c];
    a2 += a * a;
    bma2 += (s->gn[c] - a) * (s->gn[c] - a);
  }
  const double cc = bdota - a2;
  const double d = sqrt(cc * cc + bma2 * (s->radius * s->radius - a2));
  const double beta = (cc <= 0) ? (d - cc) / bma2 : (s->radius * s->radius - a2) / (d + cc);
  for (int c = 0; c < np; ++c) {
    const double a = -s->alpha * s->gradient[c];
    step[c] = (a + beta * (s->gn[c] - a)) / s->diagonal[c];
  }
  s->step_norm = s->radius;
}

/* returns 0 on LINEAR_SOLVER_FAILURE; n = residual count, J is n x np row-major */
static int dogleg_compute_step(dogleg_state* s, int32_t n, const double* J, const double* r,
                               double* step) {
  const int np = s->np;
  if (!s->reuse) {
    s->reuse = 1;
    double JtJ[ORC_MAXP * ORC_MAXP] = {0}, Jtr[ORC_MAXP] = {0};
    for (int32_t k = 0; k < n; ++k)
      for (int a = 0; a < np; ++a) {
        Jtr[a] += J[np * k + a] * r[k];
        for (int b = 0; b < np; ++b) JtJ[np * a + b] += J[np * k + a] * J[np * k + b];
      }
    for (int c = 0; c < np; ++c) {
      double d = JtJ[(np + 1) * c];
      d = fmin(fmax(d, 1e-6), 1e32);
      s->diagonal[c] = sqrt(d);
      s->gradient[c] = Jtr[c] / s->diagonal[c];
    }
    /* Cauchy point: alpha = |g|^2 / |J D^-1 g|^2 */
    {
      double sgv[ORC_MAXP], num = 0, den = 0;
      for (int c = 0; c < np; ++c) {
        sgv[c] = s->gradient[c] / s->diagonal[c];
        num += s->gradient[c] * s->gradient[c];
      }
      for (int a = 0; a < np; ++a)
        for (int b = 0; b < np; ++b) den += sgv[a] * JtJ[np * a + b] * sgv[b];
      s->alpha = num / den;
    }
    /* Gauss-Newton step with regulariser mu * diag */
    int ok = 0;
    while (s->mu < 1.0) {
      double A[ORC_MAXP * ORC_MAXP];
      memcpy(A, JtJ, sizeof(A));
      for (int c = 0; c < np; ++c) {
        const double lm = s->diagonal[c] * sqrt(s->mu);
        A[(np + 1) * c] += lm * lm;
      }
      if (chol_solve(np, A, Jtr, s->gn)) {
        ok = 1;
        break;
      }
      s->mu *= 10.0;
    }
    if (!ok) return 0;
    for (int c = 0; c < np; ++c) s->gn[c] *= -s->diagonal[c];
    /* subspace model: orthonormal basis of span{gradient, gn} (Gram-Schmidt with pivoting,
     * standing in for Eigen::ColPivHouseholderQR; signs of the basis do not matter) */
    {
      double v0[ORC_MAXP], v1[ORC_MAXP];
      double n0 = 0, n1 = 0;
      for (int c = 0; c < np; ++c) {
        n0 += s->gradient[c] * s->gradient[c];
        n1 += s->gn[c] * s->gn[c];
      }
      const double* first = (n0 >= n1) ? s->gradient : s->gn;
      const double* second = (n0 >= n1) ? s->gn : s->gradient;
      const double nf = sqrt(fmax(n0, n1));
      for (int c = 0; c < np; ++c) v0[c] = first[c] / nf;
      double dot = 0;
      for (int c = 0; c < np; ++c) dot += second[c] * v0[c];
      double nr = 0;
      for (int c = 0; c < np; ++c) {
        v1[c] = second[c] - dot * v0[c];
        nr += v1[c] * v1[c];
      }
      nr = sqrt(nr);
      /* rank test as ColPivHouseholderQR: |R11| <= eps * n * |R00| -> rank 1 */
      s->one_dim = !(nr > (double)np * DBL_EPSILON * nf);
      if (!s->one_dim) {
        for (int c = 0; c < np; ++c) {
          v1[c] /= nr;
          s->basis[c][0] = v0[c];
          s->basis[c][1] = v1[c];
        }
        double u[2][ORC_MAXP];
        for (int c = 0; c < np; ++c) {
          u[0][c] = v0[c] / s->diagonal[c];
          u[1][c] = v1[c] / s->diagonal[c];
        }
        for (int a = 0; a < 2; ++a) {
          s->sg[a] = 0;
          for (int c = 0; c < np; ++c) s->sg[a] += s->basis[c][a] * s->gradient[c];
          for (int b = 0; b < 2; ++b) {
            double acc = 0;
            for (int c = 0; c < np; ++c)
              for (int d = 0; d < np; ++d) acc += u[a][c] * JtJ[np * c + d] * u[b][d];
            s->sB[2 * a + b] = acc;
          }
        }
      }
    }
  }
  /* ComputeSubspaceDoglegStep */
  const double gnn = vnorm(s->gn, np);
  if (gnn <= s->radius) {
    for (int c = 0; c < np; ++c) step[c] = s->gn[c] / s->diagonal[c];
    s->step_norm = gnn;
    return 1;
  }
  if (s->one_dim) {
    const double gn_ = vnorm(s->gradient, np);
    for (int c = 0; c < np; ++c) step[c] = -(s->radius / gn_) * s->gradient[c] / s->diagonal[c];
    s->step_norm = s->radius;
    return 1;
  }
  double y2[2];
  min_on_circle(s->sB, s->sg, s->radius, y2);
  if (!isfinite(y2[0]) || !isfinite(y2[1])) {
    dogleg_traditional(s, step);
    return 1;
  }
  for (int c = 0; c < np; ++c)
    step[c] = (s->basis[c][0] * y2[0] + s->basis[c][1] * y2[1]) / s->diagonal[c];
  s->step_norm = s->radius;
  return 1;
}

/* x has lsq_np(q) entries */
static int32_t trust_region_minimize(const lsq_problem* q, double* x, double* final_cost) {
  const int np = lsq_np(q);
  const int32_t n = lsq_nres(q);
  if (n == 0) {
    *final_cost = 0;
    return 0;
  }
  double* r = (double*)malloc(sizeof(double) * (size_t)n);
  double* J = (double*)malloc(sizeof(double) * (size_t)np * (size_t)n);
  double scale[ORC_MAXP];
  dogleg_state st;
  memset(&st, 0, sizeof(st));
  st.np = np;
  st.radius = 1e4;
  st.mu = 1e-8;
  st.reuse = 0;

  double x_cost = lsq_eval(q, x, r, J);
  double x_norm = vnorm(x, np);
  double grad[ORC_MAXP] = {0};
  for (int32_t k = 0; k < n; ++k)
    for (int c = 0; c < np; ++c) grad[c] += J[np * k + c] * r[k];
  /* jacobi scaling, computed once at iteration 0 */
  for (int c = 0; c < np; ++c) {
    double sq = 0;
    for (int32_t k = 0; k < n; ++k) sq += J[np * k + c] * J[np * k + c];
    scale[c] = 1.0 / (1.0 + sqrt(sq));
  }
  for (int32_t k = 0; k < n; ++k)
    for (int c = 0; c < np; ++c) J[np * k + c] *= scale[c];

  int32_t iter = 0;
  int invalid = 0;
  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (iter >= 50) break;
    double gmax = 0;
    for (int c = 0; c < np; ++c) gmax = fmax(gmax, fabs(grad[c]));
    if (gmax <= 1e-10) break;
    if (st.radius <= 1e-32) break;
    ++iter;

    double step[ORC_MAXP];
    int valid = dogleg_compute_step(&st, n, J, r, step);
    double model_cost_change = 0;
    if (valid) {
      for (int32_t k = 0; k < n; ++k) {
        double mr = 0;
        for (int c = 0; c < np; ++c) mr += J[np * k + c] * step[c];
        model_cost_change -= mr * (r[k] + mr / 2.0);
      }
      valid = model_cost_change > 0.0;
    }
    if (!valid) {
      if (++invalid >= 5) break;
      st.mu *= 10.0; /* StepIsInvalid */
      st.reuse = 0;
      continue;
    }
    invalid = 0;
    double cand[ORC_MAXP], delta[ORC_MAXP];
    for (int c = 0; c < np; ++c) {
      delta[c] = step[c] * scale[c];
      cand[c] = x[c] + delta[c];
    }
    const double cand_cost = lsq_eval(q, cand, NULL, NULL);
    /* ParameterToleranceReached */
    double dn = 0;
    for (int c = 0; c < np; ++c) dn += (x[c] - cand[c]) * (x[c] - cand[c]);
    const double step_norm = sqrt(dn);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) break;
    /* FunctionToleranceReached */
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * x_cost) break;
    const double rel = cost_change / model_cost_change;
    if (rel > 1e-3) {
      /* HandleSuccessfulStep */
      for (int c = 0; c < np; ++c) x[c] = cand[c];
      x_norm = vnorm(x, np);
      x_cost = lsq_eval(q, x, r, J);
      for (int c = 0; c < np; ++c) grad[c] = 0;
      for (int32_t k = 0; k < n; ++k)
        for (int c = 0; c < np; ++c) grad[c] += J[np * k + c] * r[k];
      for (int32_t k = 0; k < n; ++k)
        for (int c = 0; c < np; ++c) J[np * k + c] *= scale[c];
      /* DoglegStrategy::StepAccepted */
      if (rel < 0.25) st.radius *= 0.5;
      if (rel > 0.75) st.radius = fmax(st.radius, 3.0 * st.step_norm);
      if (st.radius > 1e16) st.radius = 1e16;
      st.mu = fmax(1e-8, 2.0 * st.mu / 10.0);
      st.reuse = 0;
    } else {
      st.radius *= 0.5; /* StepRejected */
      st.reuse = 1;
    }
  }
  *final_cost = x_cost;
  free(r);
  free(J);
  return iter;
}

/* ---- Pose3d2dError (include/ilcc2/Optimization.h:126-189) with HuberLoss(0.1) per 2-vector block
 * (src/Optimization.cpp:43-51): the 6-parameter problem the reference's calib_lidar_cam solves with
 * the SAME Ceres options as the path's board fit.  It exists in the oracle for one reason: the
 * reference ships inputs AND output of this solve (process_data/pointgrey*.txt ->
 * config/pointgrey.bin), which pins trust_region_minimize / dogleg / corrector code above against
 * a real Ceres run (tests/test_oracle_golden.py::test_solver_pinned_by_shipped_extrinsic). */
static void rodrigues9(const double* w, double R[9], double* b_out, double* c_out) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = sqrt(th2);
  double a, b, c;
  if (th2 > 1e-16) {
    a = sin(th) / th;
    b = (1.0 - cos(th)) / th2;
    c = (th - sin(th)) / (th2 * th);
  } else {
    a = 1.0;
    b = 0.5;
    c = 1.0 / 6.0;
  }
  const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double k2 = 0;
      for (int k = 0; k < 3; ++k) k2 += K[3 * i + k] * K[3 * k + j];
      R[3 * i + j] = a * K[3 * i + j] + b * k2 + (i == j ? 1.0 : 0.0);
    }
  if (b_out) *b_out = b;
  if (c_out) *c_out = c;
}

static double pose_eval(const lsq_problem* q, const double* x, double* r, double* J) {
  double R[9], b, c;
  rodrigues9(x, R, &b, &c);
  /* right Jacobian of SO(3): d(exp([w]x) X)/dw = -R [X]x Jr,  Jr = I - b K + c K^2 */
  double Jr[9];
  {
    const double K[9] = {0, -x[2], x[1], x[2], 0, -x[0], -x[1], x[0], 0};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double k2 = 0;
        for (int k = 0; k < 3; ++k) k2 += K[3 * i + k] * K[3 * k + j];
        Jr[3 * i + j] = -b * K[3 * i + j] + c * k2 + (i == j ? 1.0 : 0.0);
      }
  }
  const double fx = q->cam[0], cx = q->cam[1], fy = q->cam[2], cy = q->cam[3];
  double cost = 0;
  for (int32_t i = 0; i < q->n; ++i) {
    const double* X = q->p3 + 3 * i;
    double pc[3];
    for (int k = 0; k < 3; ++k) pc[k] = R[3 * k] * X[0] + R[3 * k + 1] * X[1] + R[3 * k + 2] * X[2] + x[3 + k];
    const double iz = 1.0 / pc[2];
    const double r0 = q->p2[2 * i] - (fx * pc[0] * iz + cx);
    const double r1 = q->p2[2 * i + 1] - (fy * pc[1] * iz + cy);
    double rho0, rho1;
    huber(0.1, r0 * r0 + r1 * r1, &rho0, &rho1);
    cost += 0.5 * rho0;
    if (!r) continue;
    const double sr = sqrt(rho1);
    r[2 * i] = sr * r0;
    r[2 * i + 1] = sr * r1;
    if (!J) continue;
    const double Xx[9] = {0, -X[2], X[1], X[2], 0, -X[0], -X[1], X[0], 0};
    double RX[9], Jp[9];
    for (int a = 0; a < 3; ++a)
      for (int d = 0; d < 3; ++d) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += R[3 * a + k] * Xx[3 * k + d];
        RX[3 * a + d] = s;
      }
    for (int a = 0; a < 3; ++a)
      for (int d = 0; d < 3; ++d) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += RX[3 * a + k] * Jr[3 * k + d];
        Jp[3 * a + d] = -s;
      }
    const double du[3] = {fx * iz, 0, -fx * pc[0] * iz * iz};
    const double dv[3] = {0, fy * iz, -fy * pc[1] * iz * iz};
    double* j0 = J + 12 * i;
    double* j1 = j0 + 6;
    for (int d = 0; d < 3; ++d) {
      j0[d] = -sr * (du[0] * Jp[d] + du[1] * Jp[3 + d] + du[2] * Jp[6 + d]);
      j1[d] = -sr * (dv[0] * Jp[d] + dv[1] * Jp[3 + d] + dv[2] * Jp[6 + d]);
      j0[3 + d] = -sr * du[d];
      j1[3 + d] = -sr * dv[d];
    }
  }
  return cost;
}

int32_t orc_solve_pose_3d2d(const double* pts3d, const double* pts2d, int32_t n, const double camera[4],
                            double r[3], double t[3], double* final_cost) {
  lsq_problem q;
  memset(&q, 0, sizeof(q));
  q.kind = 1;
  q.n = n;
  q.p3 = pts3d;
  q.p2 = pts2d;
  for (int c = 0; c < 4; ++c) q.cam[c] = camera[c];
  double x[6] = {r[0], r[1], r[2], t[0], t[1], t[2]};
  double fc = 0;
  const int32_t it = trust_region_minimize(&q, x, &fc);
  for (int c = 0; c < 3; ++c) {
    r[c] = x[c];
    t[c] = x[3 + c];
  }
  if (final_cost) *final_cost = fc;
  return it;
}

/* classification of Optimization.cpp:114-125 ; returns labelled count, fills compact arrays */
static int32_t classify(const float* pts_pca, int32_t m, const double gz[2], float* y, float* z,
                        int8_t* lab, int32_t counts[3]) {
  int32_t n = 0;
  counts[0] = counts[1] = counts[2] = 0;
  for (int32_t i = 0; i < m; ++i) {
    const float inten = pts_pca[4 * i + 3];
    int8_t l;
    if ((double)inten < gz[0]) {
      l = 0;
      counts[0]++;
    } else if ((double)inten > gz[1]) {
      l = 1;
      counts[2]++;
    } else {
      counts[1]++;
      continue;
    }
    y[n] = pts_pca[4 * i + 1]; /* laserPoint(temp.y, temp.z) :127 */
    z[n] = pts_pca[4 * i + 2];
    lab[n] = l;
    ++n;
  }
  return n;
}

int32_t orc_get_theta_t(const float* pts_pca, int32_t m, const double gray_zone[2],
                        const orc_params* p, int32_t topleft_white, int32_t use_oob,
                        double theta_t[3], double* cost) {
  float* y = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
  float* z = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
  int8_t* lab = (int8_t*)malloc((size_t)(m > 0 ? m : 1));
  int32_t counts[3];
  lsq_problem q;
  memset(&q, 0, sizeof(q));
  q.n = classify(pts_pca, m, gray_zone, y, z, lab, counts);
  q.y = y;
  q.z = z;
  q.lab = lab;
  q.p = p;
  q.tlw = topleft_white;
  q.oob = use_oob;
  double fc = 0;
  const int32_t it = trust_region_minimize(&q, theta_t, &fc);
  if (cost) *cost = fc;
  free(y);
  free(z);
  free(lab);
  return it;
}

/* ------------------------------------------------------------------------- */
/* exhaustive grid: specification of the GPU search (new design)   */
/* ------------------------------------------------------------------------- */
/* candidates (k,a,b,phase): theta = th_min + k*th_step etc.  Selection: smallest cost;
 * ties -> smallest squared index distance to the candidate nearest (0,0,0) (mirrors the
 * reference starting its solve at zero); ties -> smallest flat index (phase 0 =
 * topleftWhite false, the reference's first hypothesis, before phase 1). */
static inline int32_t near_zero_index(double vmin, double step, int32_t n) {
  long c = lround(-vmin / step);
  if (c < 0) c = 0;
  if (c > n - 1) c = n - 1;
  return (int32_t)c;
}

int32_t orc_grid_search(const float* y, const float* z, const int8_t* label, int32_t m,
                        const orc_params* p, int32_t use_oob, double* best_cost,
                        double* cost_out) {
  const int32_t c_th = near_zero_index(p->th_min, p->th_step, p->n_th);
  const int32_t c_ty = near_zero_index(p->ty_min, p->ty_step, p->n_ty);
  const int32_t c_tz = near_zero_index(p->tz_min, p->tz_step, p->n_tz);
  int64_t bc = INT64_MAX;
  int64_t bd = INT64_MAX;
  int32_t bi = -1;
  /* stride ~ 0.618 m, coprime to m (used only by the bounded shortcut) */
  int32_t stride = 1;
  if (m > 2) {
    stride = (int32_t)((double)m * 0.6180339) | 1;
    for (;;) {
      int32_t a = stride, b = m;
      while (b) { const int32_t t = a % b; a = b; b = t; }
      if (a == 1) break;
      stride += 2;
    }
    if (stride >= m) stride = 1;
  }
  /* A first sweep over every 4th candidate per axis only tightens `limit` (the cost of SOME complete candidate, so
   * nothing cheaper-or-equal is ever abandoned); the selection itself happens in the full sweep that follows. */
  int64_t limit = INT64_MAX;
  for (int32_t sweep = cost_out ? 1 : 0; sweep < 2; ++sweep) {
    const int32_t dec = sweep == 0 ? 4 : 1;
    for (int32_t k = (sweep == 0 ? c_th % dec : 0); k < p->n_th; k += dec) {
      const double cth = cos(p->th_min + k * p->th_step), sth = sin(p->th_min + k * p->th_step);
      for (int32_t a = (sweep == 0 ? c_ty % dec : 0); a < p->n_ty; a += dec)
        for (int32_t b = (sweep == 0 ? c_tz % dec : 0); b < p->n_tz; b += dec) {
          const double x[3] = {p->th_min + k * p->th_step, p->ty_min + a * p->ty_step,
                               p->tz_min + b * p->tz_step};
          const int64_t d2 = (int64_t)(k - c_th) * (k - c_th) + (int64_t)(a - c_ty) * (a - c_ty) +
                             (int64_t)(b - c_tz) * (b - c_tz);
          for (int32_t ph = 0; ph < 2; ++ph) {
            const int64_t c = cost_out ? orc_cost_q(x, y, z, label, m, p, ph, use_oob)
                                       : cost_q_bounded(x, cth, sth, y, z, label, m, p, ph, use_oob, limit, stride);
            if (c < limit) limit = c;
            if (sweep == 0) continue;
            const int32_t flat = ((k * p->n_ty + a) * p->n_tz + b) * 2 + ph;
            if (cost_out) cost_out[flat] = (double)c / ORC_COST_Q_ONE;
            if (c < bc || (c == bc && (d2 < bd || (d2 == bd && flat < bi)))) {
              bc = c;
              bd = d2;
              bi = flat;
            }
          }
        }
    }
  }
  if (best_cost) *best_cost = (double)bc / ORC_COST_Q_ONE;
  return bi;
}

/* ------------------------------------------------------------------------- */
/* ORC_SOLVER_GRID refinement: pattern search + neighbouring basins            */
/* ------------------------------------------------------------------------- */
/* Replaces, for the grid mode only, the reference's two local Ceres solves (LidarCornersEst.cpp:398-409):
 * restated faithfully those run into their 50-iteration caps without converging and can leave the with-OOB
 * cost HIGHER than where they started.  This search only ever moves to a strictly cheaper lattice point. */
void orc_lattice_point(const orc_params* p, const int32_t lat[3], double theta_t[3]) {
  const double div = (double)(p->refine_div > 0 ? p->refine_div : 1);
  theta_t[0] = p->th_min + (double)lat[0] * (p->th_step / div);
  theta_t[1] = p->ty_min + (double)lat[1] * (p->ty_step / div);
  theta_t[2] = p->tz_min + (double)lat[2] * (p->tz_step / div);
}

static int64_t lattice_cost(const float* y, const float* z, const int8_t* label, int32_t m, const orc_params* p,
                            const int32_t lat[3], int32_t phase) {
  double x[3];
  orc_lattice_point(p, lat, x);
  return orc_cost_q(x, y, z, label, m, p, phase, 1);
}

int64_t orc_pattern_refine(const float* y, const float* z, const int8_t* label, int32_t m, const orc_params* p,
                           int32_t lat[3], int32_t* phase, int64_t* alt_cost, int32_t* rounds, int32_t* hops) {
  return orc_pattern_refine2(y, z, label, m, p, lat, phase, alt_cost, rounds, hops, NULL);
}

int64_t orc_pattern_refine2(const float* y, const float* z, const int8_t* label, int32_t m, const orc_params* p, int32_t lat[3],
                            int32_t* phase, int64_t* alt_cost, int32_t* rounds, int32_t* hops, int32_t* capped) {
  int32_t was_capped = 0;
  const int32_t div = p->refine_div > 0 ? p->refine_div : 1;
  /* theta may leave the grid's range by refine_th_margin grid steps (the extent of the GPU kernel's cos/sin table) */
  const int32_t th_lo = -p->refine_th_margin * div, th_hi = (p->n_th - 1 + p->refine_th_margin) * div;
  /* one square along y / z in lattice units */
  const int32_t hop_y = (int32_t)lround(p->grid_length / (p->ty_step / (double)div));
  const int32_t hop_z = (int32_t)lround(p->grid_length / (p->tz_step / (double)div));
  int64_t c = lattice_cost(y, z, label, m, p, lat, *phase);
  int64_t alt = INT64_MAX;
  int32_t n_rounds = 0, n_hops = 0;
  for (;;) {
    /* pattern search at strides div, div/2, ..., 1 */
    int32_t stride = div, r = 0;
    while (stride >= 1 && r < p->refine_max_rounds && p->refine_div > 0) {
      int64_t bc = INT64_MAX;
      int32_t bd = 0, bq[3] = {0, 0, 0};
      for (int32_t dk = -1; dk <= 1; ++dk)
        for (int32_t da = -1; da <= 1; ++da)
          for (int32_t db = -1; db <= 1; ++db) {
            if (dk == 0 && da == 0 && db == 0) continue;
            const int32_t q[3] = {lat[0] + dk * stride, lat[1] + da * stride, lat[2] + db * stride};
            if (q[0] < th_lo || q[0] > th_hi) continue;
            const int64_t cc = lattice_cost(y, z, label, m, p, q, *phase);
            const int32_t d2 = dk * dk + da * da + db * db;
            if (cc < bc || (cc == bc && d2 < bd)) { /* ties: nearer, then first in (dk, da, db) order */
              bc = cc;
              bd = d2;
              bq[0] = q[0];
              bq[1] = q[1];
              bq[2] = q[2];
            }
          }
      ++r;
      if (bc < c) {
        c = bc;
        lat[0] = bq[0];
        lat[1] = bq[1];
        lat[2] = bq[2];
      } else {
        stride >>= 1;
      }
    }
    n_rounds += r;
    if (p->refine_div > 0 && stride >= 1) was_capped = 1; /* left the loop on the round cap, not on the stride */
    /* the eight neighbouring basins: one square along y and/or z; an odd shift swaps the colours */
    alt = INT64_MAX;
    int32_t aq[3] = {0, 0, 0}, aph = 0;
    for (int32_t da = -1; da <= 1; ++da)
      for (int32_t db = -1; db <= 1; ++db) {
        if (da == 0 && db == 0) continue;
        const int32_t q[3] = {lat[0], lat[1] + da * hop_y, lat[2] + db * hop_z};
        const int32_t ph = *phase ^ ((da + db) & 1);
        const int64_t cc = lattice_cost(y, z, label, m, p, q, ph);
        if (cc < alt) { /* ties: first in (da, db) order */
          alt = cc;
          aq[0] = q[0];
          aq[1] = q[1];
          aq[2] = q[2];
          aph = ph;
        }
      }
    if (alt < c && n_hops < 2 && p->refine_div > 0) {
      ++n_hops;
      c = alt;
      lat[0] = aq[0];
      lat[1] = aq[1];
      lat[2] = aq[2];
      *phase = aph;
      continue;
    }
    break;
  }
  if (alt_cost) *alt_cost = alt;
  if (rounds) *rounds = n_rounds;
  if (hops) *hops = n_hops;
  if (capped) *capped = was_capped;
  return c;
}

/* ------------------------------------------------------------------------- */
/* coverage of the virtual board by the labelled points (confidence signal)   */
/* ------------------------------------------------------------------------- */
void orc_coverage(const double theta_t[3], const float* y, const float* z, int32_t m, const orc_params* p, int32_t* cells_hit,
                  int32_t* n_oob) {
  const int32_t W = p->board_w, H = p->board_h;
  const double g = p->grid_length, c = cos(theta_t[0]), s = sin(theta_t[0]);
  uint8_t* hit = (uint8_t*)calloc((size_t)(W * H), 1);
  int32_t oob = 0, cells = 0;
  for (int32_t k = 0; k < m; ++k) {
    const double yy = c * (double)y[k] - s * (double)z[k] + theta_t[1]; /* Optimization.h:37-42 */
    const double zz = s * (double)y[k] + c * (double)z[k] + theta_t[2];
    const double i = (yy + W * g / 2.0) / g, j = (zz + H * g / 2.0) / g; /* :45-46 */
    if (i > 0.0 && i < (double)W && j > 0.0 && j < (double)H) {            /* :48-49, strict */
      hit[(int32_t)floor(i) * H + (int32_t)floor(j)] = 1;
    } else {
      ++oob;
    }
  }
  for (int32_t k = 0; k < W * H; ++k) cells += hit[k];
  free(hit);
  *cells_hit = cells;
  *n_oob = oob;
}

/* ------------------------------------------------------------------------- */
/* a9  getPCDcorners (LidarCornersEst.cpp:501-556), inverse = false            */
/* ------------------------------------------------------------------------- */
static void inv_rigid_apply(const float T[16], const float in[3], float out[3]) {
  /* inverse of [R|t] applied to a point: R^T (p - t), float */
  const float dx = in[0] - T[3], dy = in[1] - T[7], dz = in[2] - T[11];
  for (int c = 0; c < 3; ++c) {
    float s = T[0 + c] * dx;
    s = s + T[4 + c] * dy;
    s = s + T[8 + c] * dz;
    out[c] = s;
  }
}

int32_t orc_corners(const float pca[16], const double theta_t[3], const orc_params* p,
                    float* corners) {
  /* transf = pcl::getTransformation(0, ty, tz, theta, 0, 0) : float Affine3f (:412) */
  const float roll = (float)theta_t[0];
  const float E = cosf(roll), F = sinf(roll);
  float T[16] = {1, 0, 0, 0, 0, E, -F, (float)theta_t[1], 0, F, E, (float)theta_t[2], 0, 0, 0, 1};
  const int W = p->board_w, H = p->board_h;
  int32_t n = 0;
  for (int i = 1; i < W; ++i)       /* x_grid_arr :513-516 */
    for (int j = 1; j < H; ++j) {   /* y_grid_arr :517-520 */
      const double xg = (i - (double)W / 2.0) * p->grid_length;
      const double yg = (j - (double)H / 2.0) * p->grid_length;
      const float pt[3] = {0.0f, (float)xg, (float)yg}; /* :528-530 */
      float q[3], w[3];
      inv_rigid_apply(T, pt, q);   /* transOptim.inverse() :548 */
      inv_rigid_apply(pca, q, w);  /* transPCA.inverse()   :549 */
      if (n >= ORC_MAX_CORNERS) return n;
      corners[3 * n] = w[0];
      corners[3 * n + 1] = w[1];
      corners[3 * n + 2] = w[2];
      ++n;
    }
  return n;
}

/* ------------------------------------------------------------------------- */
/* whole path (get_lidar_corners.cpp:183-201, LidarCornersEst.cpp:374-450)     */
/* ------------------------------------------------------------------------- */
int32_t orc_extract(const float* xyzi, int32_t n, const float click[3], const orc_params* p,
                    orc_result* out, float* cloud_chessboard, float* cloud_pca) {
  memset(out, 0, sizeof(*out));
  out->grid_index = -1;
  if (n <= 0) {
    out->status = ORC_NO_ROI_POINTS;
    return out->status;
  }
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  float* a = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  float* b = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  int32_t status = ORC_OK;

  /* setROI */
  const int32_t m_roi = orc_roi_crop(xyzi, n, click, p, idx);
  out->n_roi = m_roi;
  if (m_roi == 0) { status = ORC_NO_ROI_POINTS; goto done; }
  for (int32_t i = 0; i < m_roi; ++i) memcpy(a + 4 * i, xyzi + 4 * idx[i], 16);

  /* EuclideanCluster */
  const int32_t m_clu = orc_cluster(a, m_roi, click, p, idx, NULL);
  out->n_cluster = m_clu;
  if (m_clu == 0) { status = ORC_NO_CLUSTER; goto done; }
  for (int32_t i = 0; i < m_clu; ++i) memcpy(b + 4 * i, a + 4 * idx[i], 16);

  /* getPlane */
  const int32_t m_pl = orc_ransac_plane(b, m_clu, p, idx, NULL);
  out->n_plane = m_pl;
  if (m_pl < 3) { status = ORC_NO_PLANE; goto done; }
  for (int32_t i = 0; i < m_pl; ++i) memcpy(a + 4 * i, b + 4 * idx[i], 16);
  if (cloud_chessboard) memcpy(cloud_chessboard, a, sizeof(float) * 4 * (size_t)m_pl);

  /* PCA() : transformbyPCA + get_gray_zone(m_cloud_chessboard, 2.5) */
  status = orc_plane_frame(a, m_pl, p, out->pca, b);
  if (status != ORC_OK) goto done;
  if (cloud_pca) memcpy(cloud_pca, b, sizeof(float) * 4 * (size_t)m_pl);
  {
    float* inten = (float*)malloc(sizeof(float) * (size_t)m_pl);
    for (int32_t i = 0; i < m_pl; ++i) inten[i] = a[4 * i + 3];
    double rlrh[2];
    status = orc_gray_zone(inten, m_pl, p, rlrh, out->gray_zone);
    free(inten);
    if (status != ORC_OK) goto done;
  }

  /* get_corners */
  {
    float* y = (float*)malloc(sizeof(float) * (size_t)m_pl);
    float* z = (float*)malloc(sizeof(float) * (size_t)m_pl);
    int8_t* lab = (int8_t*)malloc((size_t)m_pl);
    int32_t counts[3];
    const int32_t nl = classify(b, m_pl, out->gray_zone, y, z, lab, counts);
    out->n_black = counts[0];
    out->n_gray = counts[1];
    out->n_white = counts[2];
    if (p->solver == ORC_SOLVER_GRID) {
      /* exhaustive grid on the pass-A cost, then the monotone refinement (no Ceres in this mode) */
      double gc = 0;
      const int32_t flat = orc_grid_search(y, z, lab, nl, p, 1, &gc, NULL);
      out->grid_index = flat;
      out->grid_cost = gc;
      int32_t phase = flat & 1;
      const int32_t cell = flat >> 1;
      const int32_t div = p->refine_div > 0 ? p->refine_div : 1;
      int32_t lat[3] = {(cell / (p->n_tz * p->n_ty)) * div, ((cell / p->n_tz) % p->n_ty) * div, (cell % p->n_tz) * div};
      int64_t alt = 0;
      int32_t rounds = 0, hops = 0;
      int32_t capped = 0;
      const int64_t c = orc_pattern_refine2(y, z, lab, nl, p, lat, &phase, &alt, &rounds, &hops, &capped);
      if (capped) out->flags |= ORC_FLAG_REFINE_CAPPED;
      orc_lattice_point(p, lat, out->theta_t);
      out->phase = phase;
      out->iters_a = rounds;
      out->iters_b = hops;
      out->cost_a = out->sel_cost = (double)c / ORC_COST_Q_ONE;
      out->cost_b = (double)alt / ORC_COST_Q_ONE; /* GRID mode: cheapest neighbouring basin */
      out->basin_margin = ((double)alt - (double)c) / (double)(c > 0 ? c : 1);
      if (p->ambiguity_eps > 0.0 && out->basin_margin < p->ambiguity_eps) status = ORC_AMBIGUOUS;
    } else {
    double th[3] = {0, 0, 0};
    lsq_problem q;
    memset(&q, 0, sizeof(q));
    q.n = nl;
    q.y = y;
    q.z = z;
    q.lab = lab;
    q.p = p;
    /* phases to try: the reference's first loop turn uses topleftWhite=false (:379,398-401); the user's 'd' key
     * toggles it (Visualization.cpp:45-48).  phase_mode 2 replaces the key press by trying both
     * from (0,0,0) and keeping the lower final cost measured WITH the out-of-board term. */
    int32_t ph_lo, ph_hi;
    if (p->phase_mode == 2) { ph_lo = 0; ph_hi = 1; }
    else { ph_lo = ph_hi = (p->phase_mode == 1); }
    double best_sel = DBL_MAX;
    for (int32_t ph = ph_lo; ph <= ph_hi; ++ph) {
      double t[3] = {th[0], th[1], th[2]};
      double ca = 0, cb = 0;
      q.tlw = ph;
      q.oob = 1; /* pass A :403-405 */
      const int32_t ia = trust_region_minimize(&q, t, &ca);
      q.oob = 0; /* pass B :406-408 */
      const int32_t ib = trust_region_minimize(&q, t, &cb);
      const double sel = orc_cost(t, y, z, lab, nl, p, ph, 1);
      if (sel < best_sel) {
        best_sel = sel;
        out->iters_a = ia;
        out->iters_b = ib;
        out->cost_a = ca;
        out->cost_b = cb;
        out->sel_cost = sel;
        out->phase = ph;
        for (int c = 0; c < 3; ++c) out->theta_t[c] = t[c];
      }
    }
    }
    orc_coverage(out->theta_t, y, z, nl, p, &out->cells_hit, &out->n_oob);
    if (p->min_cell_coverage > 0.0 && (double)out->cells_hit < p->min_cell_coverage * (double)(p->board_w * p->board_h))
      out->flags |= ORC_FLAG_LOW_COVERAGE;
    free(y);
    free(z);
    free(lab);
  }
  out->n_corners = orc_corners(out->pca, out->theta_t, p, out->corners);

done:
  out->status = status;
  free(idx);
  free(a);
  free(b);
  return status;
}

/* ------------------------------------------------------------------------- */
/* f2  get_chessboard_by_point (LidarCornersEst.cpp:72-115) + colouring        */
/* ------------------------------------------------------------------------- */
/* No ROI crop: the whole (finite) cloud is clustered with p->cluster_tol (the reference hard-codes
 * 0.10 here), the cluster around `point` goes through getPlane; false (ORC_BOARD_NOT_FOUND) when the
 * plane has fewer than min_plane points or no admissible cluster holds the nearest point.  The
 * online node then colours the plane by gray zone (lidar_chessboard_online.cpp:98-99,
 * LidarCornersEst.cpp:452-499): classes 0 black / 1 gray / 2 white. */
int32_t orc_chessboard_by_point(const float* xyzi, int32_t n, const float point[3], const orc_params* p,
                                int32_t min_plane, orc_result* out, float* cloud_chessboard,
                                uint8_t* classes) {
  memset(out, 0, sizeof(*out));
  out->grid_index = -1;
  if (n <= 0) {
    out->status = ORC_NO_ROI_POINTS;
    return out->status;
  }
  orc_params q = *p;
  q.roi_half[0] = q.roi_half[1] = q.roi_half[2] = (double)INFINITY;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  float* a = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  float* b = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  int32_t status = ORC_OK, found = 0;
  const int32_t m_all = orc_roi_crop(xyzi, n, point, &q, idx); /* drops non-finite points only */
  out->n_roi = m_all;
  if (m_all == 0) { status = ORC_NO_ROI_POINTS; goto done; }
  for (int32_t i = 0; i < m_all; ++i) memcpy(a + 4 * i, xyzi + 4 * idx[i], 16);
  const int32_t m_clu = orc_cluster2(a, m_all, point, &q, idx, NULL, &found);
  out->n_cluster = m_clu;
  if (m_clu == 0) { status = ORC_NO_CLUSTER; goto done; }
  for (int32_t i = 0; i < m_clu; ++i) memcpy(b + 4 * i, a + 4 * idx[i], 16);
  const int32_t m_pl = orc_ransac_plane(b, m_clu, &q, idx, NULL);
  out->n_plane = m_pl;
  if (m_pl < 3) { status = ORC_NO_PLANE; goto done; }
  for (int32_t i = 0; i < m_pl; ++i) memcpy(a + 4 * i, b + 4 * idx[i], 16);
  if (cloud_chessboard) memcpy(cloud_chessboard, a, sizeof(float) * 4 * (size_t)m_pl);
  status = orc_plane_frame(a, m_pl, &q, out->pca, NULL);
  if (status != ORC_OK) goto done;
  {
    float* inten = (float*)malloc(sizeof(float) * (size_t)m_pl);
    for (int32_t i = 0; i < m_pl; ++i) inten[i] = a[4 * i + 3];
    double rlrh[2];
    status = orc_gray_zone(inten, m_pl, &q, rlrh, out->gray_zone);
    if (status == ORC_OK)
      for (int32_t i = 0; i < m_pl; ++i) {
        const double v = (double)inten[i];
        const uint8_t cl = (v < out->gray_zone[0]) ? 0 : ((v > out->gray_zone[1]) ? 2 : 1);
        if (classes) classes[i] = cl;
        if (cl == 0) out->n_black++;
        else if (cl == 1) out->n_gray++;
        else out->n_white++;
      }
    free(inten);
    if (status != ORC_OK) goto done;
  }
  if (m_pl < min_plane || !found) status = ORC_BOARD_NOT_FOUND; /* :111-112 */
done:
  out->status = status;
  out->phase = found; /* reported through the phase field: find_board */
  free(idx);
  free(a);
  free(b);
  return status;
}

/* ------------------------------------------------------------------------- */
/* a11 save_corners2txt number formatting (get_lidar_corners.cpp:33)           */
/* ------------------------------------------------------------------------- */
/* ostream << float with default flags: precision 6, %g conversion of the value widened to
 * double (std::num_put). */
int32_t orc_format_float(float v, char* buf, int32_t cap) {
  return (int32_t)snprintf(buf, (size_t)cap, "%g", (double)v);
}


/* ------------------------------------------------------------------------- */
/* f4: LiDAR -> image projection (after calibration)                          */
/* ------------------------------------------------------------------------- */

/* ImageCornersEst::spaceToPlane (src/ImageCornersEst.cpp:135-155) */
static int space_to_plane(const orc_camera_model* c, const float* q, double dis, double* cu, double* cv) {
  const double X = (double)q[0], Y = (double)q[1], Z = (double)q[2];
  const double pc0 = c->R[0] * X + c->R[1] * Y + c->R[2] * Z + c->t[0];
  const double pc1 = c->R[3] * X + c->R[4] * Y + c->R[5] * Z + c->t[1];
  const double pc2 = c->R[6] * X + c->R[7] * Y + c->R[8] * Z + c->t[2];
  if (pc2 < 0 || pc2 > dis) return 0;
  const double u = pc0 / pc2, v = pc1 / pc2;
  *cu = c->fx * u + c->cx;
  *cv = c->fy * v + c->cy;
  return (*cu > 0 && *cu < (double)c->width && *cv > 0 && *cv < (double)c->height) ? 1 : 0;
}

/* what `unsigned char = float` / `int = double` compile to on x86-64 (cvttss2si / cvttsd2si) */
static uint8_t to_u8(float v) { return (uint8_t)(int32_t)v; }
static int32_t x86_d2i(double v) {
  return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000;
}

/* ImageCornersEst::HSVtoRGB (src/ImageCornersEst.cpp:373-428) */
void orc_hsv_to_rgb(int32_t h, int32_t s, int32_t v, uint8_t rgb[3]) {
  const float rgb_max = v * 2.55f;
  const float rgb_min = rgb_max * (100 - s) / 100.0f;
  const int32_t i = h / 60;
  const int32_t difs = h % 60;
  const float adj = (rgb_max - rgb_min) * difs / 60.0f;
  switch (i) {
    case 0: rgb[0] = to_u8(rgb_max); rgb[1] = to_u8(rgb_min + adj); rgb[2] = to_u8(rgb_min); break;
    case 1: rgb[0] = to_u8(rgb_max - adj); rgb[1] = to_u8(rgb_max); rgb[2] = to_u8(rgb_min); break;
    case 2: rgb[0] = to_u8(rgb_min); rgb[1] = to_u8(rgb_max); rgb[2] = to_u8(rgb_min + adj); break;
    case 3: rgb[0] = to_u8(rgb_min); rgb[1] = to_u8(rgb_max - adj); rgb[2] = to_u8(rgb_max); break;
    case 4: rgb[0] = to_u8(rgb_min + adj); rgb[1] = to_u8(rgb_min); rgb[2] = to_u8(rgb_max); break;
    default: rgb[0] = to_u8(rgb_max); rgb[1] = to_u8(rgb_min); rgb[2] = to_u8(rgb_max - adj); break;
  }
}

/* test/pcd2image.cpp:56-82 without the drawing: one record per accepted point */
int32_t orc_project_intensity(const float* xyzi, int32_t n, const orc_camera_model* cam, double dis, double lo,
                              double hi, orc_pixel_hit* hits) {
  int32_t m = 0;
  for (int32_t i = 0; i < n; ++i) {
    double cu, cv;
    if (!space_to_plane(cam, xyzi + 4 * i, dis, &cu, &cv)) continue;
    const double h = ((double)xyzi[4 * i + 3] - lo) / (hi - lo) * 255;
    uint8_t rgb[3];
    orc_hsv_to_rgb(x86_d2i(h), 100, 100, rgb);
    hits[m].x = (int32_t)cu;
    hits[m].y = (int32_t)cv;
    hits[m].r = rgb[0];
    hits[m].g = rgb[1];
    hits[m].b = rgb[2];
    hits[m].pad = 0;
    hits[m].index = (uint32_t)i;
    ++m;
  }
  return m;
}

/* test/rgblidar.cpp:50-74: XYZ + packed rgb of the BGR image at the truncated pixel */
int32_t orc_colourise(const float* xyzi, int32_t n, const orc_camera_model* cam, double dis, const uint8_t* image_bgr,
                      uint32_t image_step, float* xyzrgb) {
  int32_t m = 0;
  for (int32_t i = 0; i < n; ++i) {
    double cu, cv;
    if (!space_to_plane(cam, xyzi + 4 * i, dis, &cu, &cv)) continue;
    const int32_t x = (int32_t)cu, y = (int32_t)cv;
    const uint8_t* p = image_bgr + (size_t)y * image_step + (size_t)x * 3;
    const uint32_t rgb = ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[0];
    xyzrgb[4 * m + 0] = xyzi[4 * i + 0];
    xyzrgb[4 * m + 1] = xyzi[4 * i + 1];
    xyzrgb[4 * m + 2] = xyzi[4 * i + 2];
    memcpy(&xyzrgb[4 * m + 3], &rgb, 4);
    ++m;
  }
  return m;
}

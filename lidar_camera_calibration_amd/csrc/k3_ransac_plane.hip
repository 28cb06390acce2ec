// K3 ransac_plane -- replaces pcl::SACSegmentation (SACMODEL_PLANE, SAC_RANSAC, threshold
// 0.03 m, optimize coefficients) of LidarCornersEst::getPlane
// (/root/reference/ilcc2/src/LidarCornersEst.cpp:190-221).
//
// One workgroup per frame (4 wavefronts; 16 in small batches).  Each wavefront scores whole
// hypotheses: the three sample indices come from a counter-based hash (PCL's boost::mt19937
// stream cannot be reproduced without PCL), every lane strides over the cluster points and
// the inlier count (|n.p+d| < thr, strict, float, unfused) is reduced with ballot/popcount.
// How many hypotheses: PCL's own rule (RandomSampleConsensus: k = log(1 - p) / log(1 - w^3), round 5).
// Winner = most inliers, ties -> lowest hypothesis index.  Then PCL's refinement:
// PCA plane of the inliers (double accumulation, Jacobi eigen-solver) and re-selection of
// the inliers with the refined plane, emitted in input order (= m_cloud_chessboard).
#include "eig3.h"
#include "ilcc_internal.h"

namespace ilcc {

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t sample_index(uint32_t seed, uint32_t h, uint32_t k, uint32_t m) {
  const uint32_t r = hash_u32(seed ^ hash_u32(h * 3u + k + 0x9E3779B9u));
  return (uint32_t)(((uint64_t)r * (uint64_t)m) >> 32);
}

__device__ __forceinline__ bool plane_from_3(const float4 p0, const float4 p1, const float4 p2,
                                             float pl[4]) {
  const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
  const float bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
  float nx = ay * bz - az * by;
  float ny = az * bx - ax * bz;
  float nz = ax * by - ay * bx;
  float n2 = nx * nx;
  n2 = n2 + ny * ny;
  n2 = n2 + nz * nz;
  if (!(n2 > 1e-12f)) return false;
  const float nrm = sqrtf(n2);
  nx = nx / nrm;
  ny = ny / nrm;
  nz = nz / nrm;
  float d = nx * p0.x;
  d = d + ny * p0.y;
  d = d + nz * p0.z;
  pl[0] = nx;
  pl[1] = ny;
  pl[2] = nz;
  pl[3] = -d;
  return true;
}

__device__ __forceinline__ float plane_dist(const float pl[4], const float4 q) {
  float s = pl[0] * q.x;
  s = s + pl[1] * q.y;
  s = s + pl[2] * q.z;
  s = s + pl[3];
  return fabsf(s);
}

constexpr int kRansacLdsPoints = 2048;   // cluster points staged in LDS (32 KiB); larger clusters are read through L2

// Launched with kPlaneThreads (256) per frame in large batches -- the kernel then runs BESIDE another batch's K6 and small
// workgroups leave it the CUs -- and with kPlaneThreadsSmallBatch (1024) when the batch has too few frames to fill the
// chip anyway (one click of the reference's node; config 5's 64 dense frames): 16 instead of 4 wavefronts share the 128
// hypotheses.  Results do not depend on the width (hypotheses are ranked by (inliers, index), sums are block sums).
__global__ __launch_bounds__(kPlaneThreadsSmallBatch) void k3_ransac_plane(Ctx c) {
  const uint32_t kThreads = blockDim.x;
  __shared__ float4 s_P[kRansacLdsPoints];
  __shared__ uint32_t sc[64];
  __shared__ double scd[16 * 6 + 8];
  __shared__ float s_plane[4];
  const uint32_t f = blockIdx.x;
  ilcc_result* r = &c.res[f];
  if (r->status != ILCC_OK) return;
  const uint32_t M = (uint32_t)r->n_cluster;
  const uint64_t beg = c.off[f];
  const float4* __restrict__ G = c.cluster + beg;
  const uint32_t tid = threadIdx.x;
  const int lane = lane_id(), wid = wave_id();
  const float thr = (float)c.p.ransac_thresh;
  if (M < 3) {
    if (tid == 0) r->status = ILCC_NO_PLANE;
    return;
  }
  // every hypothesis re-reads the whole cluster (128 x M points): from LDS, not from L2, when it fits
  const float4* P = G;
  if (M <= (uint32_t)kRansacLdsPoints) {
    for (uint32_t i = tid; i < M; i += kThreads) s_P[i] = G[i];
    __syncthreads();
    P = s_P;
  }

  // ---- score hypotheses, one per wavefront pass
  const uint32_t n_waves = kThreads / ILCC_WAVE;
  auto score = [&](uint32_t h, uint32_t beat, uint32_t& cnt) -> bool {   // false: degenerate sample.  cnt is exact whenever it exceeds `beat`
    const uint32_t i0 = sample_index(c.p.ransac_seed, h, 0, M);
    const uint32_t i1 = sample_index(c.p.ransac_seed, h, 1, M);
    const uint32_t i2 = sample_index(c.p.ransac_seed, h, 2, M);
    float pl[4];
    cnt = 0;
    if (i0 == i1 || i0 == i2 || i1 == i2) return false;
    if (!plane_from_3(P[i0], P[i1], P[i2], pl)) return false;
    for (uint32_t base = 0; base < M; base += ILCC_WAVE) {
      const uint32_t i = base + lane;
      const bool in = (i < M) && plane_dist(pl, P[i]) < thr;
      cnt += (uint32_t)__popcll(__ballot(in));
      // exact early exit: even if every point still to come were an inlier, this hypothesis could not beat (>) a count that
      // has already been reached in full (wave-uniform: a scalar branch)
      if (cnt + (M - min(M, base + (uint32_t)ILCC_WAVE)) <= beat) break;
    }
    return true;
  };
  if (c.p.ransac_probability > 0.0) {
    // pcl::RandomSampleConsensus::computeModel's loop (PCL 1.8 ransac.hpp; SACSegmentation: probability 0.99, max_iterations 50),
    // operation for operation the oracle's orc_ransac_plane: hypotheses are SCORED a round at a time, one per wavefront, and then
    // walked in index order by one thread exactly as the serial loop would -- k = log(1 - p) / log(1 - w^3) after every new
    // best, degenerate samples skipped without counting, stop at iterations >= k.  Hypotheses of the last round that lie behind
    // the stop are ignored: what was scored beyond PCL's last iteration never counts.  A board cluster stops after 3-5 (rounds
    // 1-4 scored 128 whatever the data said: 34.5 M of the path's 385 M VALU instructions per 1024 frames).
    __shared__ uint32_t s_state[5];   // best count, best hypothesis, iterations, skipped, stop
    __shared__ double s_k;
    if (tid == 0) {
      s_state[0] = 0u;
      s_state[1] = 0xFFFFFFFFu;
      s_state[2] = s_state[3] = s_state[4] = 0u;
      s_k = 1.0;
    }
    __syncthreads();
    const double log_probability = log(1.0 - c.p.ransac_probability);
    const double one_over_indices = 1.0 / (double)M;
    const uint32_t max_it = (uint32_t)c.p.ransac_hyp, max_skip = max_it * 10u;
    for (uint32_t h0 = 0;; h0 += n_waves) {
      uint32_t cnt;
      const bool valid = score(h0 + (uint32_t)wid, s_state[0], cnt);
      if (lane == 0) sc[wid] = valid ? cnt : 0xFFFFFFFFu;
      __syncthreads();
      if (tid == 0) {
        uint32_t best = s_state[0], best_h = s_state[1], it = s_state[2], skip = s_state[3], stop = 0u;
        double k = s_k;
        for (uint32_t j = 0; j < n_waves; ++j) {
          if (!((double)it < k && skip < max_skip)) {
            stop = 1u;
            break;
          }
          const uint32_t v = sc[j];
          if (v == 0xFFFFFFFFu) {
            ++skip;
            continue;
          }
          if (v > best) {   // (a count cut short by the early exit is <= the best of the round's start: never taken)
            best = v;
            best_h = h0 + j;
            const double w = (double)best * one_over_indices;
            double p_no_outliers = 1.0 - w * w * w;
            p_no_outliers = fmax(2.220446049250313e-16, p_no_outliers);
            p_no_outliers = fmin(1.0 - 2.220446049250313e-16, p_no_outliers);
            k = log_probability / log(p_no_outliers);
          }
          ++it;
          if (it > max_it) {
            stop = 1u;
            break;
          }
        }
        if (!((double)it < k && skip < max_skip)) stop = 1u;   // (the serial loop's next test: spares a round)
        s_state[0] = best;
        s_state[1] = best_h;
        s_state[2] = it;
        s_state[3] = skip;
        s_state[4] = stop;
        s_k = k;
      }
      __syncthreads();
      if (s_state[4] != 0u) break;
    }
    if (tid == 0) {
      sc[32] = s_state[0];
      sc[33] = s_state[1];
    }
  } else {
    // ransac_probability <= 0: a fixed number of hypotheses (rounds 1-4), most inliers, ties -> lowest index
    uint32_t best_cnt = 0, best_h = 0xFFFFFFFFu;
    for (uint32_t h = (uint32_t)wid; h < (uint32_t)c.p.ransac_hyp; h += n_waves) {
      uint32_t cnt;
      if (!score(h, best_cnt, cnt)) continue;
      if (cnt > best_cnt) {   // h ascending within a wavefront: ties keep the lowest h
        best_cnt = cnt;
        best_h = h;
      }
    }
    if (lane == 0) {
      sc[wid] = best_cnt;
      sc[16 + wid] = best_h;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t bc = 0, bh = 0xFFFFFFFFu;
      for (int w = 0; w < (int)n_waves; ++w)
        if (sc[w] > bc || (sc[w] == bc && sc[w] > 0 && sc[16 + w] < bh)) {
          bc = sc[w];
          bh = sc[16 + w];
        }
      sc[32] = bc;
      sc[33] = bh;
    }
  }
  if (tid == 0 && sc[32] > 0u) {
    const uint32_t bh = sc[33];
    float pl[4];
    plane_from_3(P[sample_index(c.p.ransac_seed, bh, 0, M)], P[sample_index(c.p.ransac_seed, bh, 1, M)],
                 P[sample_index(c.p.ransac_seed, bh, 2, M)], pl);
    for (int k = 0; k < 4; ++k) s_plane[k] = pl[k];
  }
  __syncthreads();
  const uint32_t bc = sc[32];
  if (bc == 0) {
    if (tid == 0) r->status = ILCC_NO_PLANE;
    return;
  }
  float pl[4] = {s_plane[0], s_plane[1], s_plane[2], s_plane[3]};

  // ---- optimizeModelCoefficients: PCA plane of the inliers (needs > 3 of them)
  if (bc > 3) {
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t i = tid; i < M; i += kThreads) {
      const float4 q = P[i];
      if (plane_dist(pl, q) < thr) {
        sx += q.x;
        sy += q.y;
        sz += q.z;
      }
    }
    double sums3[3] = {sx, sy, sz};
    block_sum_n<3>(sums3, scd);
    const double cx = sums3[0] / bc, cy = sums3[1] / bc, cz = sums3[2] / bc;
    double cv[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = tid; i < M; i += kThreads) {
      const float4 q = P[i];
      if (plane_dist(pl, q) < thr) {
        const double dx = q.x - cx, dy = q.y - cy, dz = q.z - cz;
        cv[0] += dx * dx;
        cv[1] += dx * dy;
        cv[2] += dx * dz;
        cv[3] += dy * dy;
        cv[4] += dy * dz;
        cv[5] += dz * dz;
      }
    }
    __syncthreads();   // scd is reused
    block_sum_n<6>(cv, scd);
    double cs[6];
    for (int k = 0; k < 6; ++k) cs[k] = cv[k] / bc;
    if (tid == 0) {
      const double cov[9] = {cs[0], cs[1], cs[2], cs[1], cs[3], cs[4], cs[2], cs[4], cs[5]};
      double w[3], v[3][3];
      eig3_sym(cov, w, v);
      double n[3] = {v[0][0], v[0][1], v[0][2]};
      if (n[0] * pl[0] + n[1] * pl[1] + n[2] * pl[2] < 0) {
        n[0] = -n[0];
        n[1] = -n[1];
        n[2] = -n[2];
      }
      s_plane[0] = (float)n[0];
      s_plane[1] = (float)n[1];
      s_plane[2] = (float)n[2];
      s_plane[3] = (float)(-(n[0] * cx + n[1] * cy + n[2] * cz));
    }
    __syncthreads();
    for (int k = 0; k < 4; ++k) pl[k] = s_plane[k];
  }

  // ---- re-select inliers with the refined plane, stable order
  float4* __restrict__ dst = c.board + beg;
  uint32_t running = 0;
  for (uint32_t base = 0; base < M; base += kThreads) {
    const uint32_t i = base + tid;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    bool keep = false;
    if (i < M) {
      q = P[i];
      keep = plane_dist(pl, q) < thr;
    }
    uint32_t tot;
    const uint32_t rank = block_rank(keep, sc + 40, tot);
    if (keep) dst[running + rank] = q;
    running += tot;
  }
  if (tid == 0) {
    r->n_plane = (int32_t)running;
    for (int k = 0; k < 4; ++k) r->plane[k] = pl[k];
    if (running < 3) r->status = ILCC_NO_PLANE;
  }
}

void launch_ransac_plane(const Ctx& c, hipStream_t s) {
  const int threads = (c.n_frames <= (uint32_t)kSmallBatchFrames || c.wide) ? kPlaneThreadsSmallBatch : kPlaneThreads;
  hipLaunchKernelGGL(k3_ransac_plane, dim3(c.n_frames), dim3(threads), 0, s, c);
}

}  // namespace ilcc

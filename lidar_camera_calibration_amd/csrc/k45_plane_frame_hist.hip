// K4 plane_frame + K5 gray_zone_hist, fused: one workgroup of kHistThreads (256) threads per frame.
//
// K4 replaces LidarCornersEst::transformbyPCA (/root/reference/ilcc2/src/LidarCornersEst.cpp:330-364):
//   compute3DCentroid, computeCovarianceMatrixNormalized (/N), SelfAdjointEigenSolver
//   (ascending), col(2) = col(0) x col(1), T = [E^T | -E^T c], transformPointCloud.
//   Accumulation is double (PCL: float); the eigenvector sign convention (Eigen: unspecified)
//   is: normal towards the sensor, e1's largest-magnitude component positive.
// K5 replaces calHist + get_gray_zone (:224-328) on the intensities of m_cloud_chessboard and
//   the black/gray/white classification of Optimization::get_theta_t
//   (/root/reference/ilcc2/src/Optimization.cpp:114-125); the non-gray points are written as a
//   compact (y,z) + label stream, the only thing the cost kernels read.
#include "eig3.h"
#include "ilcc_internal.h"

namespace ilcc {

__global__ __launch_bounds__(kHistThreads) void k45_plane_frame_hist(Ctx c) {
  __shared__ uint32_t sc[64];
  __shared__ double scd[16 * 6 + 8];
  __shared__ float s_pca[16];
  __shared__ float s_mm[34];   // [0..16): per-wavefront minima, [16..32): maxima, 32/33: the totals (<= 16 wavefronts)
  __shared__ double s_gz[2];
  __shared__ int s_status;
  extern __shared__ int s_hist[];   // hist_bins + 1 counters

  const uint32_t f = blockIdx.x;
  ilcc_result* r = &c.res[f];
  if (r->status != ILCC_OK) return;
  const uint32_t M = (uint32_t)r->n_plane;
  const uint64_t beg = c.off[f];
  const float4* __restrict__ P = c.board + beg;
  const uint32_t tid = threadIdx.x;
  const int lane = lane_id(), wid = wave_id();
  if (M < 3) {
    if (tid == 0) r->status = ILCC_TOO_FEW_POINTS;
    return;
  }

  // ------------------------------------------------------------------ K4
  double sx = 0, sy = 0, sz = 0, si = 0;
  float vmin = 3.402823466e38f, vmax = -3.402823466e38f;
  for (uint32_t i = tid; i < M; i += kHistThreads) {
    const float4 q = P[i];
    sx += q.x;
    sy += q.y;
    sz += q.z;
    si += (double)q.w;
    vmin = fminf(vmin, q.w);
    vmax = fmaxf(vmax, q.w);
  }
  // centroid narrowed to float like pcl's Vector4f, then used in double
  double s4[4] = {sx, sy, sz, si};
  block_sum_n<4>(s4, scd);   // one pair of barriers, totals bit-identical to four block_sum calls
  const double cx = (double)(float)(s4[0] / M);
  const double cy = (double)(float)(s4[1] / M);
  const double cz = (double)(float)(s4[2] / M);
  const double isum = s4[3];
  double cv[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < M; i += kHistThreads) {
    const float4 q = P[i];
    const double dx = q.x - cx, dy = q.y - cy, dz = q.z - cz;
    cv[0] += dx * dx;
    cv[1] += dx * dy;
    cv[2] += dx * dz;
    cv[3] += dy * dy;
    cv[4] += dy * dz;
    cv[5] += dz * dz;
  }
  __syncthreads();   // scd is reused
  block_sum_n<6>(cv, scd);
  double cs[6];
  for (int k = 0; k < 6; ++k) cs[k] = cv[k] / M;
  // min / max intensity
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vmin = fminf(vmin, __shfl_down(vmin, o, ILCC_WAVE));
    vmax = fmaxf(vmax, __shfl_down(vmax, o, ILCC_WAVE));
  }
  if (lane == 0) {
    s_mm[wid] = vmin;
    s_mm[16 + wid] = vmax;
  }
  __syncthreads();
  if (tid == 0) {
    const double cov[9] = {cs[0], cs[1], cs[2], cs[1], cs[3], cs[4], cs[2], cs[4], cs[5]};
    double w[3], v[3][3];
    eig3_sym(cov, w, v);
    double e0[3] = {v[0][0], v[0][1], v[0][2]};
    double e1[3] = {v[1][0], v[1][1], v[1][2]};
    if (e0[0] * cx + e0[1] * cy + e0[2] * cz > 0) {
      e0[0] = -e0[0];
      e0[1] = -e0[1];
      e0[2] = -e0[2];
    }
    int big = 0;
    for (int a = 1; a < 3; ++a)
      if (fabs(e1[a]) > fabs(e1[big])) big = a;
    if (e1[big] < 0) {
      e1[0] = -e1[0];
      e1[1] = -e1[1];
      e1[2] = -e1[2];
    }
    float f0[3], f1[3], f2[3];
    for (int a = 0; a < 3; ++a) {
      f0[a] = (float)e0[a];
      f1[a] = (float)e1[a];
    }
    f2[0] = f0[1] * f1[2] - f0[2] * f1[1];   // col(2) = col(0).cross(col(1)) in float (:343)
    f2[1] = f0[2] * f1[0] - f0[0] * f1[2];
    f2[2] = f0[0] * f1[1] - f0[1] * f1[0];
    const float cf[3] = {(float)cx, (float)cy, (float)cz};
    const float* rows[3] = {f0, f1, f2};
    for (int rr = 0; rr < 3; ++rr) {
      for (int a = 0; a < 3; ++a) s_pca[4 * rr + a] = rows[rr][a];
      float t = rows[rr][0] * cf[0];
      t = t + rows[rr][1] * cf[1];
      t = t + rows[rr][2] * cf[2];
      s_pca[4 * rr + 3] = -1.0f * t;         // :349
    }
    s_pca[12] = s_pca[13] = s_pca[14] = 0.f;
    s_pca[15] = 1.f;
    for (int k = 0; k < 16; ++k) r->pca[k] = s_pca[k];
    float mn = s_mm[0], mx = s_mm[16];
    for (int w2 = 1; w2 < kHistThreads / ILCC_WAVE; ++w2) {
      mn = fminf(mn, s_mm[w2]);
      mx = fmaxf(mx, s_mm[16 + w2]);
    }
    s_mm[32] = mn;
    s_mm[33] = mx;
  }
  const int HL = c.p.hist_bins;
  for (int b = (int)tid; b <= HL; b += kHistThreads) s_hist[b] = 0;
  __syncthreads();

  // transformPointCloud (float, unfused) -> m_cloud_PCA
  float4* __restrict__ Q = c.pca + beg;
  const double mn = (double)s_mm[32], mx = (double)s_mm[33];
  const bool flat = !(mx > mn);
  const double factor = flat ? 0.0 : HL / (mx - mn);   // :235
  for (uint32_t i = tid; i < M; i += kHistThreads) {
    const float4 q = P[i];
    float o[3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      float s = s_pca[4 * rr] * q.x;
      s = s + s_pca[4 * rr + 1] * q.y;
      s = s + s_pca[4 * rr + 2] * q.z;
      s = s + s_pca[4 * rr + 3];
      o[rr] = s;
    }
    Q[i] = make_float4(o[0], o[1], o[2], q.w);
    // ---------------------------------------------------------------- K5 histogram (:237-241)
    if (!flat) {
      const double sample = (double)q.w - mn;
      int bin = (int)round(sample * factor);   // == HL for the maximum: the reference's UB write,
      bin = bin < 0 ? 0 : (bin > HL ? HL : bin);   // counted in a spare slot and ignored
      atomicAdd(&s_hist[bin], 1);
    }
  }
  __syncthreads();

  // std::map<count, first bin with that count>, walked from the largest count down until one edge above and one
  // below the mean have been seen (:261-282).  Equivalent, and parallel over the bins: among the bins that are
  // the FIRST with their count, `high` is the edge of the one with the largest count on the upper side of the
  // mean, `low` the same on the lower side (representatives have distinct counts, so there are no ties; an edge
  // equal to the mean is on neither side).  The serial walk by one thread was ~50 us of this kernel.
  __shared__ unsigned long long s_top[2];   // per side of the mean: (count + 1) << 32 | bin, 0 = none
  if (tid == 0) s_top[0] = s_top[1] = 0ull;
  __syncthreads();
  const bool hist_ok = !(flat || HL <= 0);
  const double mean = isum / M;              // :245-248
  const double bin_width = (mx - mn) / HL;   // :258
  if (hist_ok) {
    for (int bb = (int)tid; bb < HL; bb += kHistThreads) {
      const int cb = s_hist[bb];
      bool first = true;
      for (int b2 = 0; b2 < bb; ++b2) first = first && (s_hist[b2] != cb);
      if (!first) continue;
      const double edge = bin_width * (double)bb + mn;   // :269
      const int side = edge > mean ? 1 : (edge < mean ? 0 : -1);
      if (side >= 0) atomicMax(&s_top[side], ((unsigned long long)(uint32_t)(cb + 1) << 32) | (unsigned long long)(uint32_t)bb);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int status = ILCC_OK;
    if (!hist_ok || s_top[0] == 0ull || s_top[1] == 0ull) {
      status = ILCC_DEGENERATE_HIST;
    } else {
      const double low = bin_width * (double)(uint32_t)(s_top[0] & 0xFFFFFFFFull) + mn;
      const double high = bin_width * (double)(uint32_t)(s_top[1] & 0xFFFFFFFFull) + mn;
      const double rate = c.p.gray_rate;
      s_gz[0] = ((rate - 1) * low + high) / rate;   // :322
      s_gz[1] = (low + (rate - 1) * high) / rate;   // :323
      r->gray_zone[0] = s_gz[0];
      r->gray_zone[1] = s_gz[1];
    }
    s_status = status;
    if (status != ILCC_OK) r->status = status;
  }
  __syncthreads();
  if (s_status != ILCC_OK) return;

  // ---- classification (Optimization.cpp:114-125) + compact (y,z,label) stream, input order
  const double gz0 = s_gz[0], gz1 = s_gz[1];
  float2* __restrict__ YZ = c.yz + beg;
  uint8_t* __restrict__ LB = c.lab + beg;
  uint8_t* __restrict__ CL = c.cls + beg;
  uint32_t running = 0, nb = 0, nw = 0;
  for (uint32_t base = 0; base < M; base += kHistThreads) {
    const uint32_t i = base + tid;
    bool keep = false;
    uint8_t l = 0;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < M) {
      q = Q[i];
      if ((double)q.w < gz0) {
        keep = true;
        l = 0;
      } else if ((double)q.w > gz1) {
        keep = true;
        l = 1;
      }
      CL[i] = keep ? (uint8_t)(2 * l) : (uint8_t)1;   // color_by_gray_zone thresholds (:465-485)
    }
    uint32_t tot, totw;
    const uint32_t rank = block_rank(keep, sc, tot);
    (void)block_rank(keep && l == 1, sc + 20, totw);
    if (keep) {
      YZ[running + rank] = make_float2(q.y, q.z);   // laserPoint(temp.y, temp.z) :127
      LB[running + rank] = l;
    }
    running += tot;
    nw += totw;
    nb += tot - totw;
  }
  if (tid == 0) {
    r->n_black = (int32_t)nb;
    r->n_white = (int32_t)nw;
    r->n_gray = (int32_t)(M - running);
    c.n_lab[f] = running;
    c.walk_stride[f] = walk_stride(running);
  }
}

void launch_plane_frame_hist(const Ctx& c, hipStream_t s) {
  const size_t lds = sizeof(int) * (size_t)(c.p.hist_bins + 1);
  hipLaunchKernelGGL(k45_plane_frame_hist, dim3(c.n_frames), dim3(kHistThreads), lds, s, c);
}

}  // namespace ilcc

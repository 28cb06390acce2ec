// K7 refine_and_corners -- one 256-thread workgroup per frame.
//
//  (1) picks the start: the argmin of the K6 per-workgroup partials (ILCC_SOLVER_GRID) or
//      (0,0,0) (ILCC_SOLVER_REFERENCE_LOCAL, the reference's own start);
//  (2) runs the reference's two local solves, pass A (useOutofBoard = true) then pass B (false)
//      -- LidarCornersEst::get_corners, /root/reference/ilcc2/src/LidarCornersEst.cpp:398-409 --
//      each a restatement of what ceres::Solve does for Optimization::get_theta_t
//      (/root/reference/ilcc2/src/Optimization.cpp:94-160): TRUST_REGION, DOGLEG/SUBSPACE_DOGLEG,
//      DENSE_NORMAL_CHOLESKY, HuberLoss(0.1) through Ceres' Corrector, Jacobi scaling, Ceres 1.14
//      default tolerances.  Residuals/Jacobians are evaluated in double by all threads; the
//      3-parameter trust-region bookkeeping runs on thread 0;
//  (3) builds the corner lattice: LidarCornersEst::getPCDcorners (:501-556) with
//      transf = pcl::getTransformation(0, ty, tz, theta, 0, 0) (:412), and the display cloud
//      m_cloud_optim (:413).
#include "ilcc_internal.h"

namespace ilcc {

// ------------------------------------------------------------------ residual (Optimization.h:31-107)
struct Board {
  double W, H, g, delta;
};

// raw residual; jac = d r / d(theta, ty, tz) when JAC.  cs = (cos theta, sin theta).
template <bool JAC>
__device__ __forceinline__ double residual(const double x[3], double c, double s, double y, double z,
                                           const Board& bd, bool tlw, bool laser_white, bool use_oob,
                                           double jac[3]) {
  const double ry = c * y - s * z;
  const double rz = s * y + c * z;
  const double r1 = ry + x[1];
  const double r2 = rz + x[2];
  const double i = (r1 + bd.W * bd.g / 2.0) / bd.g;
  const double j = (r2 + bd.H * bd.g / 2.0) / bd.g;
  double si = 0, sj = 0, res = 0;
  if (i > 0 && i < bd.W && j > 0 && j < bd.H) {
    const double ifl = floor(i), jfl = floor(j);
    const double ii = floor(ifl / 2.0) * 2.0, jj = floor(jfl / 2.0) * 2.0;
    bool white = !tlw;
    if (ifl == ii && jfl == jj) white = tlw;
    if (ifl != ii && jfl != jj) white = tlw;
    if (laser_white != white) {
      double ie, je;
      if (i - ifl > 0.5) { ie = ceil(i) - i; si = -1; } else { ie = i - ifl; si = 1; }
      if (j - jfl > 0.5) { je = ceil(j) - j; sj = -1; } else { je = j - jfl; sj = 1; }
      res = ie + je;
    }
  } else if (use_oob) {
    double ie, je;
    if (fabs(i) < fabs(i - bd.W)) { ie = fabs(i); si = (i < 0) ? -1 : 1; }
    else { ie = fabs(i - bd.W); si = (i - bd.W < 0) ? -1 : 1; }
    if (fabs(j) < fabs(j - bd.H)) { je = fabs(j); sj = (j < 0) ? -1 : 1; }
    else { je = fabs(j - bd.H); sj = (j - bd.H < 0) ? -1 : 1; }
    res = ie + je;
  }
  if (JAC) {
    const double dith = (-s * y - c * z) / bd.g, djth = (c * y - s * z) / bd.g;
    jac[0] = si * dith + sj * djth;
    jac[1] = si / bd.g;
    jac[2] = sj / bd.g;
  }
  return res;
}

__device__ __forceinline__ void huber(double a, double s, double& rho0, double& rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho0 = 2.0 * a * r - b;
    rho1 = a / r;
    if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
  } else {
    rho0 = s;
    rho1 = 1.0;
  }
}

// ------------------------------------------------------------------ workgroup-wide evaluation
struct Problem {
  const float2* yz;
  const uint8_t* lab;
  uint32_t n;
  Board bd;
  bool tlw, oob;
};

constexpr int kNW = kSolveThreads / ILCC_WAVE;

// sums[0] = cost ; if JAC: sums[1..3] = J^T r, sums[4..9] = upper J^T J (00,01,02,11,12,22),
// all with Ceres' Corrector applied (rows scaled by sqrt(rho')).  Result valid on every thread.
template <bool JAC>
__device__ void evaluate(const Problem& q, const double x[3], double* s_red /*[kNW*10 + 10]*/,
                         double sums[10]) {
  double sn, cs;
  sincos(x[0], &sn, &cs);
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  for (uint32_t p = threadIdx.x; p < q.n; p += kSolveThreads) {
    const float2 v = q.yz[p];
    double jac[3];
    const double res = residual<JAC>(x, cs, sn, (double)v.x, (double)v.y, q.bd, q.tlw, q.lab[p] != 0,
                                     q.oob, jac);
    double r0, r1;
    huber(q.bd.delta, res * res, r0, r1);
    acc[0] += 0.5 * r0;
    if (JAC) {
      const double sr = sqrt(r1);
      const double rc = sr * res;
      const double j0 = sr * jac[0], j1 = sr * jac[1], j2 = sr * jac[2];
      acc[1] += j0 * rc;
      acc[2] += j1 * rc;
      acc[3] += j2 * rc;
      acc[4] += j0 * j0;
      acc[5] += j0 * j1;
      acc[6] += j0 * j2;
      acc[7] += j1 * j1;
      acc[8] += j1 * j2;
      acc[9] += j2 * j2;
    }
  }
  constexpr int NV = JAC ? 10 : 1;
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = wave_sum(acc[k]);
  __syncthreads();
  if (lane_id() == 0)
    for (int k = 0; k < NV; ++k) s_red[wave_id() * 10 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < NV) {
    double t = s_red[threadIdx.x];
    for (int w = 1; w < kNW; ++w) t += s_red[w * 10 + threadIdx.x];
    s_red[kNW * 10 + threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) sums[k] = s_red[kNW * 10 + k];
}

// ------------------------------------------------------------------ thread-0 dogleg bookkeeping
struct Dog {
  double radius, mu;
  int reuse;
  double diagonal[3], gradient[3], gn[3];
  double alpha, step_norm;
  int one_dim;
  double basis[3][2], sg[2], sB[4];
  double JtJ[9], Jtr[3];   // of the column-scaled Jacobian
};

__device__ inline bool chol3_solve(const double A[9], const double b[3], double x[3]) {
  double L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[3 * i + j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double yv[3];
  for (int i = 0; i < 3; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * yv[k];
    yv[i] = s / L[i][i];
  }
  for (int i = 2; i >= 0; --i) {
    double s = yv[i];
    for (int k = i + 1; k < 3; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  for (int i = 0; i < 3; ++i)
    if (!isfinite(x[i])) return false;
  return true;
}

// argmin of 1/2 y'By + g'y on |y| = radius (Ceres: quartic roots; here scan + Newton polish)
__device__ inline void min_on_circle(const double B[4], const double g[2], double radius, double y[2]) {
  const int NS = 720;
  const double kPi = 3.14159265358979323846;
  double best = 1.7976931348623157e308, bt = 0;
  for (int k = 0; k < NS; ++k) {
    const double t = 2.0 * kPi * k / NS;
    const double a = radius * cos(t), b = radius * sin(t);
    const double f = 0.5 * (B[0] * a * a + 2 * B[1] * a * b + B[3] * b * b) + g[0] * a + g[1] * b;
    if (f < best) {
      best = f;
      bt = t;
    }
  }
  double t = bt;
  for (int it = 0; it < 50; ++it) {
    const double c = cos(t), s = sin(t);
    const double a = radius * c, b = radius * s, da = -radius * s, db = radius * c;
    const double Ba = B[0] * a + B[1] * b, Bb = B[1] * a + B[3] * b;
    const double f1 = Ba * da + Bb * db + g[0] * da + g[1] * db;
    const double Bda = B[0] * da + B[1] * db, Bdb = B[1] * da + B[3] * db;
    const double f2 = Bda * da + Bdb * db + Ba * (-a) + Bb * (-b) + g[0] * (-a) + g[1] * (-b);
    if (!(f2 > 0)) break;
    const double step = f1 / f2;
    if (fabs(step) > kPi / NS) break;
    t -= step;
    if (fabs(step) < 1e-15) break;
  }
  y[0] = radius * cos(t);
  y[1] = radius * sin(t);
}

__device__ inline double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

__device__ inline void dogleg_traditional(Dog& s, double step[3]) {
  const double gnn = norm3(s.gn), gn_ = norm3(s.gradient);
  if (gnn <= s.radius) {
    for (int c = 0; c < 3; ++c) step[c] = s.gn[c] / s.diagonal[c];
    s.step_norm = gnn;
    return;
  }
  if (gn_ * s.alpha >= s.radius) {
    for (int c = 0; c < 3; ++c) step[c] = -(s.radius / gn_) * s.gradient[c] / s.diagonal[c];
    s.step_norm = s.radius;
    return;
  }
  double bdota = 0, a2 = 0, bma2 = 0;
  for (int c = 0; c < 3; ++c) {
    const double a = -s.alpha * s.gradient[c];
    bdota += a * s.gn[c];
    a2 += a * a;
    bma2 += (s.gn[c] - a) * (s.gn[c] - a);
  }
  const double cc = bdota - a2;
  const double d = sqrt(cc * cc + bma2 * (s.radius * s.radius - a2));
  const double beta = (cc <= 0) ? (d - cc) / bma2 : (s.radius * s.radius - a2) / (d + cc);
  for (int c = 0; c < 3; ++c) {
    const double a = -s.alpha * s.gradient[c];
    step[c] = (a + beta * (s.gn[c] - a)) / s.diagonal[c];
  }
  s.step_norm = s.radius;
}

// DoglegStrategy::ComputeStep; JtJ/Jtr (scaled Jacobian) must be current when !reuse.
__device__ inline bool dogleg_compute_step(Dog& s, double step[3]) {
  if (!s.reuse) {
    s.reuse = 1;
    for (int c = 0; c < 3; ++c) {
      double d = s.JtJ[4 * c];
      d = fmin(fmax(d, 1e-6), 1e32);
      s.diagonal[c] = sqrt(d);
      s.gradient[c] = s.Jtr[c] / s.diagonal[c];
    }
    {
      double sgv[3], num = 0, den = 0;
      for (int c = 0; c < 3; ++c) {
        sgv[c] = s.gradient[c] / s.diagonal[c];
        num += s.gradient[c] * s.gradient[c];
      }
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) den += sgv[a] * s.JtJ[3 * a + b] * sgv[b];
      s.alpha = num / den;
    }
    bool ok = false;
    while (s.mu < 1.0) {
      double A[9];
      for (int k = 0; k < 9; ++k) A[k] = s.JtJ[k];
      for (int c = 0; c < 3; ++c) {
        const double lm = s.diagonal[c] * sqrt(s.mu);
        A[4 * c] += lm * lm;
      }
      if (chol3_solve(A, s.Jtr, s.gn)) {
        ok = true;
        break;
      }
      s.mu *= 10.0;
    }
    if (!ok) return false;
    for (int c = 0; c < 3; ++c) s.gn[c] *= -s.diagonal[c];
    {
      double v0[3], v1[3];
      double n0 = 0, n1 = 0;
      for (int c = 0; c < 3; ++c) {
        n0 += s.gradient[c] * s.gradient[c];
        n1 += s.gn[c] * s.gn[c];
      }
      const bool gfirst = n0 >= n1;
      const double nf = sqrt(fmax(n0, n1));
      double dot = 0;
      for (int c = 0; c < 3; ++c) {
        v0[c] = (gfirst ? s.gradient[c] : s.gn[c]) / nf;
      }
      for (int c = 0; c < 3; ++c) dot += (gfirst ? s.gn[c] : s.gradient[c]) * v0[c];
      double nr = 0;
      for (int c = 0; c < 3; ++c) {
        v1[c] = (gfirst ? s.gn[c] : s.gradient[c]) - dot * v0[c];
        nr += v1[c] * v1[c];
      }
      nr = sqrt(nr);
      s.one_dim = !(nr > 3.0 * 2.220446049250313e-16 * nf);
      if (!s.one_dim) {
        double u[2][3];
        for (int c = 0; c < 3; ++c) {
          v1[c] /= nr;
          s.basis[c][0] = v0[c];
          s.basis[c][1] = v1[c];
          u[0][c] = v0[c] / s.diagonal[c];
          u[1][c] = v1[c] / s.diagonal[c];
        }
        for (int a = 0; a < 2; ++a) {
          s.sg[a] = 0;
          for (int c = 0; c < 3; ++c) s.sg[a] += s.basis[c][a] * s.gradient[c];
          for (int b = 0; b < 2; ++b) {
            double acc = 0;
            for (int c = 0; c < 3; ++c)
              for (int d = 0; d < 3; ++d) acc += u[a][c] * s.JtJ[3 * c + d] * u[b][d];
            s.sB[2 * a + b] = acc;
          }
        }
      }
    }
  }
  const double gnn = norm3(s.gn);
  if (gnn <= s.radius) {
    for (int c = 0; c < 3; ++c) step[c] = s.gn[c] / s.diagonal[c];
    s.step_norm = gnn;
    return true;
  }
  if (s.one_dim) {
    const double gn_ = norm3(s.gradient);
    for (int c = 0; c < 3; ++c) step[c] = -(s.radius / gn_) * s.gradient[c] / s.diagonal[c];
    s.step_norm = s.radius;
    return true;
  }
  double y2[2];
  min_on_circle(s.sB, s.sg, s.radius, y2);
  if (!isfinite(y2[0]) || !isfinite(y2[1])) {
    dogleg_traditional(s, step);
    return true;
  }
  for (int c = 0; c < 3; ++c) step[c] = (s.basis[c][0] * y2[0] + s.basis[c][1] * y2[1]) / s.diagonal[c];
  s.step_norm = s.radius;
  return true;
}

// shared control block written by thread 0, read by everyone
struct Control {
  int action;          // 0: stop, 1: evaluate candidate, 2: retry step without evaluation
  double cand[3];
  double x[3];
};

struct SolveShared {
  Dog dog;
  Control ctl;
  double red[kNW * 10 + 10];
  double scale[3];
  double x_cost, x_norm, model_cost_change;
  double grad[3];
  int iter, invalid;
};

// TrustRegionMinimizer::Minimize for 3 parameters.  x in/out (all threads hold the same copy).
__device__ int trust_region_minimize(const Problem& q, double x[3], double& final_cost, int max_iter,
                                     SolveShared& S) {
  const bool t0 = threadIdx.x == 0;
  if (q.n == 0) {
    final_cost = 0.0;
    return 0;
  }
  double sums[10];
  evaluate<true>(q, x, S.red, sums);
  if (t0) {
    S.dog.radius = 1e4;
    S.dog.mu = 1e-8;
    S.dog.reuse = 0;
    S.x_cost = sums[0];
    S.x_norm = norm3(x);
    for (int c = 0; c < 3; ++c) S.grad[c] = sums[1 + c];
    // jacobi scaling from the initial Jacobian, kept for the whole solve
    S.scale[0] = 1.0 / (1.0 + sqrt(sums[4]));
    S.scale[1] = 1.0 / (1.0 + sqrt(sums[7]));
    S.scale[2] = 1.0 / (1.0 + sqrt(sums[9]));
    S.iter = 0;
    S.invalid = 0;
  }
  bool have_fresh = true;   // sums hold the evaluation at the current x
  for (;;) {
    if (t0) {
      Control& C = S.ctl;
      C.action = 0;
      if (have_fresh) {
        const double* sc = S.scale;
        const double u[6] = {sums[4], sums[5], sums[6], sums[7], sums[8], sums[9]};
        S.dog.JtJ[0] = u[0] * sc[0] * sc[0];
        S.dog.JtJ[1] = S.dog.JtJ[3] = u[1] * sc[0] * sc[1];
        S.dog.JtJ[2] = S.dog.JtJ[6] = u[2] * sc[0] * sc[2];
        S.dog.JtJ[4] = u[3] * sc[1] * sc[1];
        S.dog.JtJ[5] = S.dog.JtJ[7] = u[4] * sc[1] * sc[2];
        S.dog.JtJ[8] = u[5] * sc[2] * sc[2];
        for (int c = 0; c < 3; ++c) S.dog.Jtr[c] = S.grad[c] * sc[c];
      }
      // FinalizeIterationAndCheckIfMinimizerCanContinue
      const double gmax = fmax(fabs(S.grad[0]), fmax(fabs(S.grad[1]), fabs(S.grad[2])));
      bool go = S.iter < max_iter && !(gmax <= 1e-10) && !(S.dog.radius <= 1e-32);
      while (go) {
        ++S.iter;
        double step[3];
        bool valid = dogleg_compute_step(S.dog, step);
        double mcc = 0;
        if (valid) {
          // model_cost_change = -(J step)^T (r + J step / 2) = -(g^T step + step^T JtJ step / 2)
          double gs = 0, sBs = 0;
          for (int a = 0; a < 3; ++a) {
            gs += S.dog.Jtr[a] * step[a];
            for (int b = 0; b < 3; ++b) sBs += step[a] * S.dog.JtJ[3 * a + b] * step[b];
          }
          mcc = -(gs + 0.5 * sBs);
          valid = mcc > 0.0;
        }
        if (!valid) {
          if (++S.invalid >= 5) {
            go = false;
            break;
          }
          S.dog.mu *= 10.0;
          S.dog.reuse = 0;
          if (!(S.iter < max_iter)) go = false;
          continue;
        }
        S.invalid = 0;
        S.model_cost_change = mcc;
        for (int c = 0; c < 3; ++c) C.cand[c] = x[c] + step[c] * S.scale[c];
        C.action = 1;
        break;
      }
    }
    __syncthreads();
    if (S.ctl.action == 0) break;
    double cand[3] = {S.ctl.cand[0], S.ctl.cand[1], S.ctl.cand[2]};
    double cs[10];
    evaluate<false>(q, cand, S.red, cs);
    const double cand_cost = cs[0];
    // decisions are pure functions of shared values: every thread takes the same branch
    const double dx0 = x[0] - cand[0], dx1 = x[1] - cand[1], dx2 = x[2] - cand[2];
    const double step_norm = sqrt(dx0 * dx0 + dx1 * dx1 + dx2 * dx2);
    const double x_cost = S.x_cost, x_norm = S.x_norm, mcc = S.model_cost_change;
    __syncthreads();
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) break;            // ParameterToleranceReached
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * x_cost) break;             // FunctionToleranceReached
    const double rel = cost_change / mcc;
    if (rel > 1e-3) {                                          // HandleSuccessfulStep
      x[0] = cand[0];
      x[1] = cand[1];
      x[2] = cand[2];
      evaluate<true>(q, x, S.red, sums);
      have_fresh = true;
      if (t0) {
        S.x_cost = sums[0];
        S.x_norm = norm3(x);
        for (int c = 0; c < 3; ++c) S.grad[c] = sums[1 + c];
        if (rel < 0.25) S.dog.radius *= 0.5;
        if (rel > 0.75) S.dog.radius = fmax(S.dog.radius, 3.0 * S.dog.step_norm);
        if (S.dog.radius > 1e16) S.dog.radius = 1e16;
        S.dog.mu = fmax(1e-8, 2.0 * S.dog.mu / 10.0);
        S.dog.reuse = 0;
      }
    } else {                                                   // StepRejected
      have_fresh = false;
      if (t0) {
        S.dog.radius *= 0.5;
        S.dog.reuse = 1;
      }
    }
  }
  __syncthreads();
  final_cost = S.x_cost;
  const int it = S.iter;
  __syncthreads();
  return it;
}

__device__ double cost_only(const Problem& q, const double x[3], SolveShared& S) {
  double cs[10];
  evaluate<false>(q, x, S.red, cs);
  const double v = cs[0];
  __syncthreads();
  return v;
}

// ------------------------------------------------------------------ corners (getPCDcorners)
__device__ __forceinline__ void inv_rigid_apply(const float* T, const float in[3], float out[3]) {
  const float dx = in[0] - T[3], dy = in[1] - T[7], dz = in[2] - T[11];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = T[0 + c] * dx;
    s = s + T[4 + c] * dy;
    s = s + T[8 + c] * dz;
    out[c] = s;
  }
}

__global__ __launch_bounds__(kSolveThreads) void k7_refine_corners(Ctx c) {
  __shared__ SolveShared S;
  __shared__ float s_T[16];
  __shared__ uint32_t s_pick[4];
  const uint32_t f = blockIdx.x;
  ilcc_result* r = &c.res[f];
  if (r->status != ILCC_OK) return;
  const uint64_t beg = c.off[f];
  const uint32_t tid = threadIdx.x;

  Problem q;
  q.yz = c.yz + beg;
  q.lab = c.lab + beg;
  q.n = c.n_lab[f];
  q.bd = Board{(double)c.p.board_w, (double)c.p.board_h, c.p.grid_length, c.p.huber_delta};

  double start[3] = {0.0, 0.0, 0.0};
  int ph_lo = 0, ph_hi = 0;
  if (c.p.solver == ILCC_SOLVER_GRID) {
    // argmin over this frame's K6 partials: cost, then index distance to zero, then flat index
    if (tid == 0) {
      const GridPartial* gp = c.partial + (uint64_t)f * c.grid_blocks;
      GridPartial b = gp[0];
      for (uint32_t k = 1; k < c.grid_blocks; ++k) {
        const GridPartial t = gp[k];
        if (t.cost < b.cost || (t.cost == b.cost && (t.d2 < b.d2 || (t.d2 == b.d2 && t.flat < b.flat)))) b = t;
      }
      s_pick[0] = b.flat;
      r->grid_index = (int32_t)b.flat;
      r->grid_cost = b.cost;
    }
    __syncthreads();
    const uint32_t flat = s_pick[0];
    if (flat == 0xFFFFFFFFu) {
      if (tid == 0) r->status = ILCC_BAD_ARGUMENT;
      return;
    }
    const uint32_t cell = flat >> 1;
    const uint32_t bz = cell % (uint32_t)c.p.n_tz, ay = (cell / (uint32_t)c.p.n_tz) % (uint32_t)c.p.n_ty,
                   k = cell / ((uint32_t)c.p.n_tz * (uint32_t)c.p.n_ty);
    start[0] = c.p.th_min + k * c.p.th_step;
    start[1] = c.p.ty_min + ay * c.p.ty_step;
    start[2] = c.p.tz_min + bz * c.p.tz_step;
    ph_lo = ph_hi = (int)(flat & 1u);
  } else {
    if (c.p.phase_mode == 2) {
      ph_lo = 0;
      ph_hi = 1;
    } else {
      ph_lo = ph_hi = (c.p.phase_mode == 1) ? 1 : 0;
    }
  }

  double best_sel = 1.7976931348623157e308;
  double bx[3] = {0, 0, 0}, bca = 0, bcb = 0;
  int bph = ph_lo, bia = 0, bib = 0;
  for (int ph = ph_lo; ph <= ph_hi; ++ph) {
    double x[3] = {start[0], start[1], start[2]};
    double ca = 0, cb = 0;
    q.tlw = ph != 0;
    q.oob = true;    // pass A (LidarCornersEst.cpp:403-405)
    const int ia = trust_region_minimize(q, x, ca, c.p.max_iterations, S);
    q.oob = false;   // pass B (:406-408)
    const int ib = trust_region_minimize(q, x, cb, c.p.max_iterations, S);
    q.oob = true;
    const double sel = cost_only(q, x, S);
    if (sel < best_sel) {
      best_sel = sel;
      bx[0] = x[0];
      bx[1] = x[1];
      bx[2] = x[2];
      bca = ca;
      bcb = cb;
      bph = ph;
      bia = ia;
      bib = ib;
    }
  }

  // transf = pcl::getTransformation(0, ty, tz, theta, 0, 0): float Affine3f (:412)
  if (tid == 0) {
    r->theta_t[0] = bx[0];
    r->theta_t[1] = bx[1];
    r->theta_t[2] = bx[2];
    r->cost_a = bca;
    r->cost_b = bcb;
    r->sel_cost = best_sel;
    r->phase = bph;
    r->iters_a = bia;
    r->iters_b = bib;
    const float roll = (float)bx[0];
    const float E = cosf(roll), F = sinf(roll);
    const float T[16] = {1, 0, 0, 0, 0, E, -F, (float)bx[1], 0, F, E, (float)bx[2], 0, 0, 0, 1};
    for (int k = 0; k < 16; ++k) s_T[k] = T[k];
  }
  __syncthreads();

  const int W = c.p.board_w, H = c.p.board_h;
  const int nc = (W - 1) * (H - 1);
  const int ncc = nc < ILCC_MAX_CORNERS ? nc : ILCC_MAX_CORNERS;
  for (int t = (int)tid; t < ncc; t += kSolveThreads) {
    const int i = 1 + t / (H - 1), j = 1 + t % (H - 1);          // :513-534
    const double xg = (i - (double)W / 2.0) * c.p.grid_length;
    const double yg = (j - (double)H / 2.0) * c.p.grid_length;
    const float pt[3] = {0.0f, (float)xg, (float)yg};
    float a[3], w[3];
    inv_rigid_apply(s_T, pt, a);        // transOptim.inverse() :548
    inv_rigid_apply(r->pca, a, w);      // transPCA.inverse()   :549
    r->corners[3 * t] = w[0];
    r->corners[3 * t + 1] = w[1];
    r->corners[3 * t + 2] = w[2];
  }
  if (tid == 0) r->n_corners = ncc;

  // m_cloud_optim = transf * m_cloud_PCA (:413), float
  const uint32_t M = (uint32_t)r->n_plane;
  const float4* __restrict__ Q = c.pca + beg;
  float4* __restrict__ O = c.optim + beg;
  for (uint32_t i = tid; i < M; i += kSolveThreads) {
    const float4 v = Q[i];
    float o[3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      float s = s_T[4 * rr] * v.x;
      s = s + s_T[4 * rr + 1] * v.y;
      s = s + s_T[4 * rr + 2] * v.z;
      s = s + s_T[4 * rr + 3];
      o[rr] = s;
    }
    O[i] = make_float4(o[0], o[1], o[2], v.w);
  }
}

// test entry: one solve on frame 0's labelled points
__global__ __launch_bounds__(kSolveThreads) void k7_local_solve(Ctx c, int tlw, int use_oob,
                                                                double* theta_t, double* cost_iters) {
  __shared__ SolveShared S;
  Problem q;
  q.yz = c.yz;
  q.lab = c.lab;
  q.n = c.n_lab[0];
  q.bd = Board{(double)c.p.board_w, (double)c.p.board_h, c.p.grid_length, c.p.huber_delta};
  q.tlw = tlw != 0;
  q.oob = use_oob != 0;
  double x[3] = {theta_t[0], theta_t[1], theta_t[2]};
  __syncthreads();
  double cost = 0;
  const int it = trust_region_minimize(q, x, cost, c.p.max_iterations, S);
  if (threadIdx.x == 0) {
    theta_t[0] = x[0];
    theta_t[1] = x[1];
    theta_t[2] = x[2];
    cost_iters[0] = cost;
    cost_iters[1] = (double)it;
  }
}

void launch_refine_corners(const Ctx& c, hipStream_t s) {
  hipLaunchKernelGGL(k7_refine_corners, dim3(c.n_frames), dim3(kSolveThreads), 0, s, c);
}

void launch_local_solve(const Ctx& c, hipStream_t s, int32_t tlw, int32_t use_oob, double* theta_t,
                        double* cost_iters) {
  hipLaunchKernelGGL(k7_local_solve, dim3(1), dim3(kSolveThreads), 0, s, c, tlw, use_oob, theta_t,
                     cost_iters);
}

}  // namespace ilcc

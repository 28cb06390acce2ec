// K7 refine_and_corners.
//
// K7a local_solve (ILCC_SOLVER_REFERENCE_LOCAL): ONE WAVEFRONT per (frame, colour phase); a 256-thread workgroup holds
//      kSolveWaves such solves (the two phases of a frame share one staged copy of its labelled points in LDS).
//  (1) starts at (0,0,0), the reference's own start;
//  (2) runs the reference's two local solves, pass A (useOutofBoard = true) then pass B (false)
//      -- LidarCornersEst::get_corners, /root/reference/ilcc2/src/LidarCornersEst.cpp:398-409 --
//      each a restatement of what ceres::Solve does for Optimization::get_theta_t
//      (/root/reference/ilcc2/src/Optimization.cpp:94-160): TRUST_REGION, DOGLEG/SUBSPACE_DOGLEG,
//      DENSE_NORMAL_CHOLESKY, HuberLoss(0.1) through Ceres' Corrector, Jacobi scaling, Ceres 1.14
//      default tolerances.  The 64 lanes stride over the points (residual + Jacobian in double), sums are
//      combined with a DPP / permlane butterfly (no LDS, no barrier: after the staging barrier the wavefronts of a
//      workgroup never meet again), and the 3-parameter trust-region bookkeeping runs once per wavefront on
//      wave-uniform values.  Round 6 (profiles/r06a_*): the 4-wavefront-per-solve layout of rounds 1-5 issued the
//      dogleg FOUR times per iteration -- 60 % of the kernel's 1.02 G wave-instructions per 1024 frames -- and at 203
//      VGPRs held two wavefronts per SIMD; the kernel was VALU-issue bound (VALU busy 70 %), not latency bound.
//      The arithmetic is unchanged value for value: divisions by a divisor that is used many times (g; the dogleg's
//      diagonal, Cholesky pivots, norms) are one true division for RN(1/d) plus Markstein's correction (exactly the
//      correctly rounded quotient), sqrt(r * r) of Huber is |r| (exact in binary floating point).
// K7r pattern_refine (ILCC_SOLVER_GRID): one 192-thread workgroup (three wavefronts, one per theta of the stencil) per frame.  Starts at the K6 grid argmin
//      (near ties of the fp32 grid pass are first re-ordered on exact fixed-point costs), then a monotone
//      pattern search on the pass-A cost and a check of the eight neighbouring basins -- the same
//      specification as the oracle's orc_pattern_refine, bit for bit: every point's term is computed in
//      fp64 exactly like the oracle's, rounded to a multiple of 2^-40 and summed as an INTEGER, so the
//      parallel reduction cannot change a single decision.
// K7b corners: picks the phase with the lower with-OOB cost, then builds the corner lattice:
//      LidarCornersEst::getPCDcorners (:501-556) with
//      transf = pcl::getTransformation(0, ty, tz, theta, 0, 0) (:412), and the display cloud
//      m_cloud_optim (:413).
#include "ilcc_internal.h"

namespace ilcc {

// ------------------------------------------------------------------ residual (Optimization.h:31-107)
// a / d, correctly rounded, for a divisor that divides many numerators: y = RN(1 / d) costs one true division, every
// quotient after that is q = RN(a y); r = a - q d (exact, one FMA); RN(q + r y) -- Markstein's theorem: with y the
// correctly rounded reciprocal and q within an ulp of a / d the corrected quotient IS RN(a / d) (the sequence the
// hardware's own v_div_* expansion ends with; tools/ubench/markstein_check.c compares it with `/` on 6e8 samples).
// Numerators here are finite and far from the over/underflow range (board coordinates, trust-region bookkeeping).
struct Divisor {
  double d, y;
};
__device__ __forceinline__ Divisor make_divisor(double d) { return Divisor{d, 1.0 / d}; }
__device__ __forceinline__ double div_by(double a, const Divisor& v) {
  const double q = a * v.y;
  const double r = __builtin_fma(-q, v.d, a);
  return __builtin_fma(r, v.y, q);
}

struct Board {   // K7r / K7b
  double W, H, g, delta;
};
// K7a: the same board with the constants the residual needs
struct SolveBoard {
  double W, H, delta;
  double Wg2, Hg2;   // W * g / 2.0, H * g / 2.0 (Optimization.h:45-46)
  Divisor g;
};
__device__ __forceinline__ SolveBoard make_board(const ilcc_params& p) {
  SolveBoard b;
  b.W = (double)p.board_w;
  b.H = (double)p.board_h;
  b.delta = p.huber_delta;
  b.Wg2 = b.W * p.grid_length / 2.0;
  b.Hg2 = b.H * p.grid_length / 2.0;
  b.g = make_divisor(p.grid_length);
  return b;
}

// raw residual; jac = d r / d(theta, ty, tz) when JAC.  cs = (cos theta, sin theta).  Value for value the oracle's
// residual_cs (oracle/ilcc_oracle.c): (floor(i) even) is read off the integer instead of floor(ifl / 2) * 2 == ifl,
// ceil(i) of a non-integer i is floor(i) + 1, d i / d theta = -(s y + c z) / g = -rz / g and d j / d theta = ry / g
// reuse the rotated point (the same products, the same sums), +-1 / g is +-RN(1 / g).
template <bool JAC>
__device__ __forceinline__ double residual(const double x[3], double c, double s, double y, double z,
                                           const SolveBoard& bd, bool tlw, bool laser_white, bool use_oob,
                                           double jac[3]) {
  // written without a branch (round 6): a divergent `if` costs the wavefront both sides plus the exec-mask bookkeeping.
  // min(frac, 1 - frac) IS the in-board distance of :70-78 -- frac = i - floor(i) is exact, and for frac > 1/2 both ceil(i) - i and
  // 1 - frac are exact (Sterbenz) and equal; min(|i|, |i - W|) IS the out-of-board distance of :86-97
  const double ry = c * y - s * z;
  const double rz = s * y + c * z;
  const double i = div_by((ry + x[1]) + bd.Wg2, bd.g);
  const double j = div_by((rz + x[2]) + bd.Hg2, bd.g);
  const bool inside = (int)(i > 0) & (int)(i < bd.W) & (int)(j > 0) & (int)(j < bd.H);
  const double ifl = floor(i), jfl = floor(j);
  const double fi = i - ifl, fj = j - jfl;
  const double res_in = fmin(fi, 1.0 - fi) + fmin(fj, 1.0 - fj);
  const bool odd = ((((int)ifl) ^ ((int)jfl)) & 1) != 0;
  const bool white = odd != tlw;   // same parity: topleftWhite, else its opposite (:57-61)
  const double iw = i - bd.W, jh = j - bd.H;
  double res_out = 0.0;
  if (use_oob) res_out = fmin(fabs(i), fabs(iw)) + fmin(fabs(j), fabs(jh));   // (wave-uniform: pass B never computes it)
  const bool take_in = (int)inside & (int)(laser_white != white), take_out = (int)!inside & (int)use_oob;
  if (JAC) {
    // d r / d i, d r / d j: -1 past the middle of a cell; out of board the sign of the nearer edge's offset (:86-97)
    const double si_in = fi > 0.5 ? -1.0 : 1.0, sj_in = fj > 0.5 ? -1.0 : 1.0;
    const double si_out = ((fabs(i) < fabs(iw) ? i : iw) < 0) ? -1.0 : 1.0, sj_out = ((fabs(j) < fabs(jh) ? j : jh) < 0) ? -1.0 : 1.0;
    const double si = take_in ? si_in : (take_out ? si_out : 0.0), sj = take_in ? sj_in : (take_out ? sj_out : 0.0);
    const double dith = -div_by(rz, bd.g), djth = div_by(ry, bd.g);
    jac[0] = si * dith + sj * djth;
    jac[1] = si * bd.g.y;
    jac[2] = sj * bd.g.y;
  }
  return take_in ? res_in : (take_out ? res_out : 0.0);
}

// HuberLoss(a) on s = r * r with r >= 0 (the residual is a sum of distances): sqrt(s) is r itself -- for binary floating point
// sqrt(RN(r * r)) == |r| barring over/underflow (Boldo 2015; the oracle calls sqrt) -- so no square root is taken
__device__ __forceinline__ void huber(double a, double r, double s, double& rho0, double& rho1) {
  const double b = a * a;
  const bool outlier = s > b;
  rho0 = outlier ? 2.0 * a * r - b : s;
  rho1 = outlier ? fmax(a / r, 2.2250738585072014e-308) : 1.0;   // (only the Jacobian pass reads rho1: the cost pass drops the division)
}

// ------------------------------------------------------------------ wavefront-wide evaluation
constexpr int kSolveWaves = kSolveThreads / ILCC_WAVE;   // solves per K7a workgroup
constexpr int kSolveLdsMax = 144 * 1024;                 // K7a: dynamic LDS bound (a CU has 160 KB)
#ifndef ILCC_K7A_WIDE_MAX
#define ILCC_K7A_WIDE_MAX 256
#endif
constexpr int kSolveWideMaxFrames = ILCC_K7A_WIDE_MAX;    // K7a: batches up to this size give every solve a whole workgroup

struct Problem {
  const float2* yz;      // LDS or global
  const uint8_t* lab;
  uint32_t n;
  SolveBoard bd;
  bool tlw, oob;
  double* red;           // WIDE only -- LDS: 2 x kSolveRedDoubles (double-buffered partial sums)
  int* flip;             // WIDE only -- per-thread toggle (register copy lives in the caller)
};

// v[lane ^ MASK] for a 64-bit value, in registers only: v_permlane32_swap / v_permlane16_swap (gfx950) and
// DPP row rotations / quad permutes on the two dwords -- no ds_bpermute round trips.
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, BANK, false);
}
template <int MASK>
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t v) {
  if constexpr (MASK == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane_id() & 32) ? r[0] : r[1];   // swap exchanges the upper half of operand 0 with the lower half of operand 1
  } else if constexpr (MASK == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (lane_id() & 16) ? r[0] : r[1];
  } else if constexpr (MASK == 8) {
    return dpp_u32<0x128>(0u, v);                       // row_ror:8
  } else if constexpr (MASK == 4) {
    const uint32_t lo = dpp_u32<0x124, 0xa>(0u, v);     // row_ror:4 -> banks 1,3 take lane i-4
    return dpp_u32<0x12C, 0x5>(lo, v);                  // row_ror:12 -> banks 0,2 take lane i+4
  } else if constexpr (MASK == 2) {
    return dpp_u32<0x4E>(0u, v);                        // quad_perm [2,3,0,1]
  } else {
    return dpp_u32<0xB1>(0u, v);                        // quad_perm [1,0,3,2]
  }
}
template <int MASK>
__device__ __forceinline__ double xor_lane_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = xor_lane_u32<MASK>((uint32_t)b), hi = xor_lane_u32<MASK>((uint32_t)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int MASK>
__device__ __forceinline__ unsigned long long xor_lane_u64(unsigned long long b) {
  const uint32_t lo = xor_lane_u32<MASK>((uint32_t)b), hi = xor_lane_u32<MASK>((uint32_t)(b >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// butterfly in the order 32, 16, 8, 4, 2, 1: identical in every lane (each step adds the same two operands
// in both partners)
__device__ __forceinline__ double wave_allsum(double v) {
  v += xor_lane_f64<32>(v);
  v += xor_lane_f64<16>(v);
  v += xor_lane_f64<8>(v);
  v += xor_lane_f64<4>(v);
  v += xor_lane_f64<2>(v);
  v += xor_lane_f64<1>(v);
  return v;
}

// sums[0] = cost ; if JAC: sums[1..3] = J^T r, sums[4..9] = upper J^T J (00,01,02,11,12,22),
// with Ceres' Corrector applied (rows scaled by sqrt(rho')).  The same value in every lane on return.
// ONE summation order for both layouts, so that a frame's result does not depend on the size of the batch it came in:
// lane l owns the points l, l + 64, l + 128, ...; the point l + 64 m goes to the lane's partial sum a[m mod 4] (each
// partial adds its points in increasing order); the lane's sum is ((a0 + a1) + a2) + a3; the 64 lane sums are
// combined by the butterfly.
//   WIDE = false: one wavefront per solve.  The cost alone keeps the four partials in registers (one walk over the
//   points); the Jacobian pass (one evaluation in six) walks the four residue classes one after the other.
//   WIDE = true (small batches, where latency counts and the chip is not full): the whole 256-thread workgroup works
//   on ONE solve, wavefront w computes the partials a[w]; wavefronts -> LDS -> everyone, ONE barrier per call
//   (double-buffered slots); every thread then runs the trust-region bookkeeping redundantly.
template <bool JAC>
__device__ __forceinline__ void add_point(const Problem& q, const double x[3], double cs, double sn, uint32_t p, double acc[10]) {
  const float2 v = q.yz[p];
  double jac[3];
  const double res = residual<JAC>(x, cs, sn, (double)v.x, (double)v.y, q.bd, q.tlw, q.lab[p] != 0, q.oob, jac);
  double r0, r1;
  huber(q.bd.delta, res, res * res, r0, r1);
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
// the partial sums a[w] of this lane: points first, first + 256, ...
template <bool JAC>
__device__ __forceinline__ void accumulate_class(const Problem& q, const double x[3], double cs, double sn, uint32_t first,
                                                 double acc[10]) {
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  // (the Jacobian pass is one evaluation in six and carries 10 sums: not unrolled, its registers set the kernel's occupancy)
#pragma unroll 1
  for (uint32_t p = first; p < q.n; p += (uint32_t)(4 * ILCC_WAVE)) add_point<JAC>(q, x, cs, sn, p, acc);
}

constexpr int kSolveRedDoubles = 10 * 4 * ILCC_WAVE;   // WIDE: one exchange buffer (10 sums x 4 partials x 64 lanes)

template <bool JAC, bool WIDE>
__device__ __forceinline__ void evaluate(const Problem& q, const double x[3], double sums[10]) {
  static_assert(kSolveThreads == 4 * ILCC_WAVE, "the summation order is defined on four partial sums per lane");
  double sn, cs;
  sincos(x[0], &sn, &cs);
  constexpr int NV = JAC ? 10 : 1;
  const uint32_t lane = (uint32_t)lane_id();
  double tot[10];
  if (WIDE) {
    double acc[10];
    accumulate_class<JAC>(q, x, cs, sn, threadIdx.x, acc);
    *q.flip ^= 1;
    double* slot = q.red + (*q.flip) * kSolveRedDoubles;
#pragma unroll
    for (int k = 0; k < NV; ++k) slot[(k * 4 + wave_id()) * ILCC_WAVE + lane] = acc[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double* sk = slot + k * 4 * ILCC_WAVE + lane;
      tot[k] = ((sk[0] + sk[ILCC_WAVE]) + sk[2 * ILCC_WAVE]) + sk[3 * ILCC_WAVE];
    }
  } else if (JAC) {
#pragma nounroll
    for (int w = 0; w < 4; ++w) {
      double acc[10];
      accumulate_class<true>(q, x, cs, sn, (uint32_t)w * ILCC_WAVE + lane, acc);
#pragma unroll
      for (int k = 0; k < NV; ++k) tot[k] = (w == 0) ? acc[k] : tot[k] + acc[k];
    }
  } else {
    double a0[10], a1[10], a2[10], a3[10];   // ([0] only: the cost)
    a0[0] = a1[0] = a2[0] = a3[0] = 0.0;
    for (uint32_t p = lane; p < q.n; p += (uint32_t)(4 * ILCC_WAVE)) {
      add_point<false>(q, x, cs, sn, p, a0);
      if (p + ILCC_WAVE < q.n) add_point<false>(q, x, cs, sn, p + ILCC_WAVE, a1);
      if (p + 2 * ILCC_WAVE < q.n) add_point<false>(q, x, cs, sn, p + 2 * ILCC_WAVE, a2);
      if (p + 3 * ILCC_WAVE < q.n) add_point<false>(q, x, cs, sn, p + 3 * ILCC_WAVE, a3);
    }
    tot[0] = ((a0[0] + a1[0]) + a2[0]) + a3[0];
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) sums[k] = wave_allsum(tot[k]);
}

// ------------------------------------------------------------------ dogleg bookkeeping (wave-uniform values)
struct Dog {
  double radius, mu;
  int reuse;
  Divisor d0, d1, d2;       // diagonal (divides the gradient, the basis and every step)
  double g0, g1, g2;        // scaled gradient
  double n0, n1, n2;        // Gauss-Newton step (scaled space)
  double alpha, step_norm;
  int one_dim;
  double b00, b01, b10, b11, b20, b21;   // subspace basis (3x2)
  double sg0, sg1, sB00, sB01, sB11;
  double A00, A01, A02, A11, A12, A22;   // J^T J of the column-scaled Jacobian
  double r0, r1, r2;                     // J^T r of the column-scaled Jacobian
  // min_on_circle's radius-independent half (eigen-decomposition of the 2x2 subspace model), computed at the first
  // boundary step of a linearisation point and kept while the rejected steps only shrink the radius
  int eig_ready;
  double e_l1, e_l2, e_v1x, e_v1y, e_v2x, e_v2y, e_g1, e_g2, e_gn;
};

// (A + diag(e)) x = b by Cholesky; false on a non-positive pivot (Eigen LLT NumericalIssue)
__device__ __forceinline__ bool chol3_solve(double a00, double a01, double a02, double a11, double a12,
                                            double a22, double b0, double b1, double b2, double& x0,
                                            double& x1, double& x2) {
  if (!(a00 > 0.0)) return false;
  const Divisor l00 = make_divisor(sqrt(a00));
  const double l10 = div_by(a01, l00), l20 = div_by(a02, l00);
  const double s11 = a11 - l10 * l10;
  if (!(s11 > 0.0)) return false;
  const Divisor l11 = make_divisor(sqrt(s11));
  const double l21 = div_by(a12 - l20 * l10, l11);
  const double s22 = a22 - l20 * l20 - l21 * l21;
  if (!(s22 > 0.0)) return false;
  const Divisor l22 = make_divisor(sqrt(s22));
  const double y0 = div_by(b0, l00);
  const double y1 = div_by(b1 - l10 * y0, l11);
  const double y2 = div_by(b2 - l20 * y0 - l21 * y1, l22);
  x2 = div_by(y2, l22);
  x1 = div_by(y1 - l21 * x2, l11);
  x0 = div_by(y0 - l10 * x1 - l20 * x2, l00);
  return isfinite(x0) && isfinite(x1) && isfinite(x2);
}

// argmin of 1/2 y'By + g'y on |y| = radius, B symmetric PSD 2x2 (Ceres: quartic roots; here
// eigen-decomposition + Newton on the secular equation, More-Sorensen; oracle/ilcc_oracle.c min_on_circle).
// First half: everything that does not depend on the radius.
__device__ __forceinline__ void circle_eigen(Dog& s) {
  const double B00 = s.sB00, B01 = s.sB01, B11 = s.sB11, gx = s.sg0, gy = s.sg1;
  const double d = 0.5 * (B00 - B11), e = B01;
  const double h = sqrt(d * d + e * e), mean = 0.5 * (B00 + B11);
  s.e_l1 = mean - h;
  s.e_l2 = mean + h;
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
  s.e_v1x = v1x;
  s.e_v1y = v1y;
  s.e_v2x = v2x;
  s.e_v2y = v2y;
  s.e_g1 = v1x * gx + v1y * gy;
  s.e_g2 = v2x * gx + v2y * gy;
  s.e_gn = sqrt(s.e_g1 * s.e_g1 + s.e_g2 * s.e_g2);
  s.eig_ready = 1;
}
// Second half: the multiplier for this radius
__device__ __forceinline__ void min_on_circle(Dog& s, double radius, double& yx, double& yy) {
  if (!s.eig_ready) circle_eigen(s);
  const double l1 = s.e_l1, l2 = s.e_l2, g1 = s.e_g1, g2 = s.e_g2;
  const Divisor rad = make_divisor(radius);
  const double gnr = div_by(s.e_gn, rad);
  double lo = fmax(0.0, -l1);
  lo = fmax(lo, gnr - l2);
  const double hi = gnr - l1;
  double lam = lo;
  if (!(l1 + lam > 0.0)) lam = lo + 1e-12 * fmax(1.0, fabs(hi));
  for (int it = 0; it < 60; ++it) {
    const double a1 = l1 + lam, a2 = l2 + lam;
    const double y1 = -g1 / a1, y2 = -g2 / a2;
    const double ny = sqrt(y1 * y1 + y2 * y2);
    const double qq = g1 * g1 / (a1 * a1 * a1) + g2 * g2 / (a2 * a2 * a2);
    if (!(qq > 0.0) || !isfinite(ny)) break;
    const double dl = (ny * ny / qq) * div_by(ny - radius, rad);
    double nl = lam + dl;
    if (!(l1 + nl > 0.0)) nl = 0.5 * (lam + fmax(0.0, -l1));
    if (fabs(nl - lam) <= 1e-15 * fmax(1.0, fabs(nl))) {
      lam = nl;
      break;
    }
    lam = nl;
  }
  const double a1 = l1 + lam, a2 = l2 + lam;
  double y1 = (a1 > 0.0) ? -g1 / a1 : 0.0, y2 = (a2 > 0.0) ? -g2 / a2 : 0.0;
  double ny = sqrt(y1 * y1 + y2 * y2);
  if (ny < radius * (1.0 - 1e-9) && !(a1 > 1e-300 * fmax(1.0, l2))) {
    y1 = sqrt(fmax(0.0, radius * radius - y2 * y2));
    ny = radius;
  }
  if (ny > 0.0) {
    const double k = radius / ny;
    y1 *= k;
    y2 *= k;
  }
  yx = s.e_v1x * y1 + s.e_v2x * y2;
  yy = s.e_v1y * y1 + s.e_v2y * y2;
}

__device__ __forceinline__ double nrm3(double a, double b, double c) { return sqrt(a * a + b * b + c * c); }

__device__ __forceinline__ void dogleg_traditional(Dog& s, double& s0, double& s1, double& s2) {
  const double gnn = nrm3(s.n0, s.n1, s.n2), gn_ = nrm3(s.g0, s.g1, s.g2);
  if (gnn <= s.radius) {
    s0 = div_by(s.n0, s.d0);
    s1 = div_by(s.n1, s.d1);
    s2 = div_by(s.n2, s.d2);
    s.step_norm = gnn;
    return;
  }
  if (gn_ * s.alpha >= s.radius) {
    const double k = -(s.radius / gn_);
    s0 = div_by(k * s.g0, s.d0);
    s1 = div_by(k * s.g1, s.d1);
    s2 = div_by(k * s.g2, s.d2);
    s.step_norm = s.radius;
    return;
  }
  const double a0 = -s.alpha * s.g0, a1 = -s.alpha * s.g1, a2 = -s.alpha * s.g2;
  const double bdota = a0 * s.n0 + a1 * s.n1 + a2 * s.n2;
  const double a2n = a0 * a0 + a1 * a1 + a2 * a2;
  const double bma2 = (s.n0 - a0) * (s.n0 - a0) + (s.n1 - a1) * (s.n1 - a1) + (s.n2 - a2) * (s.n2 - a2);
  const double cc = bdota - a2n;
  const double d = sqrt(cc * cc + bma2 * (s.radius * s.radius - a2n));
  const double beta = (cc <= 0) ? (d - cc) / bma2 : (s.radius * s.radius - a2n) / (d + cc);
  s0 = div_by(a0 + beta * (s.n0 - a0), s.d0);
  s1 = div_by(a1 + beta * (s.n1 - a1), s.d1);
  s2 = div_by(a2 + beta * (s.n2 - a2), s.d2);
  s.step_norm = s.radius;
}

// quadratic form u' A v with the symmetric 3x3 stored in Dog
__device__ __forceinline__ double qform(const Dog& s, double u0, double u1, double u2, double v0, double v1,
                                        double v2) {
  const double w0 = s.A00 * v0 + s.A01 * v1 + s.A02 * v2;
  const double w1 = s.A01 * v0 + s.A11 * v1 + s.A12 * v2;
  const double w2 = s.A02 * v0 + s.A12 * v1 + s.A22 * v2;
  return u0 * w0 + u1 * w1 + u2 * w2;
}

// DoglegStrategy::ComputeStep; A / r (scaled Jacobian) must be current when !reuse.
__device__ __forceinline__ bool dogleg_compute_step(Dog& s, double& s0, double& s1, double& s2) {
  if (!s.reuse) {
    s.reuse = 1;
    s.eig_ready = 0;
    s.d0 = make_divisor(sqrt(fmin(fmax(s.A00, 1e-6), 1e32)));
    s.d1 = make_divisor(sqrt(fmin(fmax(s.A11, 1e-6), 1e32)));
    s.d2 = make_divisor(sqrt(fmin(fmax(s.A22, 1e-6), 1e32)));
    s.g0 = div_by(s.r0, s.d0);
    s.g1 = div_by(s.r1, s.d1);
    s.g2 = div_by(s.r2, s.d2);
    {
      const double u0 = div_by(s.g0, s.d0), u1 = div_by(s.g1, s.d1), u2 = div_by(s.g2, s.d2);
      const double num = s.g0 * s.g0 + s.g1 * s.g1 + s.g2 * s.g2;
      s.alpha = num / qform(s, u0, u1, u2, u0, u1, u2);
    }
    bool ok = false;
    while (s.mu < 1.0) {
      const double sm = sqrt(s.mu);
      const double e0 = s.d0.d * sm, e1 = s.d1.d * sm, e2 = s.d2.d * sm;
      if (chol3_solve(s.A00 + e0 * e0, s.A01, s.A02, s.A11 + e1 * e1, s.A12, s.A22 + e2 * e2, s.r0, s.r1, s.r2,
                      s.n0, s.n1, s.n2)) {
        ok = true;
        break;
      }
      s.mu *= 10.0;
    }
    if (!ok) return false;
    s.n0 *= -s.d0.d;
    s.n1 *= -s.d1.d;
    s.n2 *= -s.d2.d;
    {
      const double q0 = s.g0 * s.g0 + s.g1 * s.g1 + s.g2 * s.g2;
      const double q1 = s.n0 * s.n0 + s.n1 * s.n1 + s.n2 * s.n2;
      const bool gfirst = q0 >= q1;
      const double nfv = sqrt(fmax(q0, q1));
      const Divisor nf = make_divisor(nfv);
      const double f0 = (gfirst ? s.g0 : s.n0), f1 = (gfirst ? s.g1 : s.n1), f2 = (gfirst ? s.g2 : s.n2);
      const double t0 = (gfirst ? s.n0 : s.g0), t1 = (gfirst ? s.n1 : s.g1), t2 = (gfirst ? s.n2 : s.g2);
      const double v00 = div_by(f0, nf), v01 = div_by(f1, nf), v02 = div_by(f2, nf);
      const double dot = t0 * v00 + t1 * v01 + t2 * v02;
      double v10 = t0 - dot * v00, v11 = t1 - dot * v01, v12 = t2 - dot * v02;
      const double nrv = nrm3(v10, v11, v12);
      s.one_dim = !(nrv > 3.0 * 2.220446049250313e-16 * nfv);
      if (!s.one_dim) {
        const Divisor nr = make_divisor(nrv);
        v10 = div_by(v10, nr);
        v11 = div_by(v11, nr);
        v12 = div_by(v12, nr);
        s.b00 = v00;
        s.b10 = v01;
        s.b20 = v02;
        s.b01 = v10;
        s.b11 = v11;
        s.b21 = v12;
        const double ua0 = div_by(v00, s.d0), ua1 = div_by(v01, s.d1), ua2 = div_by(v02, s.d2);
        const double ub0 = div_by(v10, s.d0), ub1 = div_by(v11, s.d1), ub2 = div_by(v12, s.d2);
        s.sg0 = v00 * s.g0 + v01 * s.g1 + v02 * s.g2;
        s.sg1 = v10 * s.g0 + v11 * s.g1 + v12 * s.g2;
        s.sB00 = qform(s, ua0, ua1, ua2, ua0, ua1, ua2);
        s.sB01 = qform(s, ua0, ua1, ua2, ub0, ub1, ub2);
        s.sB11 = qform(s, ub0, ub1, ub2, ub0, ub1, ub2);
      }
    }
  }
  const double gnn = nrm3(s.n0, s.n1, s.n2);
  if (gnn <= s.radius) {
    s0 = div_by(s.n0, s.d0);
    s1 = div_by(s.n1, s.d1);
    s2 = div_by(s.n2, s.d2);
    s.step_norm = gnn;
    return true;
  }
  if (s.one_dim) {
    const double k = -(s.radius / nrm3(s.g0, s.g1, s.g2));
    s0 = div_by(k * s.g0, s.d0);
    s1 = div_by(k * s.g1, s.d1);
    s2 = div_by(k * s.g2, s.d2);
    s.step_norm = s.radius;
    return true;
  }
  double yx, yy;
  min_on_circle(s, s.radius, yx, yy);
  if (!isfinite(yx) || !isfinite(yy)) {
    dogleg_traditional(s, s0, s1, s2);
    return true;
  }
  s0 = div_by(s.b00 * yx + s.b01 * yy, s.d0);
  s1 = div_by(s.b10 * yx + s.b11 * yy, s.d1);
  s2 = div_by(s.b20 * yx + s.b21 * yy, s.d2);
  s.step_norm = s.radius;
  return true;
}

// TrustRegionMinimizer::Minimize for 3 parameters, one wavefront, every lane on the same control flow.
#ifdef ILCC_K7_TIMING
__device__ unsigned long long g_k7_t[4];
#define K7_T0 const unsigned long long k7t0 = __builtin_readcyclecounter()
#define K7_ACC(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_k7_t[k] += __builtin_readcyclecounter() - k7t0; } while (0)
#else
#define K7_T0 do {} while (0)
#define K7_ACC(k) do {} while (0)
#endif

// One call site per evaluate<> flavour: the Jacobian pass of the start and of every accepted step is the `relinearise`
// block at the head of the loop (DoglegStrategy::StepAccepted's radius / mu updates do not read it, so running them
// first changes nothing) -- the kernel's code stays within the instruction cache.
template <bool WIDE>
__device__ __forceinline__ int trust_region_minimize(const Problem& q, double x[3], double& final_cost, int max_iter, Dog& s) {
  if (q.n == 0) {
    final_cost = 0.0;
    return 0;
  }
  double sums[10];
  // (s: this wavefront's dogleg state in LDS -- ~60 doubles that would otherwise stay live in VGPRs across every evaluation of
  // the points: 226 VGPRs, two wavefronts per SIMD; the wavefront reads them back, wave-uniform addresses, where the dogleg needs them)
  s.radius = 1e4;
  s.mu = 1e-8;
  s.reuse = 0;
  s.step_norm = 0.0;
  s.one_dim = 0;
  s.eig_ready = 0;
  double x_cost = 0.0, x_norm = 0.0;
  double gr0 = 0.0, gr1 = 0.0, gr2 = 0.0;
  double sc0 = 0.0, sc1 = 0.0, sc2 = 0.0;   // jacobi scaling from the initial Jacobian, kept for the whole solve
  int iter = 0, invalid = 0;
  bool relinearise = true, first = true;
  for (;;) {
    if (relinearise) {
      {
        K7_T0;
        evaluate<true, WIDE>(q, x, sums);
        K7_ACC(2);
      }
      x_cost = sums[0];
      x_norm = nrm3(x[0], x[1], x[2]);
      gr0 = sums[1];
      gr1 = sums[2];
      gr2 = sums[3];
      if (first) {
        sc0 = 1.0 / (1.0 + sqrt(sums[4]));
        sc1 = 1.0 / (1.0 + sqrt(sums[7]));
        sc2 = 1.0 / (1.0 + sqrt(sums[9]));
        first = false;
      }
      s.A00 = sums[4] * sc0 * sc0;
      s.A01 = sums[5] * sc0 * sc1;
      s.A02 = sums[6] * sc0 * sc2;
      s.A11 = sums[7] * sc1 * sc1;
      s.A12 = sums[8] * sc1 * sc2;
      s.A22 = sums[9] * sc2 * sc2;
      s.r0 = gr0 * sc0;
      s.r1 = gr1 * sc1;
      s.r2 = gr2 * sc2;
      relinearise = false;
    }
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iter >= max_iter) break;
    if (fmax(fabs(gr0), fmax(fabs(gr1), fabs(gr2))) <= 1e-10) break;
    if (s.radius <= 1e-32) break;
    ++iter;
    double st0 = 0, st1 = 0, st2 = 0;
    bool valid;
    {
      K7_T0;
      valid = dogleg_compute_step(s, st0, st1, st2);
      K7_ACC(0);
    }
    double mcc = 0;
    if (valid) {
      // model_cost_change = -(J step)'(r + J step/2) = -(g' step + step' JtJ step / 2)
      const double gs = s.r0 * st0 + s.r1 * st1 + s.r2 * st2;
      mcc = -(gs + 0.5 * qform(s, st0, st1, st2, st0, st1, st2));
      valid = mcc > 0.0;
    }
    if (!valid) {
      if (++invalid >= 5) break;
      s.mu *= 10.0;   // StepIsInvalid
      s.reuse = 0;
      continue;
    }
    invalid = 0;
    const double cand[3] = {x[0] + st0 * sc0, x[1] + st1 * sc1, x[2] + st2 * sc2};
    double cs[10];
    {
      K7_T0;
      evaluate<false, WIDE>(q, cand, cs);
      K7_ACC(1);
    }
    const double cand_cost = cs[0];
    const double step_norm = nrm3(x[0] - cand[0], x[1] - cand[1], x[2] - cand[2]);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) break;            // ParameterToleranceReached
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * x_cost) break;             // FunctionToleranceReached
    const double rel = cost_change / mcc;
    if (rel > 1e-3) {                                          // HandleSuccessfulStep
      x[0] = cand[0];
      x[1] = cand[1];
      x[2] = cand[2];
      relinearise = true;
      if (rel < 0.25) s.radius *= 0.5;                         // DoglegStrategy::StepAccepted
      if (rel > 0.75) s.radius = fmax(s.radius, 3.0 * s.step_norm);
      if (s.radius > 1e16) s.radius = 1e16;
      s.mu = fmax(1e-8, 2.0 * s.mu / 10.0);
      s.reuse = 0;
    } else {                                                   // StepRejected
      s.radius *= 0.5;
      s.reuse = 1;
    }
  }
  final_cost = x_cost;
  return iter;
}

// ------------------------------------------------------------------ K7a
__device__ __forceinline__ bool partial_less(const GridPartial& a, const GridPartial& b) {
  return a.cost < b.cost || (a.cost == b.cost && (a.d2 < b.d2 || (a.d2 == b.d2 && a.flat < b.flat)));
}

// pass A then pass B of (frame f, phase slot) on the points q.yz / q.lab: one wavefront (WIDE = false) or the whole workgroup
template <bool WIDE>
__device__ __forceinline__ void solve_wave(const Ctx& c, Problem& q, uint32_t f, uint32_t slot, SolveRec* out, Dog& dog) {
  double x[3] = {0.0, 0.0, 0.0};
  int phase = (int)slot;
  if (c.p.phase_mode != 2) {
    phase = (c.p.phase_mode == 1) ? 1 : 0;
  }
  q.tlw = phase != 0;
  double cost[2] = {0.0, 0.0};
  int iters[2] = {0, 0};
#pragma nounroll
  for (int pass = 0; pass < 2; ++pass) {
    q.oob = pass == 0;   // pass A: useOutofBoard (LidarCornersEst.cpp:403-405), pass B: not (:406-408)
    double fc = 0.0;
    const int it = trust_region_minimize<WIDE>(q, x, fc, c.p.max_iterations, dog);
    cost[pass] = fc;
    iters[pass] = it;
  }
#ifdef ILCC_K7_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    printf("K7a f0 slot %d: iters %d + %d; cycles dogleg %llu, evaluate<false> %llu, evaluate<true> %llu\n", (int)slot, iters[0], iters[1], g_k7_t[0], g_k7_t[1], g_k7_t[2]);
    g_k7_t[0] = g_k7_t[1] = g_k7_t[2] = 0;
  }
#endif
  q.oob = true;
  double cs[10];
  evaluate<false, WIDE>(q, x, cs);
  if (WIDE ? threadIdx.x == 0 : lane_id() == 0) {
    out->x[0] = x[0];
    out->x[1] = x[1];
    out->x[2] = x[2];
    out->cost_a = cost[0];
    out->cost_b = cost[1];
    out->sel = cs[0];
    out->iters_a = iters[0];
    out->iters_b = iters[1];
    out->phase = phase;
    out->valid = 1;
    out->margin = 0.0;
    out->flags = 0;
    out->ties = 0;
  }
}

// grid: ceil(n_frames * n_slots / kSolveWaves) workgroups; wavefront w of workgroup b runs solve b * kSolveWaves + w =
// (frame, slot) = (solve / n_slots, solve % n_slots): with two slots the wavefronts 2k and 2k + 1 share a frame and its
// staged points.  Dynamic LDS: (kSolveWaves / n_slots) frames x grid_lds_points x 9 bytes.
#ifndef ILCC_K7A_WAVES_PER_EU
#define ILCC_K7A_WAVES_PER_EU 3
#endif
__global__ __launch_bounds__(kSolveThreads) __attribute__((amdgpu_waves_per_eu(ILCC_K7A_WAVES_PER_EU, ILCC_K7A_WAVES_PER_EU))) void k7a_local_solve(Ctx c, SolveRec* rec, int n_slots) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ Dog s_dog[kSolveWaves];   // one dogleg state per wavefront (= per solve)
  const uint32_t wave = (uint32_t)wave_id();
  const uint32_t solve = blockIdx.x * (uint32_t)kSolveWaves + wave;
  const uint32_t f = solve / (uint32_t)n_slots, slot = solve % (uint32_t)n_slots;
  const bool live = f < c.n_frames && c.res[f].status == ILCC_OK;
  const uint32_t n = live ? c.n_lab[f] : 0u;
  const bool in_lds = n <= c.grid_lds_points;
  // staging: the n_slots wavefronts of a frame copy its points together
  const uint32_t local_frame = wave / (uint32_t)n_slots;
  float2* s_yz = reinterpret_cast<float2*>(smem) + (size_t)local_frame * c.grid_lds_points;
  uint8_t* s_lab = smem + sizeof(float2) * (size_t)c.grid_lds_points * (size_t)(kSolveWaves / n_slots) + (size_t)local_frame * c.grid_lds_points;
  const uint64_t beg = live ? c.off[f] : 0u;
  if (live && in_lds) {
    for (uint32_t i = slot * ILCC_WAVE + (uint32_t)lane_id(); i < n; i += (uint32_t)n_slots * ILCC_WAVE) {
      s_yz[i] = c.yz[beg + i];
      s_lab[i] = c.lab[beg + i];
    }
  }
  __syncthreads();   // the only barrier: from here on every wavefront is on its own
  if (f >= c.n_frames) return;
  SolveRec* out = &rec[2 * f + slot];
  if (!live) {
    if (lane_id() == 0) out->valid = 0;
    return;
  }
  Problem q;
  q.n = n;
  q.bd = make_board(c.p);
  if (in_lds) {
    q.yz = s_yz;
    q.lab = s_lab;
    solve_wave<false>(c, q, f, slot, out, s_dog[wave]);
  } else {   // a frame above the handle's LDS capacity (it grows after the batch): the points stay in HBM / L2
    q.yz = c.yz + beg;
    q.lab = c.lab + beg;
    solve_wave<false>(c, q, f, slot, out, s_dog[wave]);
  }
}

// Small batches (n_frames <= kSolveWideMaxFrames: fewer solves than SIMDs): one 256-thread workgroup per (frame, slot), the
// layout of rounds 1-5 on this round's arithmetic -- 128 frames alone on the chip: 0.62 ms (round 5), 1.11 ms (one wavefront per
// solve), see profiles/README.md for this variant.  Both layouts add the same terms in the same order (evaluate<>): a frame's
// result does not depend on the batch it came in.
__global__ __launch_bounds__(kSolveThreads) void k7a_local_solve_wide(Ctx c, SolveRec* rec) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double s_red[2 * kSolveRedDoubles];
  __shared__ Dog s_dog[kSolveWaves];   // every wavefront keeps its own copy of the (identical) dogleg state: no cross-wavefront hazards
  const uint32_t f = blockIdx.x, slot = blockIdx.y;
  SolveRec* out = &rec[2 * f + slot];
  if (c.res[f].status != ILCC_OK) {
    if (threadIdx.x == 0) out->valid = 0;
    return;
  }
  float2* s_yz = reinterpret_cast<float2*>(smem);
  uint8_t* s_lab = smem + sizeof(float2) * (size_t)c.grid_lds_points;
  const uint64_t beg = c.off[f];
  const uint32_t n = c.n_lab[f];
  int flip = 0;
  Problem q;
  q.n = n;
  q.bd = make_board(c.p);
  q.red = s_red;
  q.flip = &flip;
  if (n <= c.grid_lds_points) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      s_yz[i] = c.yz[beg + i];
      s_lab[i] = c.lab[beg + i];
    }
    __syncthreads();
    q.yz = s_yz;
    q.lab = s_lab;
    solve_wave<true>(c, q, f, slot, out, s_dog[wave_id()]);
  } else {
    q.yz = c.yz + beg;
    q.lab = c.lab + beg;
    solve_wave<true>(c, q, f, slot, out, s_dog[wave_id()]);
  }
}

// ------------------------------------------------------------------ K7r pattern refine (ILCC_SOLVER_GRID)
constexpr double kCostQOne = 1099511627776.0;   // 2^40: quantum of the fixed-point cost (oracle: ORC_COST_Q_ONE)
// wavefronts of the K7r workgroup: kRefineThreads / 64 in large batches, kRefineThreadsSmallBatch / 64 in small ones (the
// sums are integers: the split of the points over wavefronts cannot change a total)
#define kRefineWaves ((int)(blockDim.x >> 6))
// the 3-theta stencil gives every theta kRefineWaves / 3 wavefronts: fewer than 3 wavefronts would leave a theta without
// any (all 27 sums 0 -- a silently wrong refinement, not a build error), and the candidate lists are written by the first
// kRefineList threads
static_assert(kRefineThreads % ILCC_WAVE == 0 && kRefineThreads / ILCC_WAVE >= 3 && kRefineThreads >= kRefineList &&
              kRefineThreadsSmallBatch % (3 * ILCC_WAVE) == 0,
              "ILCC_K7R_THREADS must be a multiple of 64, at least 192");
constexpr float kBorderRisk = 4e-6f;             // squares: a board coordinate this close to an integer may be classified differently in fp32 (K6) and fp64
constexpr int32_t kNoTheta = INT32_MIN;         // candidate whose theta lies outside the lattice table: never evaluated

struct RCand {
  int32_t q0, q1, q2, phase;   // lattice coordinates (theta, ty, tz), topleftWhite
};
constexpr int kRefineWavesMax = kRefineThreadsSmallBatch / ILCC_WAVE;
constexpr int kActQueue = 128;                   // per-wavefront queue of active point indices: < 64 left over + <= 64 pushed
struct RefineShared {
  RCand cand[kRefineList];
  GridPartial meta[kRefineList];                 // near-tie recount: fp32 cost, d2, flat of the listed candidates
  unsigned long long acc[3][kRefineList];        // fixed-point sums, three rotating sets (one barrier pair per sweep)
  uint32_t queue[kRefineWavesMax][kActQueue];    // stencil_sweep: the points whose nine terms are not provably all zero
};

struct RefineState {
  int32_t lat[3];
  int32_t phase, rounds, hops, capped;
  int32_t skip_first;   // the start is the exhaustive grid's argmin with all 26 grid neighbours inside the grid (see pattern_refine)
  long long cost, alt;
};

// The fixed-point cost's per-point term, operation for operation the oracle's term_q (oracle/ilcc_oracle.c): the
// functor's residual (Optimization.h:31-107) with the grid coordinate scaled by 1/g (computed once) instead of divided
// by g, and Huber's rho taken on r directly instead of through sqrt(r^2) -- no fp64 division or square root per
// point.  AxisTerms = everything the residual needs from one axis (v = i, n = W or v = j, n = H), so that the nine
// (ty, tz) combinations of a stencil share the three i's and three j's.
struct AxisTerms {
  double in_dist;    // min(frac, 1 - frac) as :70-78 compute it
  double out_dist;   // min(|v|, |v - n|) as :86-97 compute it
  bool inside;       // 0 < v < n (strict, :48-49)
  bool odd;          // floor(v) is odd
};
__device__ __forceinline__ AxisTerms axis_terms(double v, double n) {
  AxisTerms t;
  t.inside = v > 0 && v < n;
  const double fl = floor(v);
  // (floor(v) is odd) from the integer: the same truth value as `fl != floor(fl / 2) * 2` for every |v| < 2^31 -- board
  // coordinates are a few units -- at a quarter of the fp64 operations
  t.odd = (((int)fl) & 1) != 0;
  const double fr = v - fl;
  t.in_dist = (fr > 0.5) ? (fl + 1.0) - v : fr;
  t.out_dist = fmin(fabs(v), fabs(v - n));   // == (|v| < |v - n|) ? |v| : |v - n| for finite v
  return t;
}
// rint(1/2 rho * 2^40) as a double (an integer < 2^53: sums of a few of them are exact in double, too)
__device__ __forceinline__ double term_q(const AxisTerms& ai, const AxisTerms& aj, bool tlw, bool laser_white, double delta) {
#ifndef ILCC_K7R_BRANCHFREE
#define ILCC_K7R_BRANCHFREE 0   // measured: K7r alone 0.274 -> 0.233 ms per 128 frames, pipelined bench 392 -> 383 k frames/s (both sums are always computed): off
#endif
  double res = 0.0;
  if (ILCC_K7R_BRANCHFREE) {
    // the same sums as the branches below, selected instead of jumped to: nine of these per point and theta run with less
    // than one wavefront per SIMD, where every taken branch is a bubble nothing else fills
    const bool white = tlw != (ai.odd != aj.odd);           // both even or both odd -> topleftWhite (:53-61)
    const double in_sum = ai.in_dist + aj.in_dist, out_sum = ai.out_dist + aj.out_dist;
    res = (ai.inside && aj.inside) ? ((laser_white != white) ? in_sum : 0.0) : out_sum;
  } else if (ai.inside && aj.inside) {
    const bool white = (ai.odd == aj.odd) ? tlw : !tlw;     // both even or both odd -> topleftWhite (:53-61)
    if (laser_white != white) res = ai.in_dist + aj.in_dist;
  } else {
    res = ai.out_dist + aj.out_dist;                        // useOutofBoard (pass-A cost)
  }
  const double r0 = (res > delta) ? 2.0 * delta * res - delta * delta : res * res;
  return rint(r0 * (0.5 * kCostQOne));
}

// One sweep over the frame's labelled points for the n_cand (<= 32) candidates in sh.cand.  lane -> (candidate,
// slice): every lane walks its slice of the points for ONE candidate and adds its integer partial sum to the
// candidate's LDS word -- no cross-lane reduction, and the result cannot depend on who adds first.
// Returns the index of the acc set that holds the totals.  `sweep` counts the sweeps of this workgroup.
__device__ __forceinline__ int refine_sweep(const Ctx& c, const Board& bd, const float2* yz, const uint8_t* lab, uint32_t n,
                                            RefineShared& sh, int n_cand, int& sweep) {
  const int buf = sweep % 3;
  if (threadIdx.x < kRefineList) sh.acc[(sweep + 1) % 3][threadIdx.x] = 0ull;   // last read two sweeps ago
  __syncthreads();   // candidates (written by the caller) and this sweep's zeroed set are visible
  const int lane = lane_id();
  const int slices = ILCC_WAVE / n_cand;
  const int cand = lane % n_cand, slice = lane / n_cand;
  if (slice < slices) {
    const RCand cd = sh.cand[cand];
    if (cd.q0 != kNoTheta) {
      const double div = (double)(c.p.refine_div > 0 ? c.p.refine_div : 1);
      const double2 cs = c.th_lattice[cd.q0 - c.th_lat_lo];
      const double x1 = c.p.ty_min + (double)cd.q1 * (c.p.ty_step / div), x2 = c.p.tz_min + (double)cd.q2 * (c.p.tz_step / div);
      const double inv_g = 1.0 / bd.g;
      const bool tlw = cd.phase != 0;
      long long sum = 0;
      for (uint32_t p = (uint32_t)(wave_id() * slices + slice); p < n; p += (uint32_t)(kRefineWaves * slices)) {
        const float2 v = yz[p];
        const double y = (double)v.x, z = (double)v.y;
        const double ry = cs.x * y - cs.y * z;
        const double rz = cs.y * y + cs.x * z;
        const AxisTerms ai = axis_terms(((ry + x1) + bd.W * bd.g / 2.0) * inv_g, bd.W);
        const AxisTerms aj = axis_terms(((rz + x2) + bd.H * bd.g / 2.0) * inv_g, bd.H);
        sum += (long long)term_q(ai, aj, tlw, lab[p] != 0, bd.delta);
      }
      atomicAdd(&sh.acc[buf][cand], (unsigned long long)sum);
    }
  }
  __syncthreads();
  ++sweep;
  return buf;
}

// One sweep for a 3 x 3 x 3 (or 1 x 3 x 3) STENCIL of candidates: theta in th[0..n_th), ty in ty[0..3), tz in tz[0..3),
// colour phase = phase ^ (parity ? (a + b) & 1 : 0).  Wavefront w serves theta w % n_th; its lanes take points
// (slice, slice + n_slices, ...) and evaluate the 9 translations of each: rotation once, the per-axis terms of the
// three i's and three j's once, then 9 cheap combinations -- the same doubles as 27 independent term evaluations.
// Totals land in sh.acc[buf][theta * 9 + a * 3 + b].
//
// Round 5: SILENT POINTS ARE SKIPPED (pattern-search rounds, parity == false).  Most labelled points sit in a square of their
// own colour, well away from its borders: their term is exactly 0 for the centre and for every neighbour of a fine-stride
// stencil.  A point is silent for this wavefront's theta when, with the evaluation's OWN fp64 expressions,
//   0 < i(ty[0]) and i(ty[2]) < W and floor(i(ty[0])) == floor(i(ty[2])),   the same for j, and its label is the cell's colour:
// ty[0] <= ty[1] <= ty[2] and every operation of v = ((r + x) + c) * (1/g) is monotone in x, so i(ty[1]) lies between the two --
// all three i's (and j's) are strictly inside the board and in ONE cell, the colour of all nine (a, b) cells is the one just
// compared, all nine residuals are 0 and term_q(...) = rint(0) = 0: adding them is adding nothing.  The test costs a fifth of
// the nine evaluations (one rotation, four coordinates); points that fail it are queued per wavefront in LDS (a ballot
// compaction) and evaluated 64 at a time exactly as before.  Sums are integers, so the totals are bit-identical -- not
// "equal up to order" -- to evaluating every point (tests: K7r alone and every GRID-mode parity test compare == with the oracle,
// which evaluates every point).  On the bench's frames 75-95 % of the points are silent, depending on the stride.
__device__ __forceinline__ int stencil_sweep(const Ctx& c, const Board& bd, const float2* yz, const uint8_t* lab, uint32_t n,
                                             RefineShared& sh, int n_th, const int32_t th[3], const int32_t ty[3],
                                             const int32_t tz[3], int phase, bool parity, int& sweep) {
  const int buf = sweep % 3;
  if (threadIdx.x < kRefineList) sh.acc[(sweep + 1) % 3][threadIdx.x] = 0ull;   // last read two sweeps ago
  __syncthreads();
  const int wid = __builtin_amdgcn_readfirstlane(wave_id());
  const int waves_per_theta = kRefineWaves / n_th;          // 3 wavefronts: one per theta of the stencil, or all three on the basin check's single theta
  const int it = wid % n_th, grp = wid / n_th;
  const int32_t q0 = th[it];
  if (grp < waves_per_theta && q0 != kNoTheta) {
    const int lane = lane_id();
    const double div = (double)(c.p.refine_div > 0 ? c.p.refine_div : 1);
    const double2 cs = c.th_lattice[q0 - c.th_lat_lo];
    double x1[3], x2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      x1[k] = c.p.ty_min + (double)ty[k] * (c.p.ty_step / div);
      x2[k] = c.p.tz_min + (double)tz[k] * (c.p.tz_step / div);
    }
    const double inv_g = 1.0 / bd.g;
    double acc[9];   // integer-valued partial sums of a handful of terms: exact in double
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[e] = 0.0;
    auto eval_point = [&](uint32_t p) {
      const float2 v = yz[p];
      const bool laser_white = lab[p] != 0;
      const double y = (double)v.x, z = (double)v.y;
      const double ry = cs.x * y - cs.y * z;
      const double rz = cs.y * y + cs.x * z;
      AxisTerms ai[3], aj[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        ai[k] = axis_terms(((ry + x1[k]) + bd.W * bd.g / 2.0) * inv_g, bd.W);
        aj[k] = axis_terms(((rz + x2[k]) + bd.H * bd.g / 2.0) * inv_g, bd.H);
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const bool tlw = ((phase ^ (parity ? ((a + b) & 1) : 0)) != 0);
          acc[a * 3 + b] += term_q(ai[a], aj[b], tlw, laser_white, bd.delta);
        }
    };
    const uint32_t n_slices = (uint32_t)(waves_per_theta * ILCC_WAVE);
    // the silence test needs ty[0] <= ty[2] and tz[0] <= tz[2] (pattern-search rounds: centre -/+ stride, steps > 0)
    const bool skip_silent = !parity && ty[0] <= ty[1] && ty[1] <= ty[2] && tz[0] <= tz[1] && tz[1] <= tz[2] && c.p.ty_step > 0.0 && c.p.tz_step > 0.0;   // (monotone in BOTH steps: the middle value's cell lies between the outer two's)
    if (skip_silent) {
      uint32_t* queue = sh.queue[wid];
      uint32_t head = 0, tail = 0;   // wave-uniform
      const bool tlw0 = phase != 0;
      for (uint32_t base = (uint32_t)(grp * ILCC_WAVE); base < n; base += n_slices) {
        const uint32_t p = base + (uint32_t)lane;
        bool active = false;
        if (p < n) {
          const float2 v = yz[p];
          const double y = (double)v.x, z = (double)v.y;
          const double ry = cs.x * y - cs.y * z;
          const double rz = cs.y * y + cs.x * z;
          const double i0 = ((ry + x1[0]) + bd.W * bd.g / 2.0) * inv_g, i2 = ((ry + x1[2]) + bd.W * bd.g / 2.0) * inv_g;
          const double j0 = ((rz + x2[0]) + bd.H * bd.g / 2.0) * inv_g, j2 = ((rz + x2[2]) + bd.H * bd.g / 2.0) * inv_g;
          const double fi = floor(i0), fj = floor(j0);
          const bool one_cell = i0 > 0 && i2 < bd.W && j0 > 0 && j2 < bd.H && fi == floor(i2) && fj == floor(j2);
          const bool odd_i = (((int)fi) & 1) != 0, odd_j = (((int)fj) & 1) != 0;
          const bool white = (odd_i == odd_j) ? tlw0 : !tlw0;   // term_q's colour rule (:53-61)
          active = !(one_cell && (lab[p] != 0) == white);
        }
        const unsigned long long m = __ballot(active);
        if (m != 0ull) {
          if (active) queue[(tail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (kActQueue - 1)] = p;
          tail += (uint32_t)__popcll(m);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // one wavefront: LDS executes its instructions in order; keep the compiler from reordering
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          while (tail - head >= (uint32_t)ILCC_WAVE) {
            eval_point(queue[(head + (uint32_t)lane) & (kActQueue - 1)]);
            head += ILCC_WAVE;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the reads above before the next round's writes
          __builtin_amdgcn_wave_barrier();
        }
      }
      if ((uint32_t)lane < tail - head) eval_point(queue[(head + (uint32_t)lane) & (kActQueue - 1)]);
    } else {
      for (uint32_t p = (uint32_t)(grp * ILCC_WAVE + lane); p < n; p += n_slices) eval_point(p);
    }
    // integer sums: any reduction order gives the same totals
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      // within the 16-lane rows only (plain DPP); the four rows then add their totals to the LDS word themselves -- the two
      // cross-row steps (permlane swaps and selects on both halves of a 64-bit value) cost more than three more atomics
      unsigned long long t = (unsigned long long)(long long)acc[e];
      t += xor_lane_u64<8>(t);
      t += xor_lane_u64<4>(t);
      t += xor_lane_u64<2>(t);
      t += xor_lane_u64<1>(t);
      if ((lane & 15) == 0) atomicAdd(&sh.acc[buf][it * 9 + e], t);
    }
  }
  __syncthreads();
  ++sweep;
  return buf;
}

// orc_pattern_refine, executed redundantly (and identically) by every thread of the workgroup
__device__ void pattern_refine(const Ctx& c, const Board& bd, const float2* yz, const uint8_t* lab, uint32_t n,
                               RefineShared& sh, int& sweep, RefineState& st) {
  const bool refine = c.p.refine_div > 0;
  const int div = refine ? c.p.refine_div : 1;
  st.rounds = 0;
  st.hops = 0;
  st.capped = 0;
  st.cost = 0;
  st.alt = 0;
  for (;;) {
    int stride = div, r = 0;
    if (st.skip_first && refine && div >= 2 && c.p.refine_max_rounds >= 1) {   // (first search only: a hop restarts in full)
      stride = div >> 1;
      r = 1;
    }
    st.skip_first = 0;
    while (refine && stride >= 1 && r < c.p.refine_max_rounds) {
      int32_t th[3], ty[3], tz[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q0 = st.lat[0] + (k - 1) * stride;
        th[k] = (q0 < c.th_lat_lo || q0 > c.th_lat_hi) ? kNoTheta : q0;
        ty[k] = st.lat[1] + (k - 1) * stride;
        tz[k] = st.lat[2] + (k - 1) * stride;
      }
      const int b = stencil_sweep(c, bd, yz, lab, n, sh, 3, th, ty, tz, st.phase, false, sweep);
      long long bc = LLONG_MAX;
      int bd2 = 0, be = -1;
      for (int e = 0; e < 27; ++e) {   // (dk, da, db) order; ties: nearer, then first
        if (e == 13) continue;
        const int dk = e / 9 - 1, da = (e / 3) % 3 - 1, db = e % 3 - 1;
        const int q0 = st.lat[0] + dk * stride;
        if (q0 < c.th_lat_lo || q0 > c.th_lat_hi) continue;
        const long long cc = (long long)sh.acc[b][e];
        const int d2 = dk * dk + da * da + db * db;
        if (cc < bc || (cc == bc && d2 < bd2)) {
          bc = cc;
          bd2 = d2;
          be = e;
        }
      }
      const long long centre = (long long)sh.acc[b][13];
      ++r;
      if (be >= 0 && bc < centre) {
        st.lat[0] += (be / 9 - 1) * stride;
        st.lat[1] += ((be / 3) % 3 - 1) * stride;
        st.lat[2] += (be % 3 - 1) * stride;
      } else {
        stride >>= 1;
      }
    }
    st.rounds += r;
    if (refine && stride >= 1) st.capped = 1;   // left the loop on the round cap, not on the stride
    // the eight neighbouring basins (one square along y and/or z; an odd shift swaps the colours) + the centre
    const int32_t th1[3] = {st.lat[0], kNoTheta, kNoTheta};
    const int32_t hy[3] = {st.lat[1] - c.refine_hop_y, st.lat[1], st.lat[1] + c.refine_hop_y};
    const int32_t hz[3] = {st.lat[2] - c.refine_hop_z, st.lat[2], st.lat[2] + c.refine_hop_z};
    // ((da + db) & 1 with da, db in {-1, 0, 1} == (a + b) & 1 with a = da + 1, b = db + 1)
    const int b = stencil_sweep(c, bd, yz, lab, n, sh, 1, th1, hy, hz, st.phase, true, sweep);
    st.cost = (long long)sh.acc[b][4];
    long long alt = LLONG_MAX;
    int ae = -1;
    for (int e = 0; e < 9; ++e) {
      if (e == 4) continue;
      const long long cc = (long long)sh.acc[b][e];
      if (cc < alt) {
        alt = cc;
        ae = e;
      }
    }
    st.alt = alt;
    if (alt < st.cost && st.hops < 2 && refine) {
      const int da = ae / 3 - 1, db = ae % 3 - 1;
      ++st.hops;
      st.lat[1] += da * c.refine_hop_y;
      st.lat[2] += db * c.refine_hop_z;
      st.phase ^= (da + db) & 1;
      continue;
    }
    break;
  }
}

template <bool LDS_POINTS>
__device__ void refine_frame(const Ctx& c, SolveRec* rec, float2* s_yz, uint8_t* s_lab, RefineShared& sh) {
  const uint32_t f = blockIdx.x;
  ilcc_result* r = &c.res[f];
  SolveRec* out = &rec[2 * f];
  const int lane = lane_id(), tid = (int)threadIdx.x;
  const uint64_t beg = c.off[f];
  const uint32_t n = c.n_lab[f];
  const Board bd{(double)c.p.board_w, (double)c.p.board_h, c.p.grid_length, c.p.huber_delta};
  const float2* yz = c.yz + beg;
  const uint8_t* lab = c.lab + beg;
  if (tid < kRefineList) sh.acc[0][tid] = 0ull;
  if (LDS_POINTS) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      s_yz[i] = yz[i];
      s_lab[i] = lab[i];
    }
    yz = s_yz;
    lab = s_lab;
  }
  __syncthreads();

  // argmin over this frame's K6 partials: cost, then index distance to zero, then flat index (every wavefront
  // repeats the same reduction: no broadcast needed)
  const GridPartial* gp = c.partial + (uint64_t)f * c.grid_blocks;
  GridPartial b{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
  for (uint32_t k = lane; k < c.grid_blocks; k += ILCC_WAVE) {
    const GridPartial t = gp[k];
    if (partial_less(t, b)) b = t;
  }
#pragma unroll
  for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) {
    GridPartial t;
    t.cost = __shfl_xor(b.cost, o, ILCC_WAVE);
    t.d2 = __shfl_xor(b.d2, o, ILCC_WAVE);
    t.flat = __shfl_xor(b.flat, o, ILCC_WAVE);
    if (partial_less(t, b)) b = t;
  }
  if (b.flat == 0xFFFFFFFFu) {
    if (tid == 0) out->valid = 0;
    return;
  }
  int sweep = 0;
  int flags = 0;
  uint32_t n_ties = 0;
  const int div = c.p.refine_div > 0 ? c.p.refine_div : 1;
  const uint32_t n_ty = (uint32_t)c.p.n_ty, n_tz = (uint32_t)c.p.n_tz;
  // Near ties: fp32 sums of ~1e3 terms cannot order candidates whose costs agree to ~1e-6; the oracle orders them
  // on exact fixed-point sums.  K6's full pass listed every candidate within kTieEps of the bound; recount the
  // ones within kTieEps of the fp32 minimum with the oracle's arithmetic and apply its tie-break to those values.
  if (c.tie_count != nullptr) {
    n_ties = c.tie_count[f];
    if (n_ties > (uint32_t)kTieCap) {
      flags |= ILCC_FLAG_TIE_OVERFLOW;   // the list is incomplete: keep the (deterministic) fp32 argmin
    } else {
      const GridPartial* tl = c.tie_list + (uint64_t)f * kTieCap;
      const float window = b.cost * (1.f + kTieEps);
      uint32_t close = 0;
      for (uint32_t e = 0; e < n_ties; ++e) close += (tl[e].cost <= window && tl[e].flat != b.flat) ? 1u : 0u;
      if (close > 0) {   // uniform across the workgroup
        long long best_q = LLONG_MAX;
        GridPartial pick = b;
        int k = 0;
        for (uint32_t e = 0; e <= n_ties; ++e) {   // e == n_ties: the fp32 argmin itself
          const GridPartial t = (e < n_ties) ? tl[e] : b;
          const bool take = (t.cost <= window) && !(e < n_ties && t.flat == b.flat);
          if (take) {
            if (tid == 0) {
              const uint32_t cl = t.flat >> 1;
              sh.cand[k] = RCand{(int32_t)(cl / (n_tz * n_ty)) * div, (int32_t)((cl / n_tz) % n_ty) * div, (int32_t)(cl % n_tz) * div, (int32_t)(t.flat & 1u)};
              sh.meta[k] = t;
            }
            ++k;
          }
          if (k == kRefineList || (e == n_ties && k > 0)) {
            const int bf = refine_sweep(c, bd, yz, lab, n, sh, k, sweep);
            for (int j = 0; j < k; ++j) {
              const long long cq = (long long)sh.acc[bf][j];
              const GridPartial m = sh.meta[j];
              if (cq < best_q || (cq == best_q && (m.d2 < pick.d2 || (m.d2 == pick.d2 && m.flat < pick.flat)))) {
                best_q = cq;
                pick = m;
              }
            }
            k = 0;
            __syncthreads();   // every thread has read meta / cand before thread 0 refills them
          }
        }
        b = pick;
      }
    }
  }
  if (tid == 0) {
    r->grid_index = (int32_t)b.flat;
    r->grid_cost = b.cost;
  }
  const uint32_t cell = b.flat >> 1;
  RefineState st;
  st.lat[0] = (int32_t)(cell / (n_tz * n_ty)) * div;
  st.lat[1] = (int32_t)((cell / n_tz) % n_ty) * div;
  st.lat[2] = (int32_t)(cell % n_tz) * div;
  st.phase = (int)(b.flat & 1u);
#ifndef ILCC_K7R_SKIP_FIRST
#define ILCC_K7R_SKIP_FIRST 1
#endif
  {
    // The first round of the first search looks at the 26 neighbours one GRID step away.  When they are all grid candidates,
    // K6 has already ranked them: pruned ones cost more than (1 + 2e-5) x the minimum, completed ones within that window were
    // re-ordered on exact costs above -- none can be STRICTLY cheaper than the argmin, the round cannot move and is not
    // evaluated (it still counts as a round: `rounds` stays the oracle's number).  Not when the near-tie list overflowed
    // (the argmin is then the fp32 one).
    // ... and not when K6's ranking of these 27 candidates cannot be trusted: K6 sums fp32 terms, and a point within fp32 rounding
    // of a cell border under one of them may sit in the OTHER cell there -- its term then differs from the exact one by a whole
    // residual, not by rounding, and "pruned => costs more" no longer follows.  The test below recomputes, with K6's own fp32
    // expressions (its staging's rotation, its table values), the board coordinates of every labelled point under the 3 thetas
    // x (3 + 3) axis translations of the neighbourhood and looks for one within 4e-6 square of an integer (fp32 and fp64 agree
    // to ~1e-6 there).  None: every point is classified alike in fp32 and fp64 for all 27, their fp32 costs are the exact ones
    // up to summation rounding (1e-6 relative, far inside the 2e-5 window) and the shortcut is sound.  Any: ILCC_FLAG_BORDER_RISK,
    // and the round is evaluated like every other.
    const int gk = (int)(cell / (n_tz * n_ty)), ga = (int)((cell / n_tz) % n_ty), gb = (int)(cell % n_tz);
    bool risk = false;
    for (uint32_t i = (uint32_t)tid; i < n; i += blockDim.x) {
      const float2 v = yz[i];
#pragma unroll
      for (int dk = -1; dk <= 1; ++dk) {
        const int k = gk + dk;
        if (k < 0 || k >= c.p.n_th) continue;
        const float cth = c.cth[k], sth = c.sth[k];
        const float pi = fmaf(-sth, v.y, cth * v.x), pj = fmaf(cth, v.y, sth * v.x);   // = k6_grid_cost's staging
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
          const int a = ga + d, b = gb + d;
          if (a >= 0 && a < (int)n_ty) {
            const float x = pi + c.ay[a];
            risk |= fabsf(x - rintf(x)) < kBorderRisk;
          }
          if (b >= 0 && b < (int)n_tz) {
            const float x = pj + c.az[b];
            risk |= fabsf(x - rintf(x)) < kBorderRisk;
          }
        }
      }
    }
    risk = __syncthreads_or(risk ? 1 : 0) != 0;
    if (risk) flags |= ILCC_FLAG_BORDER_RISK;
    st.skip_first = ILCC_K7R_SKIP_FIRST && !risk && !(flags & ILCC_FLAG_TIE_OVERFLOW) && gk > 0 && gk < c.p.n_th - 1 && ga > 0 &&
                    ga < (int)n_ty - 1 && gb > 0 && gb < (int)n_tz - 1;
  }
  pattern_refine(c, bd, yz, lab, n, sh, sweep, st);
  if (tid == 0) {
    const double dv = (double)div;
    out->x[0] = c.p.th_min + (double)st.lat[0] * (c.p.th_step / dv);
    out->x[1] = c.p.ty_min + (double)st.lat[1] * (c.p.ty_step / dv);
    out->x[2] = c.p.tz_min + (double)st.lat[2] * (c.p.tz_step / dv);
    out->cost_a = out->sel = (double)st.cost / kCostQOne;
    out->cost_b = (double)st.alt / kCostQOne;
    out->margin = ((double)st.alt - (double)st.cost) / (double)(st.cost > 0 ? st.cost : 1);
    out->iters_a = st.rounds;
    out->iters_b = st.hops;
    out->phase = st.phase;
    out->valid = 1;
    out->flags = flags | (st.capped ? ILCC_FLAG_REFINE_CAPPED : 0);
    out->ties = (int32_t)n_ties;
  }
}

__global__ __launch_bounds__(kRefineThreadsSmallBatch) void k7r_pattern_refine(Ctx c, SolveRec* rec) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ RefineShared sh;
  const uint32_t f = blockIdx.x;
  if (c.res[f].status != ILCC_OK) {
    if (threadIdx.x == 0) rec[2 * f].valid = 0;
    return;
  }
  float2* s_yz = reinterpret_cast<float2*>(smem);
  uint8_t* s_lab = smem + sizeof(float2) * (size_t)c.grid_lds_points;
  if (c.n_lab[f] <= c.grid_lds_points)
    refine_frame<true>(c, rec, s_yz, s_lab, sh);
  else
    refine_frame<false>(c, rec, s_yz, s_lab, sh);
}

// test entry: the refinement alone on frame 0's labelled points (global memory), start given by the caller
__global__ __launch_bounds__(kRefineThreads) void k7r_pattern_refine_test(Ctx c, RefineOut* io) {
  __shared__ RefineShared sh;
  if (threadIdx.x < kRefineList) sh.acc[0][threadIdx.x] = 0ull;
  __syncthreads();
  const Board bd{(double)c.p.board_w, (double)c.p.board_h, c.p.grid_length, c.p.huber_delta};
  RefineState st;
  st.lat[0] = io->lat[0];
  st.lat[1] = io->lat[1];
  st.lat[2] = io->lat[2];
  st.phase = io->phase;
  st.skip_first = 0;   // an arbitrary start: nothing is known about its grid neighbours
  int sweep = 0;
  __syncthreads();
  pattern_refine(c, bd, c.yz, c.lab, c.n_lab[0], sh, sweep, st);
  if (threadIdx.x == 0) {
    io->lat[0] = st.lat[0];
    io->lat[1] = st.lat[1];
    io->lat[2] = st.lat[2];
    io->phase = st.phase;
    io->rounds = st.rounds;
    io->hops = st.hops;
    io->cost_q = st.cost;
    io->alt_q = st.alt;
  }
}

// ------------------------------------------------------------------ K7b corners (getPCDcorners)
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

__global__ __launch_bounds__(kSolveThreads) void k7b_corners(Ctx c, const SolveRec* rec, int n_slots) {
  __shared__ float s_T[16];
  __shared__ uint32_t s_hit[kCoverageCellsMax / 32];
  __shared__ uint32_t s_oob;
  const uint32_t f = blockIdx.x;
  ilcc_result* r = &c.res[f];
  if (r->status != ILCC_OK) return;
  const uint32_t tid = threadIdx.x;
  const uint64_t beg = c.off[f];
  // phase selection: lower with-OOB cost, ties -> phase 0 (the reference's first hypothesis)
  SolveRec best = rec[2 * f];
  if (n_slots > 1) {
    const SolveRec o = rec[2 * f + 1];
    if (o.valid && (!best.valid || o.sel < best.sel)) best = o;
  }
  if (!best.valid) {
    if (tid == 0) r->status = ILCC_BAD_ARGUMENT;
    return;
  }
  // transf = pcl::getTransformation(0, ty, tz, theta, 0, 0): float Affine3f (:412)
  if (tid == 0) {
    r->theta_t[0] = best.x[0];
    r->theta_t[1] = best.x[1];
    r->theta_t[2] = best.x[2];
    r->cost_a = best.cost_a;
    r->cost_b = best.cost_b;
    r->sel_cost = best.sel;
    r->phase = best.phase;
    r->iters_a = best.iters_a;
    r->iters_b = best.iters_b;
    r->basin_margin = best.margin;
    r->flags = best.flags;
    r->grid_ties = best.ties;
    const float roll = (float)best.x[0];
    const float E = cosf(roll), F = sinf(roll);
    const float T[16] = {1, 0, 0, 0, 0, E, -F, (float)best.x[1], 0, F, E, (float)best.x[2], 0, 0, 0, 1};
    for (int k = 0; k < 16; ++k) s_T[k] = T[k];
    s_oob = 0u;
  }
  for (int k = (int)tid; k < kCoverageCellsMax / 32; k += kSolveThreads) s_hit[k] = 0u;
  __syncthreads();

  const int W = c.p.board_w, H = c.p.board_h;
  // Coverage of the virtual board by the labelled points (orc_coverage): the functor's own coordinates
  // (Optimization.h:37-49) in double.  What the operator checks at the viewer before pressing 'o' (:415-441).
  {
    const double g = c.p.grid_length, ct = cos(best.x[0]), st = sin(best.x[0]);
    const uint32_t nl = c.n_lab[f];
    const float2* __restrict__ yz = c.yz + beg;
    uint32_t oob = 0;
    for (uint32_t k = tid; k < nl; k += kSolveThreads) {
      const float2 v = yz[k];
      const double yy = ct * (double)v.x - st * (double)v.y + best.x[1];
      const double zz = st * (double)v.x + ct * (double)v.y + best.x[2];
      const double i = (yy + W * g / 2.0) / g, j = (zz + H * g / 2.0) / g;
      if (i > 0.0 && i < (double)W && j > 0.0 && j < (double)H) {
        const int cell = (int)floor(i) * H + (int)floor(j);
        atomicOr(&s_hit[cell >> 5], 1u << (cell & 31));
      } else {
        ++oob;
      }
    }
    if (oob) atomicAdd(&s_oob, oob);
    __syncthreads();
    if (tid == 0) {
      int cells = 0;
      for (int k = 0; k < (W * H + 31) / 32; ++k) cells += __popc(s_hit[k]);
      r->cells_hit = cells;
      r->n_oob = (int32_t)s_oob;
      if (c.p.min_cell_coverage > 0.0 && (double)cells < c.p.min_cell_coverage * (double)(W * H)) r->flags = best.flags | ILCC_FLAG_LOW_COVERAGE;
    }
  }
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
  // GRID mode: a neighbouring basin that costs (almost) the same -> the caller is told (corners stay in the record).
  // Written last: every thread above read r->status == ILCC_OK before this store can land (barrier)
  __syncthreads();
  if (tid == 0 && c.p.solver == ILCC_SOLVER_GRID && c.p.ambiguity_eps > 0.0 && best.margin < c.p.ambiguity_eps) r->status = ILCC_AMBIGUOUS;
}

// test entry: one solve (one wavefront) on frame 0's labelled points (global memory)
__global__ __launch_bounds__(ILCC_WAVE) void k7_local_solve_test(Ctx c, int tlw, int use_oob, double* theta_t,
                                                                 double* cost_iters) {
  Problem q;
  q.yz = c.yz;
  q.lab = c.lab;
  q.n = c.n_lab[0];
  q.bd = make_board(c.p);
  q.tlw = tlw != 0;
  q.oob = use_oob != 0;
  __shared__ Dog s_dog1;
  double x[3] = {theta_t[0], theta_t[1], theta_t[2]};
  double cost = 0;
  const int it = trust_region_minimize<false>(q, x, cost, c.p.max_iterations, s_dog1);
  if (threadIdx.x == 0) {
    theta_t[0] = x[0];
    theta_t[1] = x[1];
    theta_t[2] = x[2];
    cost_iters[0] = cost;
    cost_iters[1] = (double)it;
  }
}

hipError_t set_kernel_attributes_k7() {
  const int cap = (int)((sizeof(float2) + 1) * (size_t)kGridLdsPointsMax);
  // (K7a stages up to kSolveWaves frames per workgroup; launch_refine_corners falls back to the global-memory path -- LDS capacity 0 --
  // when they would not fit the 160 KB of a CU)
  hipError_t e = hipFuncSetAttribute((const void*)k7a_local_solve, hipFuncAttributeMaxDynamicSharedMemorySize, kSolveLdsMax);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k7a_local_solve_wide, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k7r_pattern_refine, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  return e;
}

void launch_pattern_refine_corners(const Ctx& c, hipStream_t s) {
  const size_t lds = (sizeof(float2) + 1) * (size_t)c.grid_lds_points;
  const int threads = c.n_frames <= (uint32_t)kSmallBatchFrames ? kRefineThreadsSmallBatch : kRefineThreads;
  hipLaunchKernelGGL(k7r_pattern_refine, dim3(c.n_frames), dim3(threads), lds, s, c, c.solve_rec);
  hipLaunchKernelGGL(k7b_corners, dim3(c.n_frames), dim3(kSolveThreads), 0, s, c, c.solve_rec, 1);
}

void launch_pattern_refine_test(const Ctx& c, hipStream_t s, RefineOut* d_io) {
  hipLaunchKernelGGL(k7r_pattern_refine_test, dim3(1), dim3(kRefineThreads), 0, s, c, d_io);
}

void launch_refine_corners(const Ctx& c, hipStream_t s) {
  const int n_slots = (c.p.solver == ILCC_SOLVER_REFERENCE_LOCAL && c.p.phase_mode == 2) ? 2 : 1;
  if (c.n_frames <= (uint32_t)kSolveWideMaxFrames) {
    const size_t lds1 = (sizeof(float2) + 1) * (size_t)c.grid_lds_points;
    hipLaunchKernelGGL(k7a_local_solve_wide, dim3(c.n_frames, n_slots), dim3(kSolveThreads), lds1, s, c, c.solve_rec);
    hipLaunchKernelGGL(k7b_corners, dim3(c.n_frames), dim3(kSolveThreads), 0, s, c, c.solve_rec, n_slots);
    return;
  }
  Ctx ck = c;
  size_t lds = (sizeof(float2) + 1) * (size_t)c.grid_lds_points * (size_t)(kSolveWaves / n_slots);
  if (lds > (size_t)kSolveLdsMax) {   // very large frames: the points stay in HBM / L2
    ck.grid_lds_points = 0;
    lds = 0;
  }
  const uint32_t solves = c.n_frames * (uint32_t)n_slots;
  hipLaunchKernelGGL(k7a_local_solve, dim3((solves + kSolveWaves - 1) / kSolveWaves), dim3(kSolveThreads), lds, s, ck, c.solve_rec, n_slots);
  hipLaunchKernelGGL(k7b_corners, dim3(c.n_frames), dim3(kSolveThreads), 0, s, c, c.solve_rec, n_slots);
}

void launch_local_solve(const Ctx& c, hipStream_t s, int32_t tlw, int32_t use_oob, double* theta_t,
                        double* cost_iters) {
  hipLaunchKernelGGL(k7_local_solve_test, dim3(1), dim3(ILCC_WAVE), 0, s, c, tlw, use_oob, theta_t, cost_iters);
}

// K9 pack_records: ilcc_result[] (device) -> fixed-size float records [n_frames, ILCC_RECORD_HEADER + 3 * n_corners]
// for the path's single collective (the gather of corner records): the records go from
// this GPU's HBM straight into RCCL, no host round trip.  Layout = sharding.pack_records.  tag = tag_base + frame
// and check = 24-bit xor-fold of the corner bits and the tag let the receiving rank verify WHOSE records arrived
// where and that their contents are intact (both are exact in a float).
__global__ __launch_bounds__(128) void k9_pack_records(const ilcc_result* __restrict__ res, uint32_t n_corners, uint32_t tag_base,
                                                       float* __restrict__ out) {
  __shared__ uint32_t s_x[2];
  const ilcc_result& r = res[blockIdx.x];
  const uint32_t width = (uint32_t)ILCC_RECORD_HEADER + 3u * n_corners;
  float* o = out + (uint64_t)blockIdx.x * width;
  const uint32_t have = (uint32_t)(r.n_corners < 0 ? 0 : r.n_corners);
  const uint32_t tag = (tag_base + blockIdx.x) & 0xFFFFFFu;
  uint32_t x = 0;
  for (uint32_t k = threadIdx.x; k < width; k += blockDim.x) {
    float v = 0.f;
    if (k >= (uint32_t)ILCC_RECORD_HEADER) {
      const uint32_t c = k - (uint32_t)ILCC_RECORD_HEADER;
      v = (c < 3u * have) ? r.corners[c] : 0.f;
      x ^= __float_as_uint(v) * (2u * c + 1u);   // position-dependent: swapped corners change the fold
    } else {
      switch (k) {
        case 0: v = (float)r.status; break;
        case 1: v = (float)r.n_corners; break;
        case 2: v = (float)r.phase; break;
        case 3: v = (float)r.grid_index; break;
        case 4: v = (float)r.iters_a; break;
        case 5: v = (float)r.iters_b; break;
        case 6: v = (float)r.cost_a; break;
        case 7: v = (float)r.cost_b; break;
        case 8: v = (float)r.sel_cost; break;
        case 9: v = (float)r.theta_t[0]; break;
        case 10: v = (float)r.theta_t[1]; break;
        case 11: v = (float)r.theta_t[2]; break;
        case 12: v = (float)r.n_plane; break;
        case 13: v = (float)r.n_black; break;
        case 14: v = (float)r.n_white; break;
        case 15: v = (float)r.basin_margin; break;
        case 16: v = (float)tag; break;
        case 18: v = (float)r.flags; break;
        case 19: v = (float)r.n_roi; break;
        default: v = 0.f;   // 17 (check) is written below
      }
    }
    if (k != 17u) o[k] = v;
  }
#pragma unroll
  for (int of = ILCC_WAVE / 2; of > 0; of >>= 1) x ^= __shfl_xor(x, of, ILCC_WAVE);
  if ((threadIdx.x & (ILCC_WAVE - 1)) == 0) s_x[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = s_x[0] ^ s_x[1] ^ (tag * 0x9E3779B1u);
    t = (t ^ (t >> 24)) & 0xFFFFFFu;
    o[17] = (float)t;
  }
}

// Device -> pinned HOST memory by a kernel's own stores (the staging buffers are hipHostMalloc'ed: mapped, fine-grained).  Round 6:
// a batch's result copies used to be hipMemcpyAsync(D2H) commands queued behind its kernels at submit time; the SDMA engine that
// also carries the NEXT batches' 472 MB input copies then sat on each of them until that batch's kernels had finished
// (tools/dev_h2d_probe.py: the H2D-inclusive pipeline moved 52.0 GB/s with kernels running against 56.3 GB/s with the kernels
// returning early) -- with the records written by the GPU itself the SDMA queue holds input copies only.
__global__ __launch_bounds__(256) void k_store_to_host(const uint32_t* __restrict__ src, uint32_t* __restrict__ host_dst, uint32_t n_words) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) host_dst[i] = src[i];
}
void launch_store_to_host(const void* d_src, void* h_dst, size_t bytes, hipStream_t s) {   // bytes: a multiple of 4
  const uint32_t n_words = (uint32_t)(bytes / 4);
  if (n_words == 0) return;
  const uint32_t blocks = std::min<uint32_t>(256u, (n_words + 255u) / 256u);
  hipLaunchKernelGGL(k_store_to_host, dim3(blocks), dim3(256), 0, s, static_cast<const uint32_t*>(d_src), static_cast<uint32_t*>(h_dst), n_words);
}

void launch_pack_records(const ilcc_result* d_res, uint32_t n_frames, uint32_t n_corners, uint32_t tag_base, float* d_out,
                         hipStream_t s) {
  if (n_frames) hipLaunchKernelGGL(k9_pack_records, dim3(n_frames), dim3(128), 0, s, d_res, n_corners, tag_base, d_out);
}

}  // namespace ilcc

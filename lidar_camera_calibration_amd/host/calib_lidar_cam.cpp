// calib_lidar_cam.cpp -- see include/ilcc_calib.h.  Host-only C++; cites /root/reference/ilcc2/.
#include "ilcc_calib.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "ilcc_hip.h"   // ilcc_read_lidar_corners

namespace {

typedef std::array<double, 3> V3;

// ------------------------------------------------------------------ small dense helpers
void mat4_mul(const double A[16], const double B[16], double C[16]) {
  double R[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
      R[4 * i + j] = s;
    }
  std::memcpy(C, R, sizeof(R));
}

// Rodrigues: R = exp([r]x)
void rodrigues(const double r[3], double R[9]) {
  const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double th = std::sqrt(th2);
  double a, b;   // R = I + a K + b K^2
  if (th2 > 1e-16) {
    a = std::sin(th) / th;
    b = (1.0 - std::cos(th)) / th2;
  } else {
    a = 1.0;
    b = 0.5;
  }
  const double K[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  double K2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
      K2[3 * i + j] = s;
    }
  for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
  R[0] += 1;
  R[4] += 1;
  R[8] += 1;
}

// d(exp([r]x) X)/dr = -R [X]x Jr(r),  Jr = I - b K + c K^2,  b = (1-cos)/th^2, c = (th - sin)/th^3
void rotated_point_jacobian(const double r[3], const double R[9], const V3& X, double J[9]) {
  const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double th = std::sqrt(th2);
  double b, c;
  if (th2 > 1e-16) {
    b = (1.0 - std::cos(th)) / th2;
    c = (th - std::sin(th)) / (th2 * th);
  } else {
    b = 0.5;
    c = 1.0 / 6.0;
  }
  const double K[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  double K2[9], Jr[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
      K2[3 * i + j] = s;
    }
  for (int i = 0; i < 9; ++i) Jr[i] = -b * K[i] + c * K2[i];
  Jr[0] += 1;
  Jr[4] += 1;
  Jr[8] += 1;
  const double Xx[9] = {0, -X[2], X[1], X[2], 0, -X[0], -X[1], X[0], 0};
  double RX[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += R[3 * i + k] * Xx[3 * k + j];
      RX[3 * i + j] = s;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += RX[3 * i + k] * Jr[3 * k + j];
      J[3 * i + j] = -s;
    }
}

// ------------------------------------------------------------------ the problem (Pose3d2dError + HuberLoss(0.1))
struct PoseProblem {
  const double* p3;
  const double* p2;
  int n;
  double fx, cx, fy, cy;

  // cost = 1/2 sum rho(|res|^2); if r/J given: corrected residuals (2n) and Jacobian (2n x 6, row-major)
  double eval(const double x[6], double* res, double* J) const {
    double R[9];
    rodrigues(x, R);
    const double a = 0.1, b2 = a * a;
    double cost = 0;
    for (int i = 0; i < n; ++i) {
      const V3 X = {p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]};
      double p[3];
      for (int k = 0; k < 3; ++k) p[k] = R[3 * k] * X[0] + R[3 * k + 1] * X[1] + R[3 * k + 2] * X[2] + x[3 + k];
      const double iz = 1.0 / p[2];
      const double u = fx * p[0] * iz + cx, v = fy * p[1] * iz + cy;
      const double r0 = p2[2 * i] - u, r1 = p2[2 * i + 1] - v;   // observation - prediction (Optimization.h:176-177)
      const double s = r0 * r0 + r1 * r1;
      double rho0, rho1;
      if (s > b2) {
        const double rr = std::sqrt(s);
        rho0 = 2 * a * rr - b2;
        rho1 = a / rr;
      } else {
        rho0 = s;
        rho1 = 1;
      }
      cost += 0.5 * rho0;
      if (res) {
        const double sr = std::sqrt(rho1);
        res[2 * i] = sr * r0;
        res[2 * i + 1] = sr * r1;
        if (J) {
          double Jp[9];
          rotated_point_jacobian(x, R, X, Jp);   // dp/dr ; dp/dt = I
          // du/dp, dv/dp
          const double du[3] = {fx * iz, 0, -fx * p[0] * iz * iz};
          const double dv[3] = {0, fy * iz, -fy * p[1] * iz * iz};
          double* j0 = J + 12 * i;
          double* j1 = j0 + 6;
          for (int c = 0; c < 3; ++c) {
            j0[c] = -sr * (du[0] * Jp[c] + du[1] * Jp[3 + c] + du[2] * Jp[6 + c]);
            j1[c] = -sr * (dv[0] * Jp[c] + dv[1] * Jp[3 + c] + dv[2] * Jp[6 + c]);
            j0[3 + c] = -sr * du[c];
            j1[3 + c] = -sr * dv[c];
          }
        }
      }
    }
    return cost;
  }
};

// ------------------------------------------------------------------ Ceres-1.14-style trust region, N = 6
// TRUST_REGION + DOGLEG (SUBSPACE_DOGLEG) + DENSE_NORMAL_CHOLESKY with Ceres' defaults (the options
// the reference leaves untouched, Optimization.cpp:55-66): max 50 iterations, function_tolerance 1e-6,
// gradient_tolerance 1e-10, parameter_tolerance 1e-8, radius 1e4, jacobi scaling, monotonic steps.
constexpr int N = 6;

bool chol_solve(const double A_in[N * N], const double b[N], double x[N]) {
  double L[N][N] = {};
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A_in[N * i + j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i][i] = std::sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double y[N];
  for (int i = 0; i < N; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = N - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < N; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  for (int i = 0; i < N; ++i)
    if (!std::isfinite(x[i])) return false;
  return true;
}

double norm(const double* v, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return std::sqrt(s);
}

// argmin of 1/2 y'By + g'y on |y| = radius (B sym. PSD 2x2): eigen-decomposition + secular Newton
void min_on_circle(const double B[4], const double g[2], double radius, double y[2]) {
  const double d = 0.5 * (B[0] - B[3]), e = B[1];
  const double h = std::sqrt(d * d + e * e), mean = 0.5 * (B[0] + B[3]);
  const double l1 = mean - h, l2 = mean + h;
  double v2x, v2y;
  if (h == 0.0) { v2x = 1; v2y = 0; }
  else if (d >= 0) { v2x = d + h; v2y = e; }
  else { v2x = e; v2y = h - d; }
  const double nv = std::sqrt(v2x * v2x + v2y * v2y);
  if (nv > 0) { v2x /= nv; v2y /= nv; } else { v2x = 1; v2y = 0; }
  const double v1x = -v2y, v1y = v2x;
  const double g1 = v1x * g[0] + v1y * g[1], g2 = v2x * g[0] + v2y * g[1];
  const double gn = std::sqrt(g1 * g1 + g2 * g2);
  double lo = std::max(0.0, -l1);
  lo = std::max(lo, gn / radius - l2);
  const double hi = gn / radius - l1;
  double lam = lo;
  if (!(l1 + lam > 0.0)) lam = lo + 1e-12 * std::max(1.0, std::fabs(hi));
  for (int it = 0; it < 60; ++it) {
    const double a1 = l1 + lam, a2 = l2 + lam;
    const double y1 = -g1 / a1, y2 = -g2 / a2;
    const double ny = std::sqrt(y1 * y1 + y2 * y2);
    const double qq = g1 * g1 / (a1 * a1 * a1) + g2 * g2 / (a2 * a2 * a2);
    if (!(qq > 0.0) || !std::isfinite(ny)) break;
    double nl = lam + (ny * ny / qq) * ((ny - radius) / radius);
    if (!(l1 + nl > 0.0)) nl = 0.5 * (lam + std::max(0.0, -l1));
    const bool stop = std::fabs(nl - lam) <= 1e-15 * std::max(1.0, std::fabs(nl));
    lam = nl;
    if (stop) break;
  }
  const double a1 = l1 + lam, a2 = l2 + lam;
  double y1 = (a1 > 0) ? -g1 / a1 : 0, y2 = (a2 > 0) ? -g2 / a2 : 0;
  double ny = std::sqrt(y1 * y1 + y2 * y2);
  if (ny < radius * (1 - 1e-9) && !(a1 > 1e-300 * std::max(1.0, l2))) {
    y1 = std::sqrt(std::max(0.0, radius * radius - y2 * y2));
    ny = radius;
  }
  if (ny > 0) { y1 *= radius / ny; y2 *= radius / ny; }
  y[0] = v1x * y1 + v2x * y2;
  y[1] = v1y * y1 + v2y * y2;
}

int trust_region_minimize(const PoseProblem& q, double x[N], double* final_cost) {
  const int m = 2 * q.n;
  std::vector<double> r(m), J((size_t)m * N);
  double x_cost = q.eval(x, r.data(), J.data());
  double x_norm = norm(x, N);
  double grad[N], scale[N];
  auto gradient = [&]() {
    for (int c = 0; c < N; ++c) grad[c] = 0;
    for (int k = 0; k < m; ++k)
      for (int c = 0; c < N; ++c) grad[c] += J[(size_t)k * N + c] * r[k];
  };
  gradient();
  for (int c = 0; c < N; ++c) {
    double sq = 0;
    for (int k = 0; k < m; ++k) sq += J[(size_t)k * N + c] * J[(size_t)k * N + c];
    scale[c] = 1.0 / (1.0 + std::sqrt(sq));
  }
  auto scale_jac = [&]() {
    for (int k = 0; k < m; ++k)
      for (int c = 0; c < N; ++c) J[(size_t)k * N + c] *= scale[c];
  };
  scale_jac();

  double radius = 1e4, mu = 1e-8, step_norm = 0;
  bool reuse = false;
  double diag[N], g[N], gn[N], JtJ[N * N], Jtr[N];
  bool one_dim = false;
  double basis[N][2], sg[2], sB[4];
  int iter = 0, invalid = 0;
  for (;;) {
    if (iter >= 50) break;
    double gmax = 0;
    for (int c = 0; c < N; ++c) gmax = std::max(gmax, std::fabs(grad[c]));
    if (gmax <= 1e-10 || radius <= 1e-32) break;
    ++iter;
    double step[N];
    bool valid = true;
    if (!reuse) {
      reuse = true;
      for (int a = 0; a < N; ++a) {
        Jtr[a] = 0;
        for (int b = 0; b < N; ++b) JtJ[N * a + b] = 0;
      }
      for (int k = 0; k < m; ++k)
        for (int a = 0; a < N; ++a) {
          Jtr[a] += J[(size_t)k * N + a] * r[k];
          for (int b = 0; b < N; ++b) JtJ[N * a + b] += J[(size_t)k * N + a] * J[(size_t)k * N + b];
        }
      for (int c = 0; c < N; ++c) {
        diag[c] = std::sqrt(std::min(std::max(JtJ[N * c + c], 1e-6), 1e32));
        g[c] = Jtr[c] / diag[c];
      }
      bool ok = false;
      while (mu < 1.0) {
        double A[N * N];
        std::memcpy(A, JtJ, sizeof(A));
        for (int c = 0; c < N; ++c) A[N * c + c] += diag[c] * diag[c] * mu;
        if (chol_solve(A, Jtr, gn)) { ok = true; break; }
        mu *= 10.0;
      }
      if (!ok) valid = false;
      if (valid) {
        for (int c = 0; c < N; ++c) gn[c] *= -diag[c];
        // subspace spanned by the scaled gradient and the Gauss-Newton step
        const double n0 = norm(g, N), n1 = norm(gn, N);
        const double* first = (n0 >= n1) ? g : gn;
        const double* second = (n0 >= n1) ? gn : g;
        const double nf = std::max(n0, n1);
        double v0[N], v1[N], dot = 0;
        for (int c = 0; c < N; ++c) { v0[c] = first[c] / nf; }
        for (int c = 0; c < N; ++c) dot += second[c] * v0[c];
        for (int c = 0; c < N; ++c) v1[c] = second[c] - dot * v0[c];
        const double nr = norm(v1, N);
        one_dim = !(nr > N * 2.220446049250313e-16 * nf);
        if (!one_dim) {
          double u[2][N];
          for (int c = 0; c < N; ++c) {
            v1[c] /= nr;
            basis[c][0] = v0[c];
            basis[c][1] = v1[c];
            u[0][c] = v0[c] / diag[c];
            u[1][c] = v1[c] / diag[c];
          }
          for (int a = 0; a < 2; ++a) {
            sg[a] = 0;
            for (int c = 0; c < N; ++c) sg[a] += basis[c][a] * g[c];
            for (int b = 0; b < 2; ++b) {
              double acc = 0;
              for (int c = 0; c < N; ++c)
                for (int d = 0; d < N; ++d) acc += u[a][c] * JtJ[N * c + d] * u[b][d];
              sB[2 * a + b] = acc;
            }
          }
        }
      }
    }
    if (valid) {
      const double gnn = norm(gn, N);
      if (gnn <= radius) {
        for (int c = 0; c < N; ++c) step[c] = gn[c] / diag[c];
        step_norm = gnn;
      } else if (one_dim) {
        const double gnrm = norm(g, N);
        for (int c = 0; c < N; ++c) step[c] = -(radius / gnrm) * g[c] / diag[c];
        step_norm = radius;
      } else {
        double y2[2];
        min_on_circle(sB, sg, radius, y2);
        for (int c = 0; c < N; ++c) step[c] = (basis[c][0] * y2[0] + basis[c][1] * y2[1]) / diag[c];
        step_norm = radius;
      }
    }
    double model_cost_change = 0;
    if (valid) {
      for (int k = 0; k < m; ++k) {
        double mr = 0;
        for (int c = 0; c < N; ++c) mr += J[(size_t)k * N + c] * step[c];
        model_cost_change -= mr * (r[k] + mr / 2.0);
      }
      valid = model_cost_change > 0.0;
    }
    if (!valid) {
      if (++invalid >= 5) break;
      mu *= 10.0;
      reuse = false;
      continue;
    }
    invalid = 0;
    double cand[N];
    for (int c = 0; c < N; ++c) cand[c] = x[c] + step[c] * scale[c];
    const double cand_cost = q.eval(cand, nullptr, nullptr);
    double dn = 0;
    for (int c = 0; c < N; ++c) dn += (x[c] - cand[c]) * (x[c] - cand[c]);
    if (std::sqrt(dn) <= 1e-8 * (x_norm + 1e-8)) break;
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) break;
    const double rel = cost_change / model_cost_change;
    if (rel > 1e-3) {
      std::memcpy(x, cand, sizeof(cand));
      x_norm = norm(x, N);
      x_cost = q.eval(x, r.data(), J.data());
      gradient();
      scale_jac();
      if (rel < 0.25) radius *= 0.5;
      if (rel > 0.75) radius = std::max(radius, 3.0 * step_norm);
      radius = std::min(radius, 1e16);
      mu = std::max(1e-8, 2.0 * mu / 10.0);
      reuse = false;
    } else {
      radius *= 0.5;
      reuse = true;
    }
  }
  if (final_cost) *final_cost = x_cost;
  return iter;
}

void swap_rows3(double* xyz, int w, int front, int end) {
  for (int k = 0; k < w; ++k)
    for (int c = 0; c < 3; ++c) std::swap(xyz[3 * (front + k) + c], xyz[3 * (end + k) + c]);
}

}  // namespace

extern "C" {

// src/ImageCornersEst.cpp:213-279
int32_t ilcc_read_cam_corners(const char* filename, int32_t num, int32_t board_w, int32_t board_h, double* out_xy) {
  (void)board_w;
  std::ifstream in(filename);
  if (!in.is_open()) return -1;
  std::vector<std::vector<std::array<double, 2>>> rows;
  std::string line;
  int counter = 0;
  while (std::getline(in, line)) {   // X block
    std::istringstream ss(line);
    std::vector<std::array<double, 2>> row;
    double v;
    while (ss >> v) {
      row.push_back({v, 0.0});
      ++counter;
    }
    rows.push_back(row);
    if (counter >= num) break;
  }
  size_t ri = 0;
  while (std::getline(in, line) && ri < rows.size()) {   // Y block
    std::istringstream ss(line);
    double v;
    size_t ci = 0;
    while (ss >> v && ci < rows[ri].size()) rows[ri][ci++][1] = v;
    ++ri;
  }
  if (rows.empty()) return 0;
  int32_t n = 0;
  if ((int32_t)rows.size() != board_h) {   // :262-266  column-major over the file's matrix
    for (size_t w = 0; w < rows[0].size(); ++w)
      for (size_t h = 0; h < rows.size(); ++h)
        if (w < rows[h].size() && n < num) {
          out_xy[2 * n] = rows[h][w][0];
          out_xy[2 * n + 1] = rows[h][w][1];
          ++n;
        }
  } else {                                  // :268-273
    for (size_t h = 0; h < rows.size(); ++h)
      for (size_t w = 0; w < rows[0].size(); ++w)
        if (w < rows[h].size() && n < num) {
          out_xy[2 * n] = rows[h][w][0];
          out_xy[2 * n + 1] = rows[h][w][1];
          ++n;
        }
  }
  return n;
}

// test/calib_lidar_cam.cpp:50-69 (the constants 1.57 / 3.14 are the reference's)
int32_t ilcc_lidar2cam_axis_roughly(const char* camera_name, double T[16]) {
  auto rot = [](char axis, double ang, double R[16]) {
    const double c = std::cos(ang), s = std::sin(ang);
    const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(R, I, sizeof(I));
    if (axis == 'x') { R[5] = c; R[6] = -s; R[9] = s; R[10] = c; }
    if (axis == 'y') { R[0] = c; R[2] = s; R[8] = -s; R[10] = c; }
    if (axis == 'z') { R[0] = c; R[1] = -s; R[4] = s; R[5] = c; }
  };
  const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(T, I, sizeof(I));
  const std::string cam = camera_name ? camera_name : "";
  double A[16], B[16];
  if (cam == "front" || cam == "car_left" || cam == "pointgrey") {
    rot('y', -1.57, A);
    rot('x', 1.57, B);
    mat4_mul(A, B, T);
  } else if (cam == "left") {
    rot('x', 1.57, T);
  } else if (cam == "right") {
    rot('x', 1.57, A);
    rot('z', 3.14, B);
    mat4_mul(A, B, T);
  } else if (cam == "back") {
    rot('y', 1.57, A);
    rot('x', 1.57, B);
    mat4_mul(A, B, T);
  } else {
    return 0;
  }
  return 1;
}

// src/ImageCornersEst.cpp:461-488
void ilcc_check_order_lidar(double* p, int32_t w, int32_t h) {
  if (p[3 * 0 + 1] > p[3 * (w + 1) + 1])
    for (int r = 0; r < h / 2; ++r) swap_rows3(p, w, w * r, w * (h - 1 - r));
  if (p[3 * 0 + 0] > p[3 * 1 + 0])
    for (int r = 0; r < h; ++r)
      for (int k = 0; k < w / 2; ++k)
        for (int c = 0; c < 3; ++c) std::swap(p[3 * (w * r + k) + c], p[3 * (w * r + w - 1 - k) + c]);
}

// src/ImageCornersEst.cpp:430-459
void ilcc_check_order_cam(double* p, int32_t w, int32_t h) {
  if (p[2 * 0 + 1] > p[2 * (w + 1) + 1])
    for (int r = 0; r < h / 2; ++r)
      for (int k = 0; k < w; ++k)
        for (int c = 0; c < 2; ++c) std::swap(p[2 * (w * r + k) + c], p[2 * (w * (h - 1 - r) + k) + c]);
  if (p[2 * 0 + 0] > p[2 * 1 + 0])
    for (int r = 0; r < h; ++r)
      for (int k = 0; k < w / 2; ++k)
        for (int c = 0; c < 2; ++c) std::swap(p[2 * (w * r + k) + c], p[2 * (w * r + w - 1 - k) + c]);
}

int32_t ilcc_solve_pose_3d2d(const double* pts3d, const double* pts2d, int32_t n, const double camera[4],
                             double r[3], double t[3], double* final_cost) {
  if (!pts3d || !pts2d || n < 3 || !camera || !r || !t) return -1;
  PoseProblem q{pts3d, pts2d, n, camera[0], camera[1], camera[2], camera[3]};
  double x[6] = {r[0], r[1], r[2], t[0], t[1], t[2]};
  const int it = trust_region_minimize(q, x, final_cost);
  for (int c = 0; c < 3; ++c) {
    r[c] = x[c];
    t[c] = x[3 + c];
  }
  return it;
}

// src/ImageCornersEst.cpp:301-306: the raw bytes of an Eigen::Matrix4d (column-major)
int32_t ilcc_extrinsic_write(const char* filename, const double T[16]) {
  double cm[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) cm[4 * j + i] = T[4 * i + j];
  std::ofstream out(filename, std::ios_base::binary);
  if (!out.is_open()) return -1;
  out.write(reinterpret_cast<const char*>(cm), sizeof(cm));
  return out.good() ? 0 : -1;
}

int32_t ilcc_extrinsic_read(const char* filename, double T[16]) {
  double cm[16];
  std::ifstream in(filename, std::ios_base::binary);
  if (!in.is_open()) return -1;
  in.read(reinterpret_cast<char*>(cm), sizeof(cm));
  if (in.gcount() != (std::streamsize)sizeof(cm)) return -1;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T[4 * i + j] = cm[4 * j + i];
  return 0;
}

// test/calib_lidar_cam.cpp:103-165 without the image display
int32_t ilcc_calib_lidar_cam(const char* dir, const char* camera_name, int32_t bag_num, int32_t board_w,
                             int32_t board_h, const double camera[4], double T_lidar2cam[16], double* mean_reproj_px) {
  const int corner_num = board_w * board_h;
  double Trough[16];
  ilcc_lidar2cam_axis_roughly(camera_name, Trough);
  std::vector<double> all3, all2;
  for (int idx = 1; idx <= bag_num; ++idx) {
    const std::string base = std::string(dir) + "/" + camera_name;
    std::vector<double> p3((size_t)3 * corner_num), p2((size_t)2 * corner_num);
    const int n3 = ilcc_read_lidar_corners((base + "_lidar_" + std::to_string(idx) + ".txt").c_str(), corner_num, p3.data());
    const int n2 = ilcc_read_cam_corners((base + std::to_string(idx) + ".txt").c_str(), corner_num, board_w, board_h, p2.data());
    if (n3 != corner_num || n2 != corner_num) return -idx;
    for (int k = 0; k < corner_num; ++k) {   // :119-120 rough axis alignment
      const double X[3] = {p3[3 * k], p3[3 * k + 1], p3[3 * k + 2]};
      for (int c = 0; c < 3; ++c)
        p3[3 * k + c] = Trough[4 * c] * X[0] + Trough[4 * c + 1] * X[1] + Trough[4 * c + 2] * X[2] + Trough[4 * c + 3];
    }
    ilcc_check_order_lidar(p3.data(), board_w, board_h);   // :122
    ilcc_check_order_cam(p2.data(), board_w, board_h);     // :123
    all3.insert(all3.end(), p3.begin(), p3.end());
    all2.insert(all2.end(), p2.begin(), p2.end());
  }
  double r[3] = {0, 0, 0}, t[3] = {0, 0, 0}, cost = 0;   // :152-153
  const int n = (int)all2.size() / 2;
  if (ilcc_solve_pose_3d2d(all3.data(), all2.data(), n, camera, r, t, &cost) < 0) return -1000;
  double R[9];
  rodrigues(r, R);
  const double Tm[16] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2], 0, 0, 0, 1};
  mat4_mul(Tm, Trough, T_lidar2cam);   // :162  T_lidar2cam = T * T_axis_roughly
  if (mean_reproj_px) {
    double acc = 0;
    for (int k = 0; k < n; ++k) {
      double p[3];
      for (int c = 0; c < 3; ++c)
        p[c] = R[3 * c] * all3[3 * k] + R[3 * c + 1] * all3[3 * k + 1] + R[3 * c + 2] * all3[3 * k + 2] + t[c];
      const double u = camera[0] * p[0] / p[2] + camera[1], v = camera[2] * p[1] / p[2] + camera[3];
      acc += std::hypot(u - all2[2 * k], v - all2[2 * k + 1]);
    }
    *mean_reproj_px = acc / n;
  }
  return 0;
}

}  // extern "C"

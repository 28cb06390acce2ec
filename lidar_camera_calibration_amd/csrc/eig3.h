// eig3.h -- 3x3 symmetric eigen-decomposition (cyclic Jacobi, double), ascending eigenvalues.
// Stands in for Eigen::SelfAdjointEigenSolver<Matrix3f> (LidarCornersEst.cpp:337-339) and for
// pcl::eigen33 inside SACSegmentation's coefficient refinement.  Single-thread device code.
#pragma once
#include <hip/hip_runtime.h>

namespace ilcc {

// a: row-major symmetric 3x3.  w[c] ascending, v[c][*] = unit eigenvector c.
__device__ inline void eig3_sym(const double a_in[9], double w[3], double v[3][3]) {
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
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k][pp], akq = a[k][qq];
          a[k][pp] = c * akp - s * akq;
          a[k][qq] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
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
  const double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (d[order[j]] > d[order[j + 1]]) {
        const int t = order[j];
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

}  // namespace ilcc

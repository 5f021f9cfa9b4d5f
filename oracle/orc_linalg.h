/*
 * oracle/orc_linalg.h -- fixed-size linear algebra kit for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it (as the checker / reported baseline).
 *
 * The reference (matiable/pointcloud-slam) delegates these pieces to Eigen
 * (version un-pinned: fast_gicp/.gitmodules:4-6, vendored copy lacks
 * Eigen/Core), so they are restated here from Eigen's published algorithms:
 *   - Quaternion::toRotationMatrix           (used by so3_exp call sites,
 *       lsq_registration_impl.hpp:114-116,139-141; Eigen/src/Geometry/Quaternion.h)
 *   - LDLT, ColPivHouseholderQR, JacobiSVD, SelfAdjointEigenSolver (compute and computeDirect), Matrix3 / Matrix4d
 *     inverse: orc_eigen.h, each function citing the in-tree Eigen file and lines it follows
 * All matrices crossing function boundaries are ROW-MAJOR.
 */
#ifndef ORC_LINALG_H
#define ORC_LINALG_H

#include <math.h>
#include <string.h>
#include <float.h>

/* ---- Isometry3d as a row-major 4x4 with last row (0,0,0,1) -------------- */
static inline void orc_iso_identity(double T[16]) {
  memset(T, 0, 16 * sizeof(double));
  T[0] = T[5] = T[10] = T[15] = 1.0;
}

/* C = A * B for affine 4x4 (Eigen::Isometry3d product: last row stays 0,0,0,1) */
static inline void orc_iso_mul(const double A[16], const double B[16], double C[16]) {
  double R[16];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      R[i * 4 + j] = A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j] + A[i * 4 + 2] * B[2 * 4 + j];
    }
    R[i * 4 + 3] = A[i * 4 + 0] * B[3] + A[i * 4 + 1] * B[7] + A[i * 4 + 2] * B[11] + A[i * 4 + 3];
  }
  R[12] = R[13] = R[14] = 0.0;
  R[15] = 1.0;
  memcpy(C, R, sizeof(R));
}

/* so3_exp: fast_gicp/include/fast_gicp/so3/so3.hpp:58-77 (quaternion w,x,y,z) */
static inline void orc_so3_exp(const double omega[3], double q_wxyz[4]) {
  double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  double imag_factor, real_factor;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    double theta = sqrt(theta_sq);
    double half_theta = 0.5 * theta;
    imag_factor = sin(half_theta) / theta;
    real_factor = cos(half_theta);
  }
  q_wxyz[0] = real_factor;
  q_wxyz[1] = imag_factor * omega[0];
  q_wxyz[2] = imag_factor * omega[1];
  q_wxyz[3] = imag_factor * omega[2];
}

/* Eigen::Quaternion::toRotationMatrix (no normalisation), row-major 3x3 out */
static inline void orc_quat_to_rot(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* delta = [so3_exp(d[0..2]) | d[3..5]]  (lsq_registration_impl.hpp:114-116) */
static inline void orc_delta_from_d(const double d[6], double delta[16]) {
  double q[4], R[9];
  orc_so3_exp(d, q);
  orc_quat_to_rot(q, R);
  orc_iso_identity(delta);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) delta[i * 4 + j] = R[i * 3 + j];
    delta[i * 4 + 3] = d[3 + i];
  }
}

/*
 * Everything the reference delegates to Eigen's decompositions lives in orc_eigen.h, restated from the Eigen sources in
 * the reference tree (thirdparty/Eigen/Eigen/src) with line citations.  The old names stay as thin wrappers.
 */
#include "orc_eigen.h"

/* Eigen::LDLT<Matrix<double,6,6>>(A).solve(rhs)   lsq_registration_impl.hpp:111,136 */
static inline void orc_ldlt6_solve(const double A_in[36], const double rhs[6], double x[6]) { orc_eig_ldlt6_solve(A_in, rhs, x); }

/* Matrix3::inverse()   Eigen/src/LU/InverseImpl.h:125-176 */
static inline void orc_inv3d(const double m[9], double inv[9]) { orc_eig_inv3d(m, inv); }
static inline void orc_inv3f(const float m[9], float inv[9]) { orc_eig_inv3f(m, inv); }

/* ColPivHouseholderQR<Matrix<T,rows,3>>(A).solve(b)   common_lib.h:208,223 */
#define ORC_QR_MAXR ORC_EIG_QR_MAXR
static inline void orc_colpivqr3f(const float *A, int rows, const float *b, float x[3]) { orc_eig_colpivqr3f(A, rows, b, x); }
static inline void orc_colpivqr3d(const double *A, int rows, const double *b, double x[3]) { orc_eig_colpivqr3d(A, rows, b, x); }

#endif /* ORC_LINALG_H */

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
 *       lsq_registration_impl.hpp:114-116,139-141)
 *   - LDLT<Matrix6d> with diagonal pivoting  (lsq_registration_impl.hpp:111,136)
 *   - ColPivHouseholderQR::solve             (jueying_lio/include/common_lib.h:208,223)
 *   - Matrix3/4 inverse (cofactor / Gauss)   (fast_gicp_impl.hpp:149)
 *   - SelfAdjointEigenSolver 3x3 (Jacobi)    (covariance_regularization.cu:18-20,
 *       voxel_grid_covariance_omp_impl.hpp:333)
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
 * Eigen::LDLT<Matrix<double,6,6>>(A).solve(rhs): robust Cholesky with diagonal
 * pivoting (P A P^T = L D L^T), pivot = largest |diagonal| of the trailing
 * block, solve skips pivots with |D_ii| <= tiny (pseudo-inverse of D).
 */
static inline void orc_ldlt6_solve(const double A_in[36], const double rhs[6], double x[6]) {
  double A[36];
  int perm[6];
  memcpy(A, A_in, sizeof(A));
  /* Eigen's LDLT<.., Lower> reads the lower triangle only; the steps below move entries symmetrically, so start from the
   * self-adjoint completion of the lower triangle (matters for the float models, whose summed H is not exactly symmetric) */
  for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++) A[i * 6 + j] = A[j * 6 + i];
  for (int i = 0; i < 6; i++) perm[i] = i;
  /* in-place lower LDLT with symmetric pivoting (only the lower triangle is read) */
  for (int k = 0; k < 6; k++) {
    int p = k;
    double best = fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; i++) {
      if (fabs(A[i * 6 + i]) > best) { best = fabs(A[i * 6 + i]); p = i; }
    }
    if (p != k) {
      /* symmetric swap of rows/cols k and p on the full symmetric matrix */
      for (int j = 0; j < 6; j++) { double t = A[k * 6 + j]; A[k * 6 + j] = A[p * 6 + j]; A[p * 6 + j] = t; }
      for (int i = 0; i < 6; i++) { double t = A[i * 6 + k]; A[i * 6 + k] = A[i * 6 + p]; A[i * 6 + p] = t; }
      int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
    }
    double dk = A[k * 6 + k];
    if (dk == 0.0) continue;
    for (int i = k + 1; i < 6; i++) A[i * 6 + k] /= dk;
    for (int i = k + 1; i < 6; i++) {
      for (int j = k + 1; j <= i; j++) {
        A[i * 6 + j] -= A[i * 6 + k] * dk * A[j * 6 + k];
        A[j * 6 + i] = A[i * 6 + j];
      }
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < 6; i++) for (int j = 0; j < i; j++) y[i] -= A[i * 6 + j] * y[j];
  /* Eigen: tolerance = 1 / NumTraits<double>::highest() -> only exact ~0 pivots are dropped */
  for (int i = 0; i < 6; i++) y[i] = (fabs(A[i * 6 + i]) > DBL_MIN) ? y[i] / A[i * 6 + i] : 0.0;
  for (int i = 5; i >= 0; i--) for (int j = i + 1; j < 6; j++) y[i] -= A[j * 6 + i] * y[j];
  for (int i = 0; i < 6; i++) x[perm[i]] = y[i];
}

/* 3x3 inverse by cofactors (Eigen fixed-size inverse), row-major. */
#define ORC_DEF_INV3(NAME, T)                                                        \
  static inline void NAME(const T m[9], T inv[9]) {                                  \
    T c00 = m[4] * m[8] - m[5] * m[7];                                               \
    T c01 = m[5] * m[6] - m[3] * m[8];                                               \
    T c02 = m[3] * m[7] - m[4] * m[6];                                               \
    T det = m[0] * c00 + m[1] * c01 + m[2] * c02;                                    \
    T id = (T)1 / det;                                                               \
    inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id; \
    inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id; \
    inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id; \
  }
ORC_DEF_INV3(orc_inv3d, double)
ORC_DEF_INV3(orc_inv3f, float)

/*
 * Symmetric 3x3 eigen-decomposition, cyclic Jacobi (double).  Eigenvalues
 * ascending in w[], eigenvectors in the COLUMNS of V (row-major storage).
 * Stands in for Eigen::SelfAdjointEigenSolver / JacobiSVD of an SPD matrix.
 */
static inline void orc_eig3_sym(const double A_in[9], double w[3], double V[9]) {
  double A[9];
  memcpy(A, A_in, sizeof(A));
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++) {
      for (int q = p + 1; q < 3; q++) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
    }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
  /* sort ascending */
  for (int i = 0; i < 2; i++) {
    for (int j = 0; j < 2 - i; j++) {
      if (w[j] > w[j + 1]) {
        double t = w[j]; w[j] = w[j + 1]; w[j + 1] = t;
        for (int k = 0; k < 3; k++) { double u = V[k * 3 + j]; V[k * 3 + j] = V[k * 3 + j + 1]; V[k * 3 + j + 1] = u; }
      }
    }
  }
}

/*
 * Eigen::ColPivHouseholderQR<Matrix<T,rows,3>>(A).solve(b), rows <= ORC_QR_MAXR.
 * Column pivoting on LAPACK-style down-dated column norms, Householder
 * reflectors (makeHouseholderInPlace), rank from Eigen's default threshold.
 * A is row-major rows x 3, modified in place.  x[3] out.
 */
#define ORC_QR_MAXR 32
#define ORC_DEF_COLPIVQR(NAME, T, SQRT, FABS, EPS, TMIN)                                      \
  static inline void NAME(T *A, int rows, const T *b_in, T x[3]) {                            \
    const int cols = 3;                                                                       \
    int size = rows < cols ? rows : cols;                                                     \
    int perm[3] = {0, 1, 2};                                                                  \
    T hcoef[3] = {0, 0, 0};                                                                   \
    T norms_upd[3], norms_dir[3];                                                             \
    T c[ORC_QR_MAXR];                                                                         \
    T maxnorm = 0;                                                                            \
    for (int j = 0; j < cols; j++) {                                                          \
      T s = 0;                                                                                \
      for (int i = 0; i < rows; i++) s += A[i * 3 + j] * A[i * 3 + j];                        \
      norms_dir[j] = norms_upd[j] = SQRT(s);                                                  \
      if (norms_upd[j] > maxnorm) maxnorm = norms_upd[j];                                     \
    }                                                                                         \
    const T thr_helper = (maxnorm * EPS) * (maxnorm * EPS) / (T)rows;                         \
    const T downdate_thr = SQRT(EPS);                                                         \
    int nonzero = size;                                                                       \
    for (int k = 0; k < size; k++) {                                                          \
      int big = k;                                                                            \
      T bign = norms_upd[k];                                                                  \
      for (int j = k + 1; j < cols; j++) if (norms_upd[j] > bign) { bign = norms_upd[j]; big = j; } \
      T big_sq = bign * bign;                                                                 \
      if (nonzero == size && big_sq < thr_helper * (T)(rows - k)) nonzero = k;                \
      if (big != k) {                                                                         \
        for (int i = 0; i < rows; i++) { T t = A[i * 3 + k]; A[i * 3 + k] = A[i * 3 + big]; A[i * 3 + big] = t; } \
        T t = norms_upd[k]; norms_upd[k] = norms_upd[big]; norms_upd[big] = t;                \
        t = norms_dir[k]; norms_dir[k] = norms_dir[big]; norms_dir[big] = t;                  \
        int ti = perm[k]; perm[k] = perm[big]; perm[big] = ti;                                \
      }                                                                                       \
      /* makeHouseholderInPlace on A[k:rows, k] */                                            \
      T tail_sq = 0;                                                                          \
      for (int i = k + 1; i < rows; i++) tail_sq += A[i * 3 + k] * A[i * 3 + k];              \
      T c0 = A[k * 3 + k], beta, tau;                                                         \
      if (tail_sq <= TMIN) {                                                                  \
        tau = 0; beta = c0;                                                                   \
        for (int i = k + 1; i < rows; i++) A[i * 3 + k] = 0;                                  \
      } else {                                                                                \
        beta = SQRT(c0 * c0 + tail_sq);                                                       \
        if (c0 >= 0) beta = -beta;                                                            \
        T inv = c0 - beta;                                                                    \
        for (int i = k + 1; i < rows; i++) A[i * 3 + k] /= inv;                               \
        tau = (beta - c0) / beta;                                                             \
      }                                                                                       \
      A[k * 3 + k] = beta;                                                                    \
      hcoef[k] = tau;                                                                         \
      /* applyHouseholderOnTheLeft to the trailing columns */                                 \
      for (int j = k + 1; j < cols; j++) {                                                    \
        if (rows - k == 1) { A[k * 3 + j] *= ((T)1 - tau); continue; }                        \
        if (tau == 0) continue;                                                               \
        T tmp = 0;                                                                            \
        for (int i = k + 1; i < rows; i++) tmp += A[i * 3 + k] * A[i * 3 + j];                \
        tmp += A[k * 3 + j];                                                                  \
        A[k * 3 + j] -= tau * tmp;                                                            \
        for (int i = k + 1; i < rows; i++) A[i * 3 + j] -= tau * A[i * 3 + k] * tmp;          \
      }                                                                                       \
      /* column-norm down-date */                                                             \
      for (int j = k + 1; j < cols; j++) {                                                    \
        if (norms_upd[j] != 0) {                                                              \
          T temp = FABS(A[k * 3 + j]) / norms_upd[j];                                         \
          temp = ((T)1 + temp) * ((T)1 - temp);                                               \
          if (temp < 0) temp = 0;                                                             \
          T r = norms_upd[j] / norms_dir[j];                                                  \
          T temp2 = temp * r * r;                                                             \
          if (temp2 <= downdate_thr) {                                                        \
            T s = 0;                                                                          \
            for (int i = k + 1; i < rows; i++) s += A[i * 3 + j] * A[i * 3 + j];              \
            norms_dir[j] = norms_upd[j] = SQRT(s);                                            \
          } else {                                                                            \
            norms_upd[j] *= SQRT(temp);                                                       \
          }                                                                                   \
        }                                                                                     \
      }                                                                                       \
    }                                                                                         \
    /* c = Q^T b : apply H_0, H_1, ... in order */                                            \
    for (int i = 0; i < rows; i++) c[i] = b_in[i];                                            \
    for (int k = 0; k < nonzero; k++) {                                                       \
      T tau = hcoef[k];                                                                       \
      if (rows - k == 1) { c[k] *= ((T)1 - tau); continue; }                                  \
      if (tau == 0) continue;                                                                 \
      T tmp = 0;                                                                              \
      for (int i = k + 1; i < rows; i++) tmp += A[i * 3 + k] * c[i];                          \
      tmp += c[k];                                                                            \
      c[k] -= tau * tmp;                                                                      \
      for (int i = k + 1; i < rows; i++) c[i] -= tau * A[i * 3 + k] * tmp;                    \
    }                                                                                         \
    /* back substitution on the nonzero x nonzero upper triangle */                           \
    for (int i = nonzero - 1; i >= 0; i--) {                                                  \
      T s = c[i];                                                                             \
      for (int j = i + 1; j < nonzero; j++) s -= A[i * 3 + j] * c[j];                         \
      c[i] = s / A[i * 3 + i];                                                                \
    }                                                                                         \
    for (int i = 0; i < 3; i++) x[i] = 0;                                                     \
    for (int i = 0; i < nonzero; i++) x[perm[i]] = c[i];                                      \
  }
ORC_DEF_COLPIVQR(orc_colpivqr3f, float, sqrtf, fabsf, FLT_EPSILON, FLT_MIN)
ORC_DEF_COLPIVQR(orc_colpivqr3d, double, sqrt, fabs, DBL_EPSILON, DBL_MIN)

#endif /* ORC_LINALG_H */

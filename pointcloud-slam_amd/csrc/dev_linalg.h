// dev_linalg.h -- the small dense decompositions the reference delegates to Eigen, for device and host code.
//
// Each routine follows an Eigen source file that is in the reference tree
//   E = /root/reference/src/pointcloud_match/fast_gicp/thirdparty/Eigen/Eigen/src
// (line ranges below).  Eigen/Core is not in that tree; where a result depends on the order of additions inside Core's
// reductions the order of upstream Eigen 3.4 on SSE2 (the reference's x86 builds) is used -- DESIGN.md section 5 lists
// these assumptions (CORE-1..3) once.
//
//   jacobi_svd<N>       JacobiSVD<Matrix<double,N,N>>, FullU | FullV   E/SVD/JacobiSVD.h:667-797, E/misc/RealSvd2x2.h:19-51,
//                                                                       E/Jacobi/Jacobi.h:92-125 (makeJacobi), :329-340 (rotation)
//   svd_solve6          JacobiSVD<Matrix6d>::solve                      E/SVD/SVDBase.h:148-157 (rank), :198-205, :308-318
//   selfadjoint3        SelfAdjointEigenSolver<Matrix3d>::compute       E/Eigenvalues/SelfAdjointEigenSolver.h:412-462, :498-566, :838-895,
//                                                                       E/Eigenvalues/Tridiagonalization.h:464-503, E/Jacobi/Jacobi.h:228-262 (makeGivens)
//   selfadjoint3_direct SelfAdjointEigenSolver<Matrix3f>::computeDirect E/Eigenvalues/SelfAdjointEigenSolver.h:577-733
//   inv3<T>             Matrix3::inverse()                              E/LU/InverseImpl.h:125-176
//   inv4d               Matrix4d::inverse(), Packet2d form              E/LU/arch/InverseSize4.h:166-351
//
// Call sites in the reference: fast_gicp_impl.hpp:273 (3x3 SVD of the neighbourhood covariance), gicp_omp_impl.hpp:110,
// ndt_omp_impl.hpp:112-114 (Newton step), voxel_grid_covariance_omp_impl.hpp:327-349 (leaf eigen-decomposition),
// covariance_regularization.cu:18-20,57-58,84-85 (computeDirect), fast_gicp_impl.hpp:146-150 (4x4 inverse).
// Matrices are ROW-MAJOR arrays.
#pragma once

#include <float.h>
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PCM_LA __host__ __device__ inline
#define PCM_UNROLL _Pragma("unroll")
#else
#define PCM_LA inline
#define PCM_UNROLL
#endif

namespace pcm {

// correctly rounded float division / square root on the device as on the host (the closed-form float eigen-solver below
// feeds discrete-looking consequences: eigenvector choices, voxel covariances compared at 1e-9)
PCM_LA float la_divf(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fdiv_rn(a, b);
#else
  return a / b;
#endif
}

// x' = c x + s y ; y' = -s x + c y   (apply_rotation_in_the_plane, Jacobi.h:329-340)
template <typename T>
PCM_LA void plane_rot(T& x, T& y, T c, T s) {
  const T xi = x, yi = y;
  x = c * xi + s * yi;
  y = -s * xi + c * yi;
}

// JacobiRotation::makeJacobi(x, y, z)  Jacobi.h:92-125
PCM_LA bool make_jacobi(double x, double y, double z, double& c, double& s) {
  const double deno = 2.0 * fabs(y);
  if (deno < DBL_MIN) { c = 1.0; s = 0.0; return false; }
  const double tau = (x - z) / deno;
  const double w = sqrt(tau * tau + 1.0);
  const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0.0 ? 1.0 : -1.0;
  const double n = 1.0 / sqrt(t * t + 1.0);
  s = -sign_t * (y / fabs(y)) * fabs(t) * n;
  c = n;
  return true;
}

// JacobiRotation::makeGivens(p, q), real scalars  Jacobi.h:228-262
PCM_LA void make_givens(double p, double q, double& c, double& s) {
  if (q == 0.0) { c = p < 0.0 ? -1.0 : 1.0; s = 0.0; }
  else if (p == 0.0) { c = 0.0; s = q < 0.0 ? 1.0 : -1.0; }
  else if (fabs(p) > fabs(q)) {
    const double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    const double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}

// Two-sided Jacobi SVD of a square matrix: A = U diag(S) V^T, S descending.  U / V may be skipped (WANT_U / WANT_V).
template <int N, bool WANT_U = true, bool WANT_V = true>
PCM_LA void jacobi_svd(const double* A, double* Uo, double* So, double* Vo) {
  double w[N][N], U[N][N], V[N][N], sv[N];
  const double precision = 2.0 * DBL_EPSILON, considerAsZero = DBL_MIN;
  double scale = 0.0;
  bool finite = true;
  for (int i = 0; i < N * N; i++) {
    const double a = fabs(A[i]);
    if (!(a <= DBL_MAX)) finite = false;
    if (a > scale) scale = a;
  }
  if (!finite) {   // JacobiSVD.h:681-685 InvalidInput
    for (int i = 0; i < N; i++) {
      So[i] = 0.0;
      for (int j = 0; j < N; j++) { if (WANT_U) Uo[i * N + j] = i == j; if (WANT_V) Vo[i * N + j] = i == j; }
    }
    return;
  }
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { w[i][j] = A[i * N + j] / scale; U[i][j] = V[i][j] = (i == j) ? 1.0 : 0.0; }
  double maxDiag = 0.0;
  for (int i = 0; i < N; i++) if (fabs(w[i][i]) > maxDiag) maxDiag = fabs(w[i][i]);
  bool finished = false;
  while (!finished) {
    finished = true;
    // (all trip counts are compile-time: unrolled, the three matrices stay in registers in device code)
PCM_UNROLL
    for (int p = 1; p < N; p++) {
PCM_UNROLL
      for (int q = 0; q < p; q++) {
        double thr = precision * maxDiag;
        if (considerAsZero > thr) thr = considerAsZero;
        if (fabs(w[p][q]) > thr || fabs(w[q][p]) > thr) {
          finished = false;
          // real_2x2_jacobi_svd  RealSvd2x2.h:19-51
          double m00 = w[p][p], m01 = w[p][q], m10 = w[q][p], m11 = w[q][q];
          double r1c, r1s;
          const double t = m00 + m11, d = m10 - m01;
          if (fabs(d) < DBL_MIN) { r1s = 0.0; r1c = 1.0; }
          else { const double u = t / d; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
          plane_rot(m00, m10, r1c, r1s);
          plane_rot(m01, m11, r1c, r1s);
          double jrc, jrs;
          make_jacobi(m00, m01, m11, jrc, jrs);
          const double tc = jrc, ts = -jrs;                       // j_right.transpose()
          const double jlc = r1c * tc - r1s * ts, jls = r1c * ts + r1s * tc;   // rot1 * j_right^T  (Jacobi.h:49-55)
PCM_UNROLL
          for (int j = 0; j < N; j++) plane_rot(w[p][j], w[q][j], jlc, jls);
          if (WANT_U) {
PCM_UNROLL
            for (int i = 0; i < N; i++) plane_rot(U[i][p], U[i][q], jlc, jls);
          }
PCM_UNROLL
          for (int i = 0; i < N; i++) plane_rot(w[i][p], w[i][q], jrc, -jrs);
          if (WANT_V) {
PCM_UNROLL
            for (int i = 0; i < N; i++) plane_rot(V[i][p], V[i][q], jrc, -jrs);
          }
          const double mx = fabs(w[p][p]) > fabs(w[q][q]) ? fabs(w[p][p]) : fabs(w[q][q]);
          if (mx > maxDiag) maxDiag = mx;
        }
      }
    }
  }
  for (int i = 0; i < N; i++) {
    const double a = w[i][i];
    sv[i] = fabs(a);
    if (WANT_U && a < 0.0) for (int r = 0; r < N; r++) U[r][i] = -U[r][i];
  }
  for (int i = 0; i < N; i++) sv[i] *= scale;
  for (int i = 0; i < N; i++) {   // selection sort, descending, first of equal maxima
    int pos = i;
    double mxv = sv[i];
    for (int j = i + 1; j < N; j++) if (sv[j] > mxv) { mxv = sv[j]; pos = j; }
    if (mxv == 0.0) break;
    if (pos != i) {
      const double t = sv[i]; sv[i] = sv[pos]; sv[pos] = t;
      for (int r = 0; r < N; r++) {
        if (WANT_U) { const double u = U[r][pos]; U[r][pos] = U[r][i]; U[r][i] = u; }
        if (WANT_V) { const double u = V[r][pos]; V[r][pos] = V[r][i]; V[r][i] = u; }
      }
    }
  }
  for (int i = 0; i < N; i++) {
    So[i] = sv[i];
    for (int j = 0; j < N; j++) { if (WANT_U) Uo[i * N + j] = U[i][j]; if (WANT_V) Vo[i * N + j] = V[i][j]; }
  }
}

// JacobiSVD<Matrix6d>(H, ComputeFullU | ComputeFullV).solve(b)
PCM_LA void svd_solve6(const double* H, const double* b, double* x) {
  double U[36], V[36], S[6], tmp[6];
  jacobi_svd<6>(H, U, S, V);
  double pre = S[0] * (6.0 * DBL_EPSILON);
  if (DBL_MIN > pre) pre = DBL_MIN;
  int nz = 6;
  for (int i = 0; i < 6; i++) if (S[i] == 0.0) { nz = i; break; }
  int rank = nz;
  while (rank > 0 && S[rank - 1] < pre) rank--;
  for (int j = 0; j < rank; j++) {   // U.col(j) . b: fixed size 6, contiguous -> three packets of two, tree, horizontal add (CORE-1)
    const double p0 = U[0 * 6 + j] * b[0], p1 = U[1 * 6 + j] * b[1], p2 = U[2 * 6 + j] * b[2], p3 = U[3 * 6 + j] * b[3], p4 = U[4 * 6 + j] * b[4],
                 p5 = U[5 * 6 + j] * b[5];
    tmp[j] = (p0 + (p2 + p4)) + (p1 + (p3 + p5));
  }
  for (int j = 0; j < rank; j++) tmp[j] = (1.0 / S[j]) * tmp[j];
  for (int i = 0; i < 6; i++) {      // V.row(i) . tmp: strided, runtime length -> left to right
    double s = 0.0;
    if (rank > 0) { s = V[i * 6 + 0] * tmp[0]; for (int j = 1; j < rank; j++) s = s + V[i * 6 + j] * tmp[j]; }
    x[i] = s;
  }
}

// SelfAdjointEigenSolver<Matrix3d>::compute: eigenvalues ascending in w, eigenvectors in the COLUMNS of V.  Reads the lower triangle.
PCM_LA bool selfadjoint3(const double* A, double* w, double* V) {
  double mat[3][3], diag[3], sub[2], Q[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mat[i][j] = (j <= i) ? A[i * 3 + j] : 0.0;
  double scale = 0.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (fabs(mat[i][j]) > scale) scale = fabs(mat[i][j]);
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++) mat[i][j] /= scale;
  diag[0] = mat[0][0];
  const double v1norm2 = mat[2][0] * mat[2][0];
  if (v1norm2 <= DBL_MIN) {   // Tridiagonalization.h:476-484
    diag[1] = mat[1][1]; diag[2] = mat[2][2]; sub[0] = mat[1][0]; sub[1] = mat[2][1];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] = i == j;
  } else {                    // :486-501
    const double beta = sqrt(mat[1][0] * mat[1][0] + v1norm2);
    const double invBeta = 1.0 / beta;
    const double m01 = mat[1][0] * invBeta, m02 = mat[2][0] * invBeta;
    const double q = 2.0 * m01 * mat[2][1] + m02 * (mat[2][2] - mat[1][1]);
    diag[1] = mat[1][1] + m02 * q;
    diag[2] = mat[2][2] - m02 * q;
    sub[0] = beta;
    sub[1] = mat[2][1] - m01 * q;
    Q[0][0] = 1; Q[0][1] = 0; Q[0][2] = 0;
    Q[1][0] = 0; Q[1][1] = m01; Q[1][2] = m02;
    Q[2][0] = 0; Q[2][1] = m02; Q[2][2] = -m01;
  }
  const int n = 3, maxIterations = 30;
  int end = n - 1, start = 0, iter = 0;
  const double precision_inv = 1.0 / DBL_EPSILON;
  while (end > 0) {
    for (int i = start; i < end; i++) {
      if (fabs(sub[i]) < DBL_MIN) sub[i] = 0.0;
      else {
        const double scaled = precision_inv * sub[i];
        if (scaled * scaled <= (fabs(diag[i]) + fabs(diag[i + 1]))) sub[i] = 0.0;
      }
    }
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > maxIterations * n) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    // tridiagonal_qr_step  SelfAdjointEigenSolver.h:838-895
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) mu -= fabs(e);
    else if (e != 0.0) {
      const double e2 = e * e;
      const double h = hypot(td, e);
      if (e2 == 0.0) mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
      else mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
    double xx = diag[start] - mu;
    double z = sub[start];
    for (int k = start; k < end && z != 0.0; k++) {
      double c, s;
      make_givens(xx, z, c, s);
      const double sdk = s * diag[k] + c * sub[k];
      const double dkp1 = s * sub[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
      diag[k + 1] = s * sdk + c * dkp1;
      sub[k] = c * sdk - s * dkp1;
      if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
      xx = sub[k];
      if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
      for (int r = 0; r < 3; r++) plane_rot(Q[r][k], Q[r][k + 1], c, -s);
    }
  }
  const bool ok = iter <= maxIterations * n;
  if (ok) {
    for (int i = 0; i < n - 1; i++) {
      int k = 0;
      double mn = diag[i];
      for (int j = 1; j < n - i; j++) if (diag[i + j] < mn) { mn = diag[i + j]; k = j; }
      if (k > 0) {
        const double t = diag[i]; diag[i] = diag[k + i]; diag[k + i] = t;
        for (int r = 0; r < 3; r++) { const double u = Q[r][i]; Q[r][i] = Q[r][k + i]; Q[r][k + i] = u; }
      }
    }
  }
  for (int i = 0; i < 3; i++) { w[i] = diag[i] * scale; for (int j = 0; j < 3; j++) V[i * 3 + j] = Q[i][j]; }
  return ok;
}

// extract_kernel  SelfAdjointEigenSolver.h:634-653
PCM_LA void direct3f_kernel(const float (&t)[3][3], float (&res)[3], float (&rep)[3]) {
  int i0 = 0;
  float best = fabsf(t[0][0]);
  for (int i = 1; i < 3; i++) if (fabsf(t[i][i]) > best) { best = fabsf(t[i][i]); i0 = i; }
  for (int r = 0; r < 3; r++) rep[r] = t[r][i0];
  const int i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
  float c0[3], c1[3];
  c0[0] = rep[1] * t[2][i1] - rep[2] * t[1][i1]; c0[1] = rep[2] * t[0][i1] - rep[0] * t[2][i1]; c0[2] = rep[0] * t[1][i1] - rep[1] * t[0][i1];
  c1[0] = rep[1] * t[2][i2] - rep[2] * t[1][i2]; c1[1] = rep[2] * t[0][i2] - rep[0] * t[2][i2]; c1[2] = rep[0] * t[1][i2] - rep[1] * t[0][i2];
  const float n0 = (c0[0] * c0[0] + c0[1] * c0[1]) + c0[2] * c0[2];
  const float n1 = (c1[0] * c1[0] + c1[1] * c1[1]) + c1[2] * c1[2];
  if (n0 > n1) { const float s = sqrtf(n0); for (int r = 0; r < 3; r++) res[r] = la_divf(c0[r], s); }
  else { const float s = sqrtf(n1); for (int r = 0; r < 3; r++) res[r] = la_divf(c1[r], s); }
}

// SelfAdjointEigenSolver<Matrix3f>::computeDirect (closed form): eigenvalues ascending, eigenvectors in the COLUMNS of V.
PCM_LA void selfadjoint3_direct(const float* A, float* w, float* V) {
  float sm[3][3], ev[3], vec[3][3];
  const float shift = la_divf((A[0] + A[4]) + A[8], 3.0f);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sm[i][j] = A[(i > j ? i : j) * 3 + (i > j ? j : i)];
  for (int i = 0; i < 3; i++) sm[i][i] -= shift;
  float scale = 0.0f;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (fabsf(sm[i][j]) > scale) scale = fabsf(sm[i][j]);
  if (scale > 0.0f) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sm[i][j] = la_divf(sm[i][j], scale);
  {   // computeRoots :589-631
    const float s_inv3 = la_divf(1.0f, 3.0f), s_sqrt3 = sqrtf(3.0f);
    const float c0 = sm[0][0] * sm[1][1] * sm[2][2] + 2.0f * sm[1][0] * sm[2][0] * sm[2][1] - sm[0][0] * sm[2][1] * sm[2][1] - sm[1][1] * sm[2][0] * sm[2][0] -
                     sm[2][2] * sm[1][0] * sm[1][0];
    const float c1 = sm[0][0] * sm[1][1] - sm[1][0] * sm[1][0] + sm[0][0] * sm[2][2] - sm[2][0] * sm[2][0] + sm[1][1] * sm[2][2] - sm[2][1] * sm[2][1];
    const float c2 = sm[0][0] + sm[1][1] + sm[2][2];
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
    a_over_3 = a_over_3 > 0.0f ? a_over_3 : 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    q = q > 0.0f ? q : 0.0f;
    const float rho = sqrtf(a_over_3);
    // atan2 / cos / sin are evaluated in double and rounded to float: a correctly rounded stand-in for the device libm of
    // the reference's CUDA build, reproducible between host and device
    const float theta = (float)atan2((double)sqrtf(q), (double)half_b) * s_inv3;
    const float cos_theta = (float)cos((double)theta), sin_theta = (float)sin((double)theta);
    ev[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    ev[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    ev[2] = c2_over_3 + 2.0f * rho * cos_theta;
  }
  if ((ev[2] - ev[0]) <= FLT_EPSILON) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) vec[i][j] = i == j;
  } else {
    float tmp[3][3], colk[3], coll[3];
    float d0 = ev[2] - ev[1], d1 = ev[1] - ev[0];
    int k = 0, l = 2;
    if (d0 > d1) { k = 2; l = 0; d0 = d1; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[i][j] = sm[i][j];
    for (int i = 0; i < 3; i++) tmp[i][i] -= ev[k];
    direct3f_kernel(tmp, colk, coll);
    if (d0 <= 2.0f * FLT_EPSILON * d1) {
      const float dt = (colk[0] * coll[0] + colk[1] * coll[1]) + colk[2] * coll[2];
      for (int r = 0; r < 3; r++) coll[r] -= dt * coll[r];
      const float nn = sqrtf((coll[0] * coll[0] + coll[1] * coll[1]) + coll[2] * coll[2]);
      for (int r = 0; r < 3; r++) coll[r] = la_divf(coll[r], nn);
    } else {
      float dummy[3];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[i][j] = sm[i][j];
      for (int i = 0; i < 3; i++) tmp[i][i] -= ev[l];
      direct3f_kernel(tmp, coll, dummy);
    }
    for (int r = 0; r < 3; r++) { vec[r][k] = colk[r]; vec[r][l] = coll[r]; }
    float cr[3] = {vec[1][2] * vec[2][0] - vec[2][2] * vec[1][0], vec[2][2] * vec[0][0] - vec[0][2] * vec[2][0], vec[0][2] * vec[1][0] - vec[1][2] * vec[0][0]};
    const float n2 = (cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2];
    if (n2 > 0.0f) { const float nn = sqrtf(n2); for (int r = 0; r < 3; r++) cr[r] = la_divf(cr[r], nn); }
    for (int r = 0; r < 3; r++) vec[r][1] = cr[r];
  }
  for (int i = 0; i < 3; i++) { w[i] = ev[i] * scale + shift; for (int j = 0; j < 3; j++) V[i * 3 + j] = vec[i][j]; }
}

// Matrix3::inverse(): cofactors of column 0 first, det = (c00 m00 + c10 m10) + c20 m20, result(r, c) = cofactor<c, r> / det
template <typename T>
PCM_LA void inv3(const T (&m)[9], T (&inv)[9]) {
  T cof[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      cof[i][j] = m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
    }
  }
  const T det = (cof[0][0] * m[0] + cof[1][0] * m[3]) + cof[2][0] * m[6];
  T invdet;
  if constexpr (sizeof(T) == 4) invdet = la_divf(1.0f, (float)det);
  else invdet = (T)1 / det;
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 3; c++) inv[r * 3 + c] = cof[c][r] * invdet;
  }
}

// Matrix4d::inverse() as every x86-64 build runs it: 2x2 sub-block form on pairs of doubles (InverseSize4.h:166-351).
// Helpers of the file: swizzle2(a, b, mask) = (a[mask & 1], b[mask >> 1]); duplane(a, p) = (a[p], a[p]).
struct P2 { double v[2]; };
PCM_LA P2 p2(double a, double b) { P2 r; r.v[0] = a; r.v[1] = b; return r; }
PCM_LA P2 p2mul(P2 a, P2 b) { return p2(a.v[0] * b.v[0], a.v[1] * b.v[1]); }
PCM_LA P2 p2add(P2 a, P2 b) { return p2(a.v[0] + b.v[0], a.v[1] + b.v[1]); }
PCM_LA P2 p2sub(P2 a, P2 b) { return p2(a.v[0] - b.v[0], a.v[1] - b.v[1]); }
PCM_LA P2 p2swz(P2 a, P2 b, int mask) { return p2(a.v[mask & 1], b.v[(mask >> 1) & 1]); }
PCM_LA P2 p2dup(P2 a, int p) { return p2(a.v[p], a.v[p]); }
PCM_LA void inv4d(const double* M, double* R) {
  double cm[16], res[16];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) cm[j * 4 + i] = M[i * 4 + j];
  const P2 A1 = p2(cm[0], cm[1]), B1 = p2(cm[2], cm[3]), A2 = p2(cm[4], cm[5]), B2 = p2(cm[6], cm[7]);
  const P2 C1 = p2(cm[8], cm[9]), D1 = p2(cm[10], cm[11]), C2 = p2(cm[12], cm[13]), D2 = p2(cm[14], cm[15]);
  P2 dA, dB, dC, dD;
  dA = p2swz(A2, A2, 1); dA = p2mul(A1, dA); dA = p2sub(dA, p2dup(dA, 1));
  dB = p2swz(B2, B2, 1); dB = p2mul(B1, dB); dB = p2sub(dB, p2dup(dB, 1));
  dC = p2swz(C2, C2, 1); dC = p2mul(C1, dC); dC = p2sub(dC, p2dup(dC, 1));
  dD = p2swz(D2, D2, 1); dD = p2mul(D1, dD); dD = p2sub(dD, p2dup(dD, 1));
  P2 AB1 = p2mul(B1, p2dup(A2, 1));
  P2 AB2 = p2mul(B2, p2dup(A1, 0));
  AB1 = p2sub(AB1, p2mul(B2, p2dup(A1, 1)));
  AB2 = p2sub(AB2, p2mul(B1, p2dup(A2, 0)));
  P2 DC1 = p2mul(C1, p2dup(D2, 1));
  P2 DC2 = p2mul(C2, p2dup(D1, 0));
  DC1 = p2sub(DC1, p2mul(C2, p2dup(D1, 1)));
  DC2 = p2sub(DC2, p2mul(C1, p2dup(D2, 0)));
  P2 d1 = p2mul(AB1, p2swz(DC1, DC2, 0));
  P2 d2 = p2mul(AB2, p2swz(DC1, DC2, 3));
  P2 rd = p2add(d1, d2);
  rd = p2add(rd, p2dup(rd, 1));
  d1 = p2mul(dA, dD);
  d2 = p2mul(dB, dC);
  P2 det = p2add(d1, d2);
  det = p2sub(det, rd);
  det = p2dup(det, 0);
  rd = p2(1.0 / det.v[0], 1.0 / det.v[1]);
  P2 iD1 = p2mul(AB1, p2dup(C1, 0));
  P2 iD2 = p2mul(AB1, p2dup(C2, 0));
  iD1 = p2add(iD1, p2mul(AB2, p2dup(C1, 1)));
  iD2 = p2add(iD2, p2mul(AB2, p2dup(C2, 1)));
  dA = p2dup(dA, 0);
  iD1 = p2sub(p2mul(D1, dA), iD1);
  iD2 = p2sub(p2mul(D2, dA), iD2);
  P2 iA1 = p2mul(DC1, p2dup(B1, 0));
  P2 iA2 = p2mul(DC1, p2dup(B2, 0));
  iA1 = p2add(iA1, p2mul(DC2, p2dup(B1, 1)));
  iA2 = p2add(iA2, p2mul(DC2, p2dup(B2, 1)));
  dD = p2dup(dD, 0);
  iA1 = p2sub(p2mul(A1, dD), iA1);
  iA2 = p2sub(p2mul(A2, dD), iA2);
  P2 iB1 = p2mul(D1, p2swz(AB2, AB1, 1));
  P2 iB2 = p2mul(D2, p2swz(AB2, AB1, 1));
  iB1 = p2sub(iB1, p2mul(p2swz(D1, D1, 1), p2swz(AB2, AB1, 2)));
  iB2 = p2sub(iB2, p2mul(p2swz(D2, D2, 1), p2swz(AB2, AB1, 2)));
  dB = p2dup(dB, 0);
  iB1 = p2sub(p2mul(C1, dB), iB1);
  iB2 = p2sub(p2mul(C2, dB), iB2);
  P2 iC1 = p2mul(A1, p2swz(DC2, DC1, 1));
  P2 iC2 = p2mul(A2, p2swz(DC2, DC1, 1));
  iC1 = p2sub(iC1, p2mul(p2swz(A1, A1, 1), p2swz(DC2, DC1, 2)));
  iC2 = p2sub(iC2, p2mul(p2swz(A2, A2, 1), p2swz(DC2, DC1, 2)));
  dC = p2dup(dC, 0);
  iC1 = p2sub(p2mul(B1, dC), iC1);
  iC2 = p2sub(p2mul(B2, dC), iC2);
  d1 = p2(rd.v[0], -rd.v[1]);
  d2 = p2(-rd.v[0], rd.v[1]);
  P2 o;
  o = p2mul(p2swz(iA2, iA1, 3), d1); res[0] = o.v[0]; res[1] = o.v[1];
  o = p2mul(p2swz(iA2, iA1, 0), d2); res[4] = o.v[0]; res[5] = o.v[1];
  o = p2mul(p2swz(iB2, iB1, 3), d1); res[2] = o.v[0]; res[3] = o.v[1];
  o = p2mul(p2swz(iB2, iB1, 0), d2); res[6] = o.v[0]; res[7] = o.v[1];
  o = p2mul(p2swz(iC2, iC1, 3), d1); res[8] = o.v[0]; res[9] = o.v[1];
  o = p2mul(p2swz(iC2, iC1, 0), d2); res[12] = o.v[0]; res[13] = o.v[1];
  o = p2mul(p2swz(iD2, iD1, 3), d1); res[10] = o.v[0]; res[11] = o.v[1];
  o = p2mul(p2swz(iD2, iD1, 0), d2); res[14] = o.v[0]; res[15] = o.v[1];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) R[i * 4 + j] = res[j * 4 + i];
}

}  // namespace pcm

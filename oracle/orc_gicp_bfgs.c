/*
 * oracle/orc_gicp_bfgs.c -- objective / gradient of pclomp's GICP-BFGS (TEST INFRASTRUCTURE ONLY).
 *
 * Restates the functor the BFGS of jueying_slam's GICP_OMP option evaluates
 * (/root/reference/src/pointcloud_match/ndt_omp/include/pclomp/gicp_omp_impl.hpp):
 *   applyState            :519-529  t <- [Rz(x5) Ry(x4) Rx(x3)] t (float, via the quaternion product Eigen forms
 *                                   for AngleAxisf * AngleAxisf), translation added to column 3
 *   operator()  (mode 0)  :246-274  f = 1/m  sum  res . (M res)      res = T p_src - p_tgt, float, summed in double
 *   df          (mode 1)  :278-327  g_t = 2/m sum M res ; R = 2/m sum (base p_src)(M res)^T ; computeRDerivative
 *   fdf         (mode 2)  :331-365  both, f from the double residual
 *   computeRDerivative    :125-176  g[3..5] = <dR/dphi, R>, <dR/dtheta, R>, <dR/dpsi, R>
 * mahalanobis_ is a vector of Matrix4f (column-major) indexed by the SOURCE index; its 4th row/column and the
 * points' w = 1 contribute exact zeros, so only the 3x3 block enters.
 *
 * PARITY PIN STATUS: "parity unpinned" -- the reference holds no fixture for these functions and cannot be built
 * here.  The reference's operator()/df add per-thread partial sums in an OpenMP-schedule-dependent order; this
 * restatement (like fdf) adds in index order.  The order of the float additions inside Eigen's fixed-size products
 * depends on its vectorisation; the left-to-right order is used here.
 */
#include "orc_internal.h"

static void quat_axis(float angle, int axis, float q[4]) {   /* Quaternionf(AngleAxisf): (x, y, z, w) */
  const float ha = 0.5f * angle;
  const float s = sinf(ha);
  q[0] = q[1] = q[2] = 0.f;
  q[axis] = s;
  q[3] = cosf(ha);
}

static void quat_mul(const float a[4], const float b[4], float r[4]) {
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

/* base, T: row-major 4x4 floats */
void orc_gicp_bfgs_apply_state(const float base[16], const double x[6], float T[16]) {
  float qz[4], qy[4], qx[4], qzy[4], q[4];
  quat_axis((float)x[5], 2, qz);
  quat_axis((float)x[4], 1, qy);
  quat_axis((float)x[3], 0, qx);
  quat_mul(qz, qy, qzy);
  quat_mul(qzy, qx, q);
  const float tx = 2.f * q[0], ty = 2.f * q[1], tz = 2.f * q[2];
  const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  const float R[9] = {1.f - (tyy + tzz), txy - twz, txz + twy,
                      txy + twz, 1.f - (txx + tzz), tyz - twx,
                      txz - twy, tyz + twx, 1.f - (txx + tyy)};
  for (int i = 0; i < 16; i++) T[i] = base[i];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 4 + j] = (R[i * 3 + 0] * base[0 * 4 + j] + R[i * 3 + 1] * base[1 * 4 + j]) + R[i * 3 + 2] * base[2 * 4 + j];
  for (int i = 0; i < 3; i++) T[i * 4 + 3] = base[i * 4 + 3] + (float)x[i];
}

static double inner_prod(const double A[9], const double B[9]) {   /* gicp_omp.h:325-334: sum_i sum_j A(j,i) B(i,j) */
  double r = 0.;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r += A[j * 3 + i] * B[i * 3 + j];
  return r;
}

void orc_gicp_bfgs_r_derivative(const double x[6], const double R[9], double g[6]) {
  const double phi = x[3], theta = x[4], psi = x[5];
  const double cphi = cos(phi), sphi = sin(phi), ctheta = cos(theta), stheta = sin(theta), cpsi = cos(psi), spsi = sin(psi);
  const double dphi[9] = {0., sphi * spsi + cphi * cpsi * stheta, cphi * spsi - cpsi * sphi * stheta,
                          0., -cpsi * sphi + cphi * spsi * stheta, -cphi * cpsi - sphi * spsi * stheta,
                          0., cphi * ctheta, -ctheta * sphi};
  const double dtheta[9] = {-cpsi * stheta, cpsi * ctheta * sphi, cphi * cpsi * ctheta,
                            -spsi * stheta, ctheta * sphi * spsi, cphi * ctheta * spsi,
                            -ctheta, -sphi * stheta, -cphi * stheta};
  const double dpsi[9] = {-ctheta * spsi, -cphi * cpsi - sphi * spsi * stheta, cpsi * sphi - cphi * spsi * stheta,
                          cpsi * ctheta, -cphi * spsi + cpsi * sphi * stheta, sphi * spsi + cphi * cpsi * stheta,
                          0., 0., 0.};
  g[3] = inner_prod(dphi, R);
  g[4] = inner_prod(dtheta, R);
  g[5] = inner_prod(dpsi, R);
}

static void xform(const float T[16], const float *p, float r[3]) {   /* rows 0..2 of T * (x, y, z, 1) */
  for (int a = 0; a < 3; a++) r[a] = ((T[a * 4 + 0] * p[0] + T[a * 4 + 1] * p[1]) + T[a * 4 + 2] * p[2]) + T[a * 4 + 3] * 1.f;
}

/* src/tgt: records of stride_f floats (x y z first); maha: 16 floats per SOURCE point, column-major Matrix4f.
 * mode 0: f only (float residual path), 1: g only, 2: f (double residual path) and g.  Returns 0, -1 if m < 1. */
int orc_gicp_bfgs_fdf(const float *src, const float *tgt, long stride_f, const int *idx_src, const int *idx_tgt, long m, const float *maha,
                      const float base[16], const double x[6], int mode, double *f_out, double g[6]) {
  if (m < 1) return -1;
  float T[16];
  orc_gicp_bfgs_apply_state(base, x, T);
  double f32sum = 0., f64sum = 0., gt[3] = {0., 0., 0.}, R[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
  for (long i = 0; i < m; i++) {
    const float *p = src + (size_t)idx_src[i] * stride_f, *q = tgt + (size_t)idx_tgt[i] * stride_f;
    const float *M = maha + (size_t)idx_src[i] * 16;   /* M(a,b) = M[b * 4 + a] */
    float pp[3], res[3];
    xform(T, p, pp);
    for (int a = 0; a < 3; a++) res[a] = pp[a] - q[a];
    {   /* operator(): res . (M res) in float */
      float Mr[3];
      for (int a = 0; a < 3; a++) Mr[a] = (M[0 * 4 + a] * res[0] + M[1 * 4 + a] * res[1]) + M[2 * 4 + a] * res[2];
      const float ret = (res[0] * Mr[0] + res[1] * Mr[1]) + res[2] * Mr[2];
      f32sum += (double)ret;
    }
    double rd[3] = {(double)res[0], (double)res[1], (double)res[2]}, temp[3];
    for (int a = 0; a < 3; a++) temp[a] = ((double)M[0 * 4 + a] * rd[0] + (double)M[1 * 4 + a] * rd[1]) + (double)M[2 * 4 + a] * rd[2];
    f64sum += (rd[0] * temp[0] + rd[1] * temp[1]) + rd[2] * temp[2];
    float pb[3];
    xform(base, p, pb);
    for (int a = 0; a < 3; a++) {
      gt[a] += temp[a];
      for (int b = 0; b < 3; b++) R[a * 3 + b] += (double)pb[a] * temp[b];
    }
  }
  if (f_out) *f_out = (mode == 0 ? f32sum : f64sum) / (double)m;
  if (g && mode != 0) {
    for (int a = 0; a < 6; a++) g[a] = 0.;
    for (int a = 0; a < 3; a++) g[a] = gt[a] * (2.0 / (double)m);
    for (int a = 0; a < 9; a++) R[a] *= 2.0 / (double)m;
    orc_gicp_bfgs_r_derivative(x, R, g);
  }
  return 0;
}

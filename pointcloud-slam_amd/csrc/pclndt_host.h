// pclndt_host.h -- the serial control flow of the pclomp NDT operator, for the host AND for the device's step kernel:
// pclomp::NormalDistributionsTransform::computeTransformation / computeStepLengthMT
// (ndt_omp/include/pclomp/ndt_omp_impl.hpp:69-156, 593-833) around a device "evaluate" callback.
// Eigen pieces the reference calls and their restatement here:
//   Matrix3f::eulerAngles(0,1,2), AngleAxis<float>::toRotationMatrix, Translation * AngleAxis products (float),
//   JacobiSVD<Matrix6d>::solve (two-sided Jacobi with real_2x2_jacobi_svd, Eigen/src/SVD/JacobiSVD.h; rank threshold
//   diagSize * epsilon * sigma_max).
#pragma once
#include "dev_linalg.h"

#include <cfloat>
#include <cmath>
#include <cstring>

#include "pcm_device.h"

namespace pcm {
namespace ndtomp {

// sinf / cosf of the pose composition: libm on the host (what the reference's Eigen::AngleAxis<float> calls); in device code the
// double routine rounded to float, which is the correctly rounded float result but for double-rounding ties (~2^-29 of the inputs)
PCM_LA float sin_f(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return (float)sin((double)x);
#else
  return std::sin(x);
#endif
}
PCM_LA float cos_f(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return (float)cos((double)x);
#else
  return std::cos(x);
#endif
}

struct Eval {   // result of one derivatives pass
  double score;
  double g[6];
  double H[36];
};

// Eigen::AngleAxis<float>::toRotationMatrix, unit coordinate axis
PCM_LA void angle_axis(float angle, int axis, float* R) {
  float ax[3] = {0.f, 0.f, 0.f};
  ax[axis] = 1.f;
  const float sn = sin_f(angle), c = cos_f(angle);
  const float sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]}, c1[3] = {(1.f - c) * ax[0], (1.f - c) * ax[1], (1.f - c) * ax[2]};
  float t;
  t = c1[0] * ax[1]; R[1] = t - sa[2]; R[3] = t + sa[2];
  t = c1[0] * ax[2]; R[2] = t + sa[1]; R[6] = t - sa[1];
  t = c1[1] * ax[2]; R[5] = t - sa[0]; R[7] = t + sa[0];
  for (int a = 0; a < 3; a++) R[a * 4] = c1[a] * ax[a] + c;
}

// final_transformation_ = Translation(p[0..2]) * AngleAxis(p3, X) * AngleAxis(p4, Y) * AngleAxis(p5, Z), float  :723-724
PCM_LA void set_pose(NdtOmpParams& P, const double* p) {
  float Rx[9], Ry[9], Rz[9], A[9];
  angle_axis((float)p[3], 0, Rx);
  angle_axis((float)p[4], 1, Ry);
  angle_axis((float)p[5], 2, Rz);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i * 3 + j] = (Rx[i * 3 + 0] * Ry[0 * 3 + j] + Rx[i * 3 + 1] * Ry[1 * 3 + j]) + Rx[i * 3 + 2] * Ry[2 * 3 + j];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) P.T[i * 4 + j] = (A[i * 3 + 0] * Rz[0 * 3 + j] + A[i * 3 + 1] * Rz[1 * 3 + j]) + A[i * 3 + 2] * Rz[2 * 3 + j];
    P.T[i * 4 + 3] = (float)p[i];
  }
  P.T[12] = P.T[13] = P.T[14] = 0.f; P.T[15] = 1.f;
}

// computeAngleDerivatives  :270-366
PCM_LA void angle_derivatives(NdtOmpParams& P, const double* p) {
  double cx, cy, cz, sx, sy, sz;
  if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
  if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
  if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
  const double J[8][3] = {{(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)}, {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
                          {(-sy * cz), sy * sz, cy}, {sx * cy * cz, (-sx * cy * sz), sx * sy}, {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
                          {(-cy * sz), (-cy * cz), 0}, {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0}, {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
  const double Hd[15][3] = {{(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy}, {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},
                            {(cx * cy * cz), (-cx * cy * sz), (cx * sy)}, {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},
                            {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0}, {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},
                            {(-cy * cz), (cy * sz), (-sy)}, {(-sx * sy * cz), (sx * sy * sz), (sx * cy)}, {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},
                            {(sy * sz), (sy * cz), 0}, {(-sx * cy * sz), (-sx * cy * cz), 0}, {(cx * cy * sz), (cx * cy * cz), 0},
                            {(-cy * cz), (cy * sz), 0}, {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0}, {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};
  for (int r = 0; r < 8; r++) { for (int c = 0; c < 3; c++) { P.j_ang_d[r][c] = J[r][c]; P.j_ang[r][c] = (float)J[r][c]; } P.j_ang[r][3] = 0.f; }
  for (int r = 0; r < 15; r++) { for (int c = 0; c < 3; c++) { P.h_ang_d[r][c] = Hd[r][c]; P.h_ang[r][c] = (float)Hd[r][c]; } P.h_ang[r][3] = 0.f; }
  P.h_ang[6][2] = (float)(sy);   // the float matrix carries (sy) where the double vector h_ang_d1_ carries (-sy)  :351 vs :327
  for (int c = 0; c < 4; c++) P.h_ang[15][c] = 0.f;
}

PCM_LA bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {   // :593-625
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

PCM_LA double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {   // :628-690
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  }
  if (fabs(g_t) <= fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? fmin(a_t + 0.66 * (a_u - a_t), a_n) : fmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

PCM_LA double dot6(const double* a, const double* b) { return ((((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]) + a[4] * b[4]) + a[5] * b[5]; }

// Matrix3f::eulerAngles(0, 1, 2)  (host: the start pose only)
inline void euler_012(const float (&R)[9], float (&res)[3]) {
  res[0] = std::atan2(R[1 * 3 + 2], R[2 * 3 + 2]);
  const float c2 = std::sqrt(R[0] * R[0] + R[1] * R[1]);
  if (res[0] > 0.f) {
    if (res[0] > 0.f) res[0] -= (float)M_PI; else res[0] += (float)M_PI;
    res[1] = std::atan2(-R[2], -c2);
  } else {
    res[1] = std::atan2(-R[2], c2);
  }
  const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * R[2 * 3 + 0] - c1 * R[1 * 3 + 0], c1 * R[1 * 3 + 1] - s1 * R[2 * 3 + 1]);
  for (int a = 0; a < 3; a++) res[a] = -res[a];
}

// eq. 6.8  ndt_omp_impl.hpp:77-82 (host: depends on the configuration only)
inline void gauss_params(NdtOmpParams& P, double outlier_ratio, float resolution, int num_neighbors, double* gauss_d3) {
  const double c1 = 10 * (1 - outlier_ratio), c2 = outlier_ratio / std::pow((double)resolution, 3);
  const double d3 = -std::log(c2);
  if (gauss_d3) *gauss_d3 = d3;
  P.gauss_d1 = -std::log(c1 + c2) - d3;
  P.gauss_d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / P.gauss_d1);
  P.num_neighbors = num_neighbors;
}

// ---------------------------------------------------------------------------------------------------------------------
// computeTransformation (:69-156) + computeStepLengthMT (:693-833) as a machine that is advanced once per evaluation: it holds
// the pose and angle tables of the evaluation it is waiting for (P, request) and, given that evaluation's sums, moves to the
// next one.  The host drives it through a callback (pcm_ndt_derivatives, tests); the device drives one machine per object from
// the step kernel that follows every batched derivatives launch (pclndt.hip), with no host decision in between.
// request: 0 = score + gradient + Hessian (float path), 1 = score + gradient, 2 = Hessian only (double path), -1 = finished
// ---------------------------------------------------------------------------------------------------------------------
struct NdtMachine {
  NdtOmpParams P;
  double step_size, eps;
  double p[6], dir[6], x_t[6];
  Eval cur;
  double phi_0, d_phi_0, a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t, psi_t, d_psi_t, step_min, step_max;
  int32_t open_interval, interval_converged, step_iterations;
  int32_t max_iterations, nr, converged, phase, request;
  int32_t n_deriv, n_hess;
};
enum { NDT_PH_FIRST = 0, NDT_PH_LS_FIRST = 1, NDT_PH_LS_ITER = 2, NDT_PH_LS_HESS = 3, NDT_PH_DONE = 4,
       NDT_PH_NEWTON = 5 /* transient: the next Newton direction is due (no evaluation pending) */ };

PCM_LA void ndt_request(NdtMachine& m, int pass, int phase) {
  m.request = pass;
  m.phase = phase;
  if (pass == 2) m.n_hess++; else m.n_deriv++;
}
PCM_LA void ndt_finish(NdtMachine& m, bool conv) { m.converged = conv ? 1 : 0; m.phase = NDT_PH_DONE; m.request = -1; }

// the line search returned step a_t along dir: p += delta * a; convergence test  :133-150
PCM_LA void ndt_iteration_end(NdtMachine& m, double a) {
  for (int i = 0; i < 6; i++) m.p[i] += m.dir[i] * a;
  const bool conv = m.nr > m.max_iterations || (m.nr && (fabs(a) < m.eps));
  m.nr++;
  if (conv) ndt_finish(m, true); else m.phase = NDT_PH_NEWTON;
}

// after an evaluation inside computeStepLengthMT: the loop test of :776, then either the next trial or the way out (:826-832)
PCM_LA void ndt_line_search_continue(NdtMachine& m) {
  const double mu = 1.e-4, nu = 0.9;
  const int max_step_iterations = 10;
  if (!m.interval_converged && m.step_iterations < max_step_iterations && !(m.psi_t <= 0 && m.d_phi_t <= -nu * m.d_phi_0)) {
    m.a_t = m.open_interval ? trial_value(m.a_l, m.f_l, m.g_l, m.a_u, m.f_u, m.g_u, m.a_t, m.psi_t, m.d_psi_t)
                            : trial_value(m.a_l, m.f_l, m.g_l, m.a_u, m.f_u, m.g_u, m.a_t, m.phi_t, m.d_phi_t);
    m.a_t = fmax(fmin(m.a_t, m.step_max), m.step_min);
    for (int i = 0; i < 6; i++) m.x_t[i] = m.p[i] + m.dir[i] * m.a_t;
    set_pose(m.P, m.x_t);
    angle_derivatives(m.P, m.x_t);
    ndt_request(m, 1, NDT_PH_LS_ITER);
    (void)mu;
    return;
  }
  if (m.step_iterations) { ndt_request(m, 2, NDT_PH_LS_HESS); return; }   // computeHessian at the accepted pose, the angle tables of the last pass
  ndt_iteration_end(m, m.a_t);
}

// Newton direction from the current gradient / Hessian, then the first trial of the line search  :108-131, :693-772
// delta_pre: JacobiSVD(cur.H).solve(-cur.g) when the caller has it already (the device's step kernel computes it with the whole
// workgroup, pclndt.hip svd_solve6_block -- the same arithmetic)
PCM_LA void ndt_newton_begin(NdtMachine& m, const double* delta_pre = nullptr) {
  double mg[6], delta[6];
  if (delta_pre) { for (int i = 0; i < 6; i++) delta[i] = delta_pre[i]; }
  else {
    for (int i = 0; i < 6; i++) mg[i] = -m.cur.g[i];
    pcm::svd_solve6(m.cur.H, mg, delta);
  }
  double nrm = 0;
  for (int i = 0; i < 6; i++) nrm += delta[i] * delta[i];
  nrm = sqrt(nrm);
  if (nrm == 0 || nrm != nrm) { ndt_finish(m, nrm == nrm); return; }   // :117-121
  for (int i = 0; i < 6; i++) m.dir[i] = delta[i] / nrm;
  // computeStepLengthMT(x = p, step_dir = dir, step_init = nrm, step_max = step_size, step_min = eps / 2)
  m.step_max = m.step_size; m.step_min = m.eps / 2;
  m.phi_0 = -m.cur.score;
  m.d_phi_0 = -dot6(m.cur.g, m.dir);
  if (m.d_phi_0 >= 0) {
    if (m.d_phi_0 == 0) { ndt_iteration_end(m, 0.0); return; }
    m.d_phi_0 *= -1;
    for (int i = 0; i < 6; i++) m.dir[i] *= -1;
  }
  const double mu = 1.e-4;
  m.step_iterations = 0;
  m.a_l = 0; m.a_u = 0;
  m.f_l = m.phi_0 - m.phi_0 - mu * m.d_phi_0 * m.a_l; m.g_l = m.d_phi_0 - mu * m.d_phi_0;
  m.f_u = m.phi_0 - m.phi_0 - mu * m.d_phi_0 * m.a_u; m.g_u = m.d_phi_0 - mu * m.d_phi_0;
  m.interval_converged = (m.step_max - m.step_min) < 0 ? 1 : 0;
  m.open_interval = 1;
  m.a_t = fmax(fmin(nrm, m.step_max), m.step_min);
  for (int i = 0; i < 6; i++) m.x_t[i] = m.p[i] + m.dir[i] * m.a_t;
  set_pose(m.P, m.x_t);
  angle_derivatives(m.P, m.x_t);
  ndt_request(m, 0, NDT_PH_LS_FIRST);
}

// computeTransformation up to its first computeDerivatives  :84-104 (host: eulerAngles, log / exp of the configuration)
inline void ndt_machine_start(NdtMachine& m, const float* guess, double step_size, double eps, double outlier_ratio, float resolution, int max_iterations, int num_neighbors) {
  std::memset(&m, 0, sizeof(m));
  m.step_size = step_size; m.eps = eps; m.max_iterations = max_iterations;
  gauss_params(m.P, outlier_ratio, resolution, num_neighbors, nullptr);
  std::memcpy(m.P.T, guess, sizeof(float) * 16);
  float R[9], eul[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = m.P.T[i * 4 + j];
  euler_012(R, eul);
  const double p0[6] = {m.P.T[3], m.P.T[7], m.P.T[11], eul[0], eul[1], eul[2]};
  for (int i = 0; i < 6; i++) m.p[i] = p0[i];
  angle_derivatives(m.P, m.p);
  ndt_request(m, 0, NDT_PH_FIRST);
}

// the sums of the requested pass have arrived: row = 36 x H, 6 x g, score (pass 2: H only)
PCM_LA void ndt_machine_advance(NdtMachine& m, const double* row, const double* delta_pre = nullptr) {
  const double mu = 1.e-4;
  switch (m.phase) {
    case NDT_PH_FIRST:
      for (int i = 0; i < 36; i++) m.cur.H[i] = row[i];
      for (int i = 0; i < 6; i++) m.cur.g[i] = row[36 + i];
      m.cur.score = row[42];
      m.phase = NDT_PH_NEWTON;
      break;
    case NDT_PH_LS_FIRST:
    case NDT_PH_LS_ITER: {
      const bool first = m.phase == NDT_PH_LS_FIRST;
      for (int i = 0; i < 36; i++) m.cur.H[i] = row[i];
      for (int i = 0; i < 6; i++) m.cur.g[i] = row[36 + i];
      m.cur.score = row[42];
      m.phi_t = -m.cur.score; m.d_phi_t = -dot6(m.cur.g, m.dir);
      m.psi_t = m.phi_t - m.phi_0 - mu * m.d_phi_0 * m.a_t; m.d_psi_t = m.d_phi_t - mu * m.d_phi_0;
      if (!first) {
        if (m.open_interval && (m.psi_t <= 0 && m.d_psi_t >= 0)) {
          m.open_interval = 0;
          m.f_l = m.f_l + m.phi_0 - mu * m.d_phi_0 * m.a_l; m.g_l = m.g_l + mu * m.d_phi_0;
          m.f_u = m.f_u + m.phi_0 - mu * m.d_phi_0 * m.a_u; m.g_u = m.g_u + mu * m.d_phi_0;
        }
        m.interval_converged = (m.open_interval ? update_interval(m.a_l, m.f_l, m.g_l, m.a_u, m.f_u, m.g_u, m.a_t, m.psi_t, m.d_psi_t)
                                                : update_interval(m.a_l, m.f_l, m.g_l, m.a_u, m.f_u, m.g_u, m.a_t, m.phi_t, m.d_phi_t)) ? 1 : 0;
        m.step_iterations++;
      }
      ndt_line_search_continue(m);
      break;
    }
    case NDT_PH_LS_HESS:
      for (int i = 0; i < 36; i++) m.cur.H[i] = row[i];
      ndt_iteration_end(m, m.a_t);
      break;
    default:
      break;
  }
  while (m.phase == NDT_PH_NEWTON) ndt_newton_begin(m, delta_pre);   // (a zero directional derivative ends an iteration without an evaluation)
}

// (int pass, const NdtOmpParams&, Eval*) -> status; pass 0: score+g+H (float path), 1: score+g, 2: H only (double path)
template <class F>
struct Solver {
  F eval;
  double step_size, eps, outlier_ratio;
  float resolution;
  int max_iterations, num_neighbors;
  NdtOmpParams P{};
  double gauss_d3 = 0.0;
  int n_deriv = 0, n_hess = 0;

  void gauss_params() { ndtomp::gauss_params(P, outlier_ratio, resolution, num_neighbors, &gauss_d3); }
  void set_pose(const double (&p)[6]) { ndtomp::set_pose(P, p); }
  void angle_derivatives(const double (&p)[6]) { ndtomp::angle_derivatives(P, p); }

  int derivatives(const double (&p)[6], bool hessian, Eval* e) {   // computeDerivatives at the pose in P.T  :168-267
    angle_derivatives(p);
    n_deriv++;
    return eval(hessian ? 0 : 1, P, e);
  }

  // computeTransformation  :69-156: the machine above, one evaluation per turn
  int align(const float (&guess)[16], Eval* last, int* iterations, int* converged) {
    NdtMachine m;
    ndt_machine_start(m, guess, step_size, eps, outlier_ratio, resolution, max_iterations, num_neighbors);
    while (m.request >= 0) {
      Eval e{};
      const int rc = eval(m.request, m.P, &e);
      if (rc != PCM_OK) return rc;
      double row[48];
      std::memcpy(row, e.H, sizeof(e.H));
      std::memcpy(row + 36, e.g, sizeof(e.g));
      row[42] = e.score;
      ndt_machine_advance(m, row);
    }
    P = m.P;
    n_deriv = m.n_deriv; n_hess = m.n_hess;
    *last = m.cur;
    *iterations = m.nr;
    *converged = m.converged;
    return PCM_OK;
  }
};

}  // namespace ndtomp
}  // namespace pcm

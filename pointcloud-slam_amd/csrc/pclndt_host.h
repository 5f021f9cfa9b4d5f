// pclndt_host.h -- host side of the pclomp NDT operator: the serial control flow of
// pclomp::NormalDistributionsTransform::computeTransformation / computeStepLengthMT
// (ndt_omp/include/pclomp/ndt_omp_impl.hpp:69-156, 593-833) around a device "evaluate" callback.
// Eigen pieces the reference calls and their restatement here:
//   Matrix3f::eulerAngles(0,1,2), AngleAxis<float>::toRotationMatrix, Translation * AngleAxis products (float),
//   JacobiSVD<Matrix6d>::solve (one-sided Jacobi; rank threshold diagSize * epsilon * sigma_max).
#pragma once
#include "dev_linalg.h"

#include <cfloat>
#include <cmath>
#include <cstring>

#include "pcm_device.h"

namespace pcm {
namespace ndtomp {

struct Eval {   // result of one derivatives pass
  double score;
  double g[6];
  double H[36];
};

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

  void gauss_params() {   // eq. 6.8  :77-82
    const double c1 = 10 * (1 - outlier_ratio), c2 = outlier_ratio / std::pow((double)resolution, 3);
    const double d3 = -std::log(c2);
    gauss_d3 = d3;
    P.gauss_d1 = -std::log(c1 + c2) - d3;
    P.gauss_d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / P.gauss_d1);
    P.num_neighbors = num_neighbors;
  }

  static void angle_axis(float angle, int axis, float (&R)[9]) {   // Eigen::AngleAxis<float>::toRotationMatrix, unit coordinate axis
    float ax[3] = {0.f, 0.f, 0.f};
    ax[axis] = 1.f;
    const float sn = std::sin(angle), c = std::cos(angle);
    const float sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]}, c1[3] = {(1.f - c) * ax[0], (1.f - c) * ax[1], (1.f - c) * ax[2]};
    float t;
    t = c1[0] * ax[1]; R[1] = t - sa[2]; R[3] = t + sa[2];
    t = c1[0] * ax[2]; R[2] = t + sa[1]; R[6] = t - sa[1];
    t = c1[1] * ax[2]; R[5] = t - sa[0]; R[7] = t + sa[0];
    for (int a = 0; a < 3; a++) R[a * 4] = c1[a] * ax[a] + c;
  }

  // final_transformation_ = Translation(p[0..2]) * AngleAxis(p3, X) * AngleAxis(p4, Y) * AngleAxis(p5, Z), float  :723-724
  void set_pose(const double (&p)[6]) {
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

  void angle_derivatives(const double (&p)[6]) {   // computeAngleDerivatives  :270-366
    double cx, cy, cz, sx, sy, sz;
    if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
    if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
    if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
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

  int derivatives(const double (&p)[6], bool hessian, Eval* e) {   // computeDerivatives at the pose in P.T  :168-267
    angle_derivatives(p);
    n_deriv++;
    return eval(hessian ? 0 : 1, P, e);
  }

  static void euler_012(const float (&R)[9], float (&res)[3]) {   // Matrix3f::eulerAngles(0, 1, 2)
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

  // Eigen::JacobiSVD<Matrix6d>(H, ComputeFullU | ComputeFullV).solve(b)  ndt_omp_impl.hpp:112-114 (two-sided Jacobi, dev_linalg.h)
  static void svd_solve6(const double (&Hin)[36], const double (&b)[6], double (&x)[6]) { pcm::svd_solve6(Hin, b, x); }

  static bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {   // :593-625
    if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
    if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    return true;
  }

  static double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {   // :628-690
    if (f_t > f_l) {
      const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
      const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
      const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
      return std::fabs(a_c - a_l) < std::fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
    }
    if (g_t * g_l < 0) {
      const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
      const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
      const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
      return std::fabs(a_c - a_t) >= std::fabs(a_s - a_t) ? a_c : a_s;
    }
    if (std::fabs(g_t) <= std::fabs(g_l)) {
      const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
      const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
      const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
      const double a_n = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
      return a_t > a_l ? std::fmin(a_t + 0.66 * (a_u - a_t), a_n) : std::fmax(a_t + 0.66 * (a_u - a_t), a_n);
    }
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = std::sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }

  static double dot6(const double* a, const double* b) { return ((((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]) + a[4] * b[4]) + a[5] * b[5]; }

  // computeStepLengthMT  :693-833
  int step_length(const double (&x)[6], double (&dir)[6], double step_init, double step_max, double step_min, Eval& cur, double* a_out) {
    const double phi_0 = -cur.score;
    double d_phi_0 = -dot6(cur.g, dir);
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) { *a_out = 0; return PCM_OK; }
      d_phi_0 *= -1;
      for (double& d : dir) d *= -1;
    }
    const int max_step_iterations = 10;
    int step_iterations = 0;
    const double mu = 1.e-4, nu = 0.9;
    double a_l = 0, a_u = 0;
    auto psi = [&](double a, double f_a) { return f_a - phi_0 - mu * d_phi_0 * a; };
    auto dpsi = [&](double g_a) { return g_a - mu * d_phi_0; };
    double f_l = psi(a_l, phi_0), g_l = dpsi(d_phi_0), f_u = psi(a_u, phi_0), g_u = dpsi(d_phi_0);
    bool interval_converged = (step_max - step_min) < 0, open_interval = true;
    double a_t = std::fmax(std::fmin(step_init, step_max), step_min);
    double x_t[6];
    for (int i = 0; i < 6; i++) x_t[i] = x[i] + dir[i] * a_t;
    set_pose(x_t);
    int rc = derivatives(x_t, true, &cur);
    if (rc != PCM_OK) return rc;
    double phi_t = -cur.score, d_phi_t = -dot6(cur.g, dir);
    double psi_t = psi(a_t, phi_t), d_psi_t = dpsi(d_phi_t);
    while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
      a_t = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t) : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
      a_t = std::fmax(std::fmin(a_t, step_max), step_min);
      for (int i = 0; i < 6; i++) x_t[i] = x[i] + dir[i] * a_t;
      set_pose(x_t);
      rc = derivatives(x_t, false, &cur);
      if (rc != PCM_OK) return rc;
      phi_t = -cur.score; d_phi_t = -dot6(cur.g, dir);
      psi_t = psi(a_t, phi_t); d_psi_t = dpsi(d_phi_t);
      if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
        open_interval = false;
        f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
        f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
      }
      interval_converged = open_interval ? update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t) : update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
      step_iterations++;
    }
    if (step_iterations) {   // computeHessian (double) at the accepted pose, angle tables of the last derivatives pass  :826-829
      n_hess++;
      Eval h;
      rc = eval(2, P, &h);
      if (rc != PCM_OK) return rc;
      std::memcpy(cur.H, h.H, sizeof(cur.H));
    }
    *a_out = a_t;
    return PCM_OK;
  }

  // computeTransformation  :69-156
  int align(const float (&guess)[16], Eval* last, int* iterations, int* converged) {
    gauss_params();
    static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(P.T, std::memcmp(guess, ident, sizeof(ident)) != 0 ? guess : ident, sizeof(ident));
    float R[9], eul[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = P.T[i * 4 + j];
    euler_012(R, eul);
    double p[6] = {P.T[3], P.T[7], P.T[11], eul[0], eul[1], eul[2]};
    Eval cur;
    int rc = derivatives(p, true, &cur);
    if (rc != PCM_OK) return rc;
    int nr = 0;
    bool conv = false;
    while (!conv) {
      double mg[6], delta[6];
      for (int i = 0; i < 6; i++) mg[i] = -cur.g[i];
      svd_solve6(cur.H, mg, delta);
      double nrm = 0;
      for (double d : delta) nrm += d * d;
      nrm = std::sqrt(nrm);
      if (nrm == 0 || nrm != nrm) { conv = nrm == nrm; break; }   // :117-121
      for (double& d : delta) d /= nrm;
      double a = 0;
      rc = step_length(p, delta, nrm, step_size, eps / 2, cur, &a);
      if (rc != PCM_OK) return rc;
      for (int i = 0; i < 6; i++) p[i] += delta[i] * a;
      if (nr > max_iterations || (nr && (std::fabs(a) < eps))) conv = true;
      nr++;
    }
    *last = cur;
    *iterations = nr;
    *converged = conv ? 1 : 0;
    return PCM_OK;
  }
};

}  // namespace ndtomp
}  // namespace pcm

// lsq_step.h -- Gauss-Newton / Levenberg-Marquardt outer-loop arithmetic, usable
// from host and device code (double precision throughout).
//
// Follows fast_gicp's LsqRegistration
// (/root/reference/src/pointcloud_match/fast_gicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp:52-172)
// re-expressed as a per-pair state machine so a whole batch of registrations
// advances with one residual kernel + one step kernel per round and no host
// round trip: LINEARIZE round -> (LM only) TRIAL rounds -> next LINEARIZE.
#pragma once

#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PCM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PCM_HD inline   // plain host build: tests/test_capi_and_host.py replays GN / LM traces through this header with g++
#endif

namespace pcm {

enum PairMode : int32_t { MODE_LINEARIZE = 0, MODE_TRIAL = 1, MODE_DONE = 2, MODE_WAIT = 3, MODE_PENDING = 4 };
// WAIT: queued behind the batch window; PENDING: handed a slot by a pair that finished in this launch, promoted to LINEARIZE by
// its OWN workgroup of the next step launch (never activated inside the launch that other workgroups are still reading)

// Per-pair optimiser state, resident in device memory for the whole align().
struct PairState {
  double x0[16];        // current pose (row-major Isometry3d)
  double xi[16];        // LM trial pose
  double delta[16];     // last update
  double H[36];         // normal matrix of the last linearize
  double b[6];
  double final_hessian[36];
  double y0;            // cost at x0
  double lambda;        // lm_lambda_ (<0: uninitialised)
  double nu;
  double d[6];
  int32_t mode;         // PairMode
  int32_t iter;         // outer iteration index i (nr_iterations_)
  int32_t lm_inner;     // inner LM try index
  int32_t converged;
  int32_t status;
  int32_t num_linearize;
  int32_t num_compute_error;
  int32_t num_inliers;
  double last_cost;
};

struct LsqParams {
  int32_t optimizer;  // 0 GN, 1 LM
  int32_t max_iterations;
  int32_t lm_max_iterations;
  double rotation_eps;
  double translation_eps;
  double lm_init_lambda_factor;
};

PCM_HD void iso_mul(const double* A, const double* B, double* C) {
  double R[16];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[i * 4 + j] = A[i * 4 + 0] * B[j] + A[i * 4 + 1] * B[4 + j] + A[i * 4 + 2] * B[8 + j];
    R[i * 4 + 3] = A[i * 4 + 0] * B[3] + A[i * 4 + 1] * B[7] + A[i * 4 + 2] * B[11] + A[i * 4 + 3];
  }
  R[12] = R[13] = R[14] = 0.0;
  R[15] = 1.0;
  for (int i = 0; i < 16; i++) C[i] = R[i];
}

// so3_exp (so3.hpp:58-77) followed by Quaternion::toRotationMatrix, and the
// translation part: delta = [exp(d[0:3]) | d[3:6]]  (lsq_registration_impl.hpp:114-116)
PCM_HD void delta_from_d(const double* d, double* delta) {
  const double theta_sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = sqrt(theta_sq);
    const double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  const double w = real, x = imag * d[0], y = imag * d[1], z = imag * d[2];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  delta[0] = 1 - (tyy + tzz); delta[1] = txy - twz;       delta[2] = txz + twy;        delta[3] = d[3];
  delta[4] = txy + twz;       delta[5] = 1 - (txx + tzz); delta[6] = tyz - twx;        delta[7] = d[4];
  delta[8] = txz - twy;       delta[9] = tyz + twx;       delta[10] = 1 - (txx + tyy); delta[11] = d[5];
  delta[12] = 0; delta[13] = 0; delta[14] = 0; delta[15] = 1;
}

// balanced tree over n terms, the order of Eigen's unrolled scalar reduction (redux_novec_unroller): f(lo half, hi half)
template <int N>
PCM_HD double eig_tree_sum(const double* v) {
  if constexpr (N == 1) return v[0];
  else return eig_tree_sum<N / 2>(v) + eig_tree_sum<N - N / 2>(v + N / 2);
}
// fixed-size contiguous doubles: Packet2d accumulators combined by the same tree, horizontal add, then the odd tail
template <int N>
PCM_HD double eig_fixed_sum(const double* v) {
  if constexpr (N == 1) {
    return v[0];
  } else {
    constexpr int NP = N / 2;
    double lo[NP], hi[NP];
    for (int i = 0; i < NP; i++) { lo[i] = v[2 * i]; hi[i] = v[2 * i + 1]; }
    double r = eig_tree_sum<NP>(lo) + eig_tree_sum<NP>(hi);
    if constexpr (N % 2) r = r + v[N - 1];
    return r;
  }
}

// Eigen::LDLT<Matrix<double,6,6>>(A).solve(rhs)  (lsq_registration_impl.hpp:111,136), restated from the Eigen sources in
// the reference tree (E = /root/reference/src/pointcloud_match/fast_gicp/thirdparty/Eigen/Eigen/src):
//   compute  E/Cholesky/LDLT.h:497-530 -> ldlt_inplace<Lower>::unblocked :297-390: left-looking, pivot = largest |diagonal|
//            of the trailing block (first of equal maxima), transposition on the LOWER triangle only (the upper one is never read)
//   solve    E/Cholesky/LDLT.h:569-611: P b, unit-lower solve, D^-1 with the pivots <= numeric_limits::min dropped, L^T solve, P^T
// Eigen/Core is not in that tree: the order of additions inside its reductions and the unrolled row-oriented substitution
// for the fixed-size right-hand side follow upstream Eigen 3.4 on SSE2 (assumptions CORE-1..3).
template <int I>
PCM_HD void ldlt6_forward(const double (&m)[6][6], double (&y)[6]) {
  if constexpr (I < 6) {
    double prod[I];
    for (int j = 0; j < I; j++) prod[j] = m[I][j] * y[j];          // row of a column-major matrix: strided -> tree sum
    y[I] -= eig_tree_sum<I>(prod);
    ldlt6_forward<I + 1>(m, y);
  }
}
template <int L>
PCM_HD void ldlt6_backward(const double (&m)[6][6], double (&y)[6]) {
  if constexpr (L < 6) {
    constexpr int i = 6 - L - 1;
    double prod[L];
    for (int j = 0; j < L; j++) prod[j] = m[i + 1 + j][i] * y[i + 1 + j];   // column of L below the diagonal: contiguous -> packets
    y[i] -= eig_fixed_sum<L>(prod);
    ldlt6_backward<L + 1>(m, y);
  }
}
PCM_HD void ldlt6_solve(const double* Ain, const double* rhs, double* x) {
  double m[6][6], temp[6];
  int tr[6];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) m[i][j] = Ain[i * 6 + j];
  for (int k = 0; k < 6; k++) tr[k] = k;
  for (int k = 0; k < 6; k++) {
    int big = k;
    double best = fabs(m[k][k]);
    for (int i = k + 1; i < 6; i++) {
      const double v = fabs(m[i][i]);
      if (v > best) { best = v; big = i; }
    }
    tr[k] = big;
    if (k != big) {
      for (int j = 0; j < k; j++) { const double t = m[k][j]; m[k][j] = m[big][j]; m[big][j] = t; }
      for (int i = big + 1; i < 6; i++) { const double t = m[i][k]; m[i][k] = m[i][big]; m[i][big] = t; }
      { const double t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
      for (int i = k + 1; i < big; i++) { const double t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
    }
    if (k > 0) {   // temp = D(0:k) .* A10^T ; A(k,k) -= A10 . temp ; A21 -= A20 * temp  (strided runtime-size operands: left to right)
      for (int j = 0; j < k; j++) temp[j] = m[j][j] * m[k][j];
      double s = m[k][0] * temp[0];
      for (int j = 1; j < k; j++) s = s + m[k][j] * temp[j];
      m[k][k] -= s;
      for (int i = k + 1; i < 6; i++) {
        double t = m[i][0] * temp[0];
        for (int j = 1; j < k; j++) t = t + m[i][j] * temp[j];
        m[i][k] -= t;
      }
    }
    const double akk = m[k][k];
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) { for (int j = 0; j < 6; j++) tr[j] = j; break; }
    if (k < 5 && valid) for (int i = k + 1; i < 6; i++) m[i][k] /= akk;
  }
  double y[6];
  for (int i = 0; i < 6; i++) y[i] = rhs[i];
  for (int k = 0; k < 6; k++) if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  ldlt6_forward<1>(m, y);
  for (int i = 0; i < 6; i++) y[i] = (fabs(m[i][i]) > 2.2250738585072014e-308) ? y[i] / m[i][i] : 0.0;
  ldlt6_backward<1>(m, y);
  for (int k = 5; k >= 0; k--) if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < 6; i++) x[i] = y[i];
}

// x^3 rounded once: the reference writes std::pow(2 * rho - 1, 3) (lsq_registration_impl.hpp:166), and glibc's pow is correctly
// rounded but for the rarest cases (0.52 ulp bound), whereas c * c * c rounds twice and is an ulp off every few hundred inputs --
// enough to give lambda another last bit and, iterations later, another pose.  Exact square by fma, then the product with the
// error term carried: the result is the correctly rounded cube unless the true value lies within ~1e-16 ulp of a rounding
// boundary.  (The device's own pow() is a 1-2 ulp routine and is not used.)
PCM_HD double cube_rn(double x) {
  const double hi = x * x, lo = fma(x, x, -hi);            // x^2 = hi + lo exactly
  const double r = hi * x, e = fma(hi, x, -r);             // hi * x = r + e exactly
  return r + (e + lo * x);
}

// is_converged  (lsq_registration_impl.hpp:81-91)
PCM_HD bool is_converged(const LsqParams& p, const double* delta) {
  double rmax = 0.0, tmax = 0.0;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      const double r = fabs(delta[i * 4 + j] - (i == j ? 1.0 : 0.0)) * (1.0 / p.rotation_eps);
      rmax = r > rmax ? r : rmax;
    }
    const double t = fabs(delta[i * 4 + 3]) * (1.0 / p.translation_eps);
    tmax = t > tmax ? t : tmax;
  }
  return (rmax > tmax ? rmax : tmax) < 1.0;
}

PCM_HD void solve_damped(PairState& s, double lambda) {
  double A[36], nb[6];
  for (int i = 0; i < 36; i++) A[i] = s.H[i];
  for (int k = 0; k < 6; k++) { A[k * 6 + k] += lambda; nb[k] = -s.b[k]; }
  ldlt6_solve(A, nb, s.d);
  delta_from_d(s.d, s.delta);
}

// end of one outer iteration: `converged_ = is_converged(delta)` and the loop
// condition of computeTransformation (lsq_registration_impl.hpp:63-75)
PCM_HD void finish_outer(PairState& s, const LsqParams& p, bool step_ok) {
  if (!step_ok) {  // "lm not converged!!" -> break (:69-72)
    s.status = -6;
    s.mode = MODE_DONE;
    return;
  }
  s.converged = is_converged(p, s.delta) ? 1 : 0;
  if (s.converged || s.iter + 1 >= p.max_iterations) {
    s.mode = MODE_DONE;
  } else {
    s.iter += 1;
    s.mode = MODE_LINEARIZE;
  }
}

// Called after a LINEARIZE round produced (H, b, cost) at x0.
PCM_HD void after_linearize(PairState& s, const LsqParams& p, const double* H, const double* b, double cost, int inliers) {
  for (int i = 0; i < 36; i++) s.H[i] = H[i];
  for (int i = 0; i < 6; i++) s.b[i] = b[i];
  s.y0 = cost;
  s.last_cost = cost;
  s.num_linearize += 1;
  s.num_inliers = inliers;
  if (p.optimizer == 0) {  // step_gn (:105-122)
    solve_damped(s, 0.0);
    iso_mul(s.delta, s.x0, s.x0);
    for (int i = 0; i < 36; i++) s.final_hessian[i] = s.H[i];
    finish_outer(s, p, true);
    return;
  }
  // step_lm head (:124-143)
  if (s.lambda < 0.0) {
    double mx = 0.0;
    for (int i = 0; i < 6; i++) { const double v = fabs(s.H[i * 6 + i]); mx = v > mx ? v : mx; }
    s.lambda = p.lm_init_lambda_factor * mx;
  }
  s.nu = 2.0;
  s.lm_inner = 0;
  if (p.lm_max_iterations <= 0) { finish_outer(s, p, false); return; }
  solve_damped(s, s.lambda);
  iso_mul(s.delta, s.x0, s.xi);
  s.mode = MODE_TRIAL;
}

// Called after a TRIAL round produced cost yi at xi  (:144-171)
PCM_HD void after_trial(PairState& s, const LsqParams& p, double yi) {
  s.num_compute_error += 1;
  double dp[6];
  for (int k = 0; k < 6; k++) dp[k] = s.d[k] * (s.lambda * s.d[k] - s.b[k]);
  const double den = eig_fixed_sum<6>(dp);   // d.dot(lm_lambda_ * d - b)  lsq_registration_impl.hpp:146: fixed-size 6, packets of two
  const double rho = (s.y0 - yi) / den;
  if (rho < 0) {
    if (is_converged(p, s.delta)) { finish_outer(s, p, true); return; }
    s.lambda = s.nu * s.lambda;
    s.nu = 2 * s.nu;
    s.lm_inner += 1;
    if (s.lm_inner >= p.lm_max_iterations) { finish_outer(s, p, false); return; }
    solve_damped(s, s.lambda);
    iso_mul(s.delta, s.x0, s.xi);
    s.mode = MODE_TRIAL;
    return;
  }
  for (int i = 0; i < 16; i++) s.x0[i] = s.xi[i];
  const double f = 1 - cube_rn(2 * rho - 1);   // std::pow(2 * rho - 1, 3)  lsq_registration_impl.hpp:166
  s.lambda = s.lambda * (f > 1.0 / 3.0 ? f : 1.0 / 3.0);
  for (int i = 0; i < 36; i++) s.final_hessian[i] = s.H[i];
  finish_outer(s, p, true);
}

PCM_HD void init_state(PairState& s, const float* guess) {
  for (int i = 0; i < 16; i++) { s.x0[i] = (double)guess[i]; s.xi[i] = s.x0[i]; s.delta[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  for (int i = 0; i < 36; i++) { s.H[i] = 0.0; s.final_hessian[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (int i = 0; i < 6; i++) { s.b[i] = 0.0; s.d[i] = 0.0; }
  s.y0 = 0.0; s.lambda = -1.0; s.nu = 2.0;
  s.mode = MODE_LINEARIZE; s.iter = 0; s.lm_inner = 0; s.converged = 0; s.status = 0;
  s.num_linearize = 0; s.num_compute_error = 0; s.num_inliers = 0; s.last_cost = 0.0;
}

}  // namespace pcm

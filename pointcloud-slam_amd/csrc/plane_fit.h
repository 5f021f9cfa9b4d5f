// plane_fit.h -- 5-point plane fit of the point-to-plane matcher, in registers.
//
// common::esti_plane (/root/reference/src/jueying_lio/include/common_lib.h:186-243):
// solve A x = -1 with Eigen's ColPivHouseholderQR (float for exactly 5 points,
// double otherwise), n = x/|x|, d = 1/|x|, reject when any |n.p_j + d| > threshold.
// Host+device so the arithmetic can be unit-checked on the CPU (tests/test_capi_and_host.py compiles it with g++ and
// compares it bit for bit with the CPU restatement under test); the shipped path only ever calls it from the HIP kernel.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PCM_HD_FN __host__ __device__
#else
#define PCM_HD_FN
struct float4 { float x, y, z, w; };
#endif

namespace pcm {

constexpr int K = 5;        // options::NUM_MATCH_POINTS      jueying_lio/include/options.h:14
constexpr int KMIN = 3;     // options::MIN_NUM_MATCH_POINTS  jueying_lio/include/options.h:15

// correctly rounded float sqrt / divide on both sides
// (hipcc lowers sqrtf() to v_sqrt_f32 + the +-1 ulp fix-up = correctly rounded;
// __fsqrt_rn() maps to the bare 1-ulp v_sqrt_f32 on ROCm 7.2 and must not be used here)
PCM_HD_FN inline float pcm_sqrtf_rn(float v) { return sqrtf(v); }
PCM_HD_FN inline float pcm_divf_rn(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fdiv_rn(a, b);
#else
  return a / b;
#endif
}

// ---------------------------------------------------------------------------
// Eigen::ColPivHouseholderQR<Matrix<T,R,3>>(A).solve(-ones)  (common_lib.h:199-208, 210-226), restated from the Eigen
// sources in the reference tree (E = /root/reference/src/pointcloud_match/fast_gicp/thirdparty/Eigen/Eigen/src):
//   computeInPlace   E/QR/ColPivHouseholderQR.h:482-580     makeHouseholder  E/Householder/Householder.h:66-98
//   applyHouseholderOnTheLeft  E/Householder/Householder.h:116-137
//   solve            E/QR/ColPivHouseholderQR.h:585-608, reflectors one by one (E/Householder/HouseholderSequence.h:402-413)
// Eigen/Core is not in that tree; the order of additions inside its reductions and the form of its triangular solve are
// restated from upstream Eigen 3.4 as the reference's SSE2 builds run them (assumptions CORE-1 and CORE-2):
// a contiguous run of n scalars is summed packet-wise -- float: ((v0+v2)+(v1+v3)) + v4.. for n >= 4, double:
// (v0+v2)+(v1+v3) for n = 4 -- and left to right below one packet; the back substitution is column-oriented.
// A is column-major in registers: A[col][row].  All loops are fully unrolled so every index is a compile-time constant
// (no scratch memory).
// ---------------------------------------------------------------------------
template <typename T> struct Num;
template <> struct Num<float> {
  static constexpr int kPacket = 4;
  static PCM_HD_FN float eps() { return 1.1920929e-07f; }
  static PCM_HD_FN float tmin() { return 1.17549435e-38f; }
  static PCM_HD_FN float sqrt_(float v) { return pcm_sqrtf_rn(v); }
  static PCM_HD_FN float div_(float a, float b) { return pcm_divf_rn(a, b); }
};
template <> struct Num<double> {
  static constexpr int kPacket = 2;
  static PCM_HD_FN double eps() { return 2.220446049250313e-16; }
  static PCM_HD_FN double tmin() { return 2.2250738585072014e-308; }
  static PCM_HD_FN double sqrt_(double v) { return sqrt(v); }
  static PCM_HD_FN double div_(double a, double b) { return a / b; }
};

// sum of N (<= 2 packets) contiguous scalars in the order Eigen's vectorised reduction adds them
template <typename T, int N>
PCM_HD_FN inline T eig_redux(const T (&v)[N < 1 ? 1 : N]) {
  constexpr int P = Num<T>::kPacket;
  if constexpr (N <= 0) {
    return (T)0;
  } else if constexpr (N < P) {
    T r = v[0];
#pragma unroll
    for (int i = 1; i < N; i++) r = r + v[i];
    return r;
  } else {
    static_assert(N < 3 * P, "eig_redux: at most two packets and a tail");
    T p[P];
#pragma unroll
    for (int l = 0; l < P; l++) p[l] = v[l];
    if constexpr (N >= 2 * P) {
#pragma unroll
      for (int l = 0; l < P; l++) p[l] = p[l] + v[P + l];
    }
    T r;
    if constexpr (P == 4) r = (p[0] + p[2]) + (p[1] + p[3]);
    else r = p[0] + p[1];
#pragma unroll
    for (int i = (N / P) * P; i < N; i++) r = r + v[i];
    return r;
  }
}

// dot of the trailing TL entries of two columns (rows k+1 .. R-1)
template <typename T, int R, int K0>
PCM_HD_FN inline T eig_tail_dot(const T (&x)[R], const T (&y)[R]) {
  constexpr int TL = R - K0 - 1;
  if constexpr (TL <= 0) {
    return (T)0;
  } else {
    T prod[TL];
#pragma unroll
    for (int i = 0; i < TL; i++) prod[i] = x[K0 + 1 + i] * y[K0 + 1 + i];
    return eig_redux<T, TL>(prod);
  }
}

template <typename T, int R, int K0>
PCM_HD_FN inline void colpiv_qr_step(T (&A)[3][R], T (&nu)[3], T (&nd)[3], int (&perm)[3], T (&hc)[3], int& nonzero, T thr_helper, T downdate_thr) {
  constexpr int k = K0;
  int big = k;
  T bign = nu[k];
#pragma unroll
  for (int j = k + 1; j < 3; j++) {
    if (nu[j] > bign) { bign = nu[j]; big = j; }
  }
  if (nonzero == 3 && bign * bign < thr_helper * (T)(R - k)) nonzero = k;
#pragma unroll
  for (int j = k + 1; j < 3; j++) {
    if (big == j) {
#pragma unroll
      for (int i = 0; i < R; i++) { const T t = A[k][i]; A[k][i] = A[j][i]; A[j][i] = t; }
      T t = nu[k]; nu[k] = nu[j]; nu[j] = t;
      t = nd[k]; nd[k] = nd[j]; nd[j] = t;
      const int ti = perm[k]; perm[k] = perm[j]; perm[j] = ti;
    }
  }
  // makeHouseholderInPlace on A[k][k..R)
  const T tail_sq = eig_tail_dot<T, R, k>(A[k], A[k]);
  const T c0 = A[k][k];
  T beta, tau;
  if (tail_sq <= Num<T>::tmin()) {
    tau = 0;
    beta = c0;
#pragma unroll
    for (int i = k + 1; i < R; i++) A[k][i] = 0;
  } else {
    beta = Num<T>::sqrt_(c0 * c0 + tail_sq);
    if (c0 >= 0) beta = -beta;
    const T den = c0 - beta;
#pragma unroll
    for (int i = k + 1; i < R; i++) A[k][i] = Num<T>::div_(A[k][i], den);
    tau = Num<T>::div_(beta - c0, beta);
  }
  A[k][k] = beta;
  hc[k] = tau;
#pragma unroll
  for (int j = k + 1; j < 3; j++) {
    if (R - k == 1) {
      A[j][k] *= ((T)1 - tau);
    } else if (tau != 0) {
      T tmp = eig_tail_dot<T, R, k>(A[k], A[j]);
      tmp += A[j][k];
      A[j][k] -= tau * tmp;
#pragma unroll
      for (int i = k + 1; i < R; i++) A[j][i] -= tmp * (tau * A[k][i]);
    }
  }
#pragma unroll
  for (int j = k + 1; j < 3; j++) {
    if (nu[j] != 0) {
      T temp = Num<T>::div_(fabs(A[j][k]), nu[j]);
      temp = ((T)1 + temp) * ((T)1 - temp);
      temp = temp < 0 ? (T)0 : temp;
      const T r = Num<T>::div_(nu[j], nd[j]);
      const T temp2 = temp * (r * r);
      if (temp2 <= downdate_thr) {
        nd[j] = nu[j] = Num<T>::sqrt_(eig_tail_dot<T, R, k>(A[j], A[j]));
      } else {
        nu[j] *= Num<T>::sqrt_(temp);
      }
    }
  }
}

template <typename T, int R, int K0>
PCM_HD_FN inline void colpiv_qr_reflect_rhs(const T (&A)[3][R], const T (&hc)[3], int nonzero, T (&c)[R]) {
  constexpr int k = K0;
  if (k < nonzero) {
    const T tau = hc[k];
    if (R - k == 1) {
      c[k] *= ((T)1 - tau);
    } else if (tau != 0) {
      T tmp = eig_tail_dot<T, R, k>(A[k], c);
      tmp += c[k];
      c[k] -= tau * tmp;
#pragma unroll
      for (int i = k + 1; i < R; i++) c[i] -= tmp * (tau * A[k][i]);
    }
  }
}

template <typename T, int R>
PCM_HD_FN inline void colpiv_qr_solve(T (&A)[3][R], T (&x)[3]) {
  T nu[3], nd[3];
  int perm[3] = {0, 1, 2};
  T hc[3] = {0, 0, 0};
  T c[R];
  T maxnorm = 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    T prod[R];
#pragma unroll
    for (int i = 0; i < R; i++) prod[i] = A[j][i] * A[j][i];
    nu[j] = nd[j] = Num<T>::sqrt_(eig_redux<T, R>(prod));
    maxnorm = nu[j] > maxnorm ? nu[j] : maxnorm;
  }
  const T thr_helper = Num<T>::div_((maxnorm * Num<T>::eps()) * (maxnorm * Num<T>::eps()), (T)R);
  const T downdate_thr = Num<T>::sqrt_(Num<T>::eps());
  int nonzero = 3;
  colpiv_qr_step<T, R, 0>(A, nu, nd, perm, hc, nonzero, thr_helper, downdate_thr);
  colpiv_qr_step<T, R, 1>(A, nu, nd, perm, hc, nonzero, thr_helper, downdate_thr);
  colpiv_qr_step<T, R, 2>(A, nu, nd, perm, hc, nonzero, thr_helper, downdate_thr);
#pragma unroll
  for (int i = 0; i < R; i++) c[i] = (T)-1;
  colpiv_qr_reflect_rhs<T, R, 0>(A, hc, nonzero, c);
  colpiv_qr_reflect_rhs<T, R, 1>(A, hc, nonzero, c);
  colpiv_qr_reflect_rhs<T, R, 2>(A, hc, nonzero, c);
  // upper-triangular solve on the leading nonzero x nonzero block, column-oriented (runtime-size solve of Eigen)
#pragma unroll
  for (int i = 2; i >= 0; i--) {
    if (i < nonzero && c[i] != 0) {
      c[i] = Num<T>::div_(c[i], A[i][i]);
#pragma unroll
      for (int r = 0; r < i; r++) c[r] -= c[i] * A[i][r];
    }
  }
  x[0] = x[1] = x[2] = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const T v = i < nonzero ? c[i] : (T)0;
#pragma unroll
    for (int p = 0; p < 3; p++) {
      if (perm[i] == p) x[p] = v;
    }
  }
}

// common::esti_plane (common_lib.h:186-243) on m (3..5) neighbours held in registers.
PCM_HD_FN inline bool esti_plane(const float (&px)[K], const float (&py)[K], const float (&pz)[K], int m, float threshold, float4* plane) {
  float nv[3];
  if (m == K) {
    float A[3][K];
#pragma unroll
    for (int j = 0; j < K; j++) { A[0][j] = px[j]; A[1][j] = py[j]; A[2][j] = pz[j]; }
    colpiv_qr_solve<float, K>(A, nv);
  } else if (m == 4) {  // dynamic-size path: solved in double (common_lib.h:210-226)
    double A[3][4], xd[3];
#pragma unroll
    for (int j = 0; j < 4; j++) { A[0][j] = px[j]; A[1][j] = py[j]; A[2][j] = pz[j]; }
    colpiv_qr_solve<double, 4>(A, xd);
    nv[0] = (float)xd[0]; nv[1] = (float)xd[1]; nv[2] = (float)xd[2];
  } else {
    double A[3][3], xd[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { A[0][j] = px[j]; A[1][j] = py[j]; A[2][j] = pz[j]; }
    colpiv_qr_solve<double, 3>(A, xd);
    nv[0] = (float)xd[0]; nv[1] = (float)xd[1]; nv[2] = (float)xd[2];
  }
  const float n = pcm_sqrtf_rn(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
  float4 pl;
  pl.x = pcm_divf_rn(nv[0], n);
  pl.y = pcm_divf_rn(nv[1], n);
  pl.z = pcm_divf_rn(nv[2], n);
  pl.w = (float)(1.0 / (double)n);
  bool ok = true;
#pragma unroll
  for (int j = 0; j < K; j++) {
    if (j < m) {
      const float d = pl.x * px[j] + pl.y * py[j] + pl.z * pz[j] + pl.w;
      if (fabsf(d) > threshold) ok = false;
    }
  }
  *plane = pl;
  return ok;
}

}  // namespace pcm

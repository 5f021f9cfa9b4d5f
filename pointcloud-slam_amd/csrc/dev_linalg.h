// dev_linalg.h -- small double/float linear algebra shared by the device kernels.
#pragma once

#include <hip/hip_runtime.h>

namespace pcm {

// Symmetric 3x3 eigen-decomposition, cyclic Jacobi: eigenvalues ascending in w, eigenvectors in the
// COLUMNS of V (row-major).  Stands in for Eigen::SelfAdjointEigenSolver / JacobiSVD of a PSD matrix.
__device__ inline void eig3_sym_jacobi(const double (&Ain)[9], double (&w)[3], double (&V)[9]) {
  double A[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
    }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (j < 2 - i && w[j] > w[j + 1]) {
        const double t = w[j]; w[j] = w[j + 1]; w[j + 1] = t;
#pragma unroll
        for (int k = 0; k < 3; k++) { const double u = V[k * 3 + j]; V[k * 3 + j] = V[k * 3 + j + 1]; V[k * 3 + j + 1] = u; }
      }
    }
  }
}


// 3x3 inverse by cofactors (Eigen's fixed-size inverse), row-major
template <typename T>
__device__ inline void inv3(const T (&m)[9], T (&inv)[9]) {
  const T c00 = m[4] * m[8] - m[5] * m[7];
  const T c01 = m[5] * m[6] - m[3] * m[8];
  const T c02 = m[3] * m[7] - m[4] * m[6];
  const T det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const T id = (T)1 / det;
  inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

}  // namespace pcm

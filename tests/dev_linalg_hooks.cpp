// Host build of the DEVICE twins of the Eigen restatements (pointcloud-slam_amd/csrc/dev_linalg.h, lsq_step.h, plane_fit.h) behind
// the batch hooks tests/test_eigen_restatements.py drives for the oracle: the same independent numpy checks then cover the code
// the kernels run, not only oracle/orc_eigen.h.  Built by the test with g++ (the headers compile for the host); test
// infrastructure, never shipped.
#include "dev_linalg.h"
#include "lsq_step.h"
#include "plane_fit.h"

using namespace pcm;

template <typename T, int R>
static void qr_batch(long n, const T* A, T* x) {   // A: n x (R x 3) row-major; the device routine takes columns and solves A x = -1
  for (long i = 0; i < n; i++) {
    T c[3][R], xs[3];
    for (int r = 0; r < R; r++) for (int k = 0; k < 3; k++) c[k][r] = A[(i * R + r) * 3 + k];
    colpiv_qr_solve<T, R>(c, xs);
    for (int k = 0; k < 3; k++) x[3 * i + k] = xs[k];
  }
}

extern "C" {
void orc_test_eig_ldlt6(long n, const double* A, const double* b, double* x) {
  for (long i = 0; i < n; i++) ldlt6_solve(A + 36 * i, b + 6 * i, x + 6 * i);
}
// the right-hand side of the plane fit is the constant -1 (common_lib.h:199-208); the device routine has it built in
void orc_test_eig_colpivqr_f(long n, int rows, const float* A, const float* b, float* x) { (void)b; if (rows == 5) qr_batch<float, 5>(n, A, x); }
void orc_test_eig_colpivqr_d(long n, int rows, const double* A, const double* b, double* x) {
  (void)b;
  if (rows == 5) qr_batch<double, 5>(n, A, x);
  else if (rows == 4) qr_batch<double, 4>(n, A, x);
  else qr_batch<double, 3>(n, A, x);
}
void orc_test_eig_jacobi_svd(long n, int dim, const double* A, double* U, double* S, double* V) {
  for (long i = 0; i < n; i++) {
    if (dim == 3) jacobi_svd<3>(A + 9 * i, U + 9 * i, S + 3 * i, V + 9 * i);
    else jacobi_svd<6>(A + 36 * i, U + 36 * i, S + 6 * i, V + 36 * i);
  }
}
void orc_test_eig_svd_solve6(long n, const double* A, const double* b, double* x) { for (long i = 0; i < n; i++) svd_solve6(A + 36 * i, b + 6 * i, x + 6 * i); }
void orc_test_eig_selfadjoint3(long n, const double* A, double* w, double* V, int* ok) { for (long i = 0; i < n; i++) ok[i] = selfadjoint3(A + 9 * i, w + 3 * i, V + 9 * i) ? 1 : 0; }
void orc_test_eig_direct3f(long n, const float* A, float* w, float* V) { for (long i = 0; i < n; i++) selfadjoint3_direct(A + 9 * i, w + 3 * i, V + 9 * i); }
void orc_test_eig_inv3d(long n, const double* A, double* R) {
  for (long i = 0; i < n; i++) { double m[9], r[9]; for (int k = 0; k < 9; k++) m[k] = A[9 * i + k]; inv3<double>(m, r); for (int k = 0; k < 9; k++) R[9 * i + k] = r[k]; }
}
void orc_test_eig_inv3f(long n, const float* A, float* R) {
  for (long i = 0; i < n; i++) { float m[9], r[9]; for (int k = 0; k < 9; k++) m[k] = A[9 * i + k]; inv3<float>(m, r); for (int k = 0; k < 9; k++) R[9 * i + k] = r[k]; }
}
void orc_test_eig_inv4d(long n, const double* A, double* R) { for (long i = 0; i < n; i++) inv4d(A + 16 * i, R + 16 * i); }
}

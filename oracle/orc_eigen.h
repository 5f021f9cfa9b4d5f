/*
 * oracle/orc_eigen.h -- C restatement of the Eigen algorithms the reference's hot path calls.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_linalg.h).
 *
 * Every function follows a source file that IS in the reference tree, under
 *   E = /root/reference/src/pointcloud_match/fast_gicp/thirdparty/Eigen/Eigen/src
 * and cites its line range.  Eigen/Core (E/Core) is NOT in the tree, so three things cannot be cited and are
 * collected here so that they are stated once:
 *
 *   [CORE-1] order of additions inside reductions (`.sum()`, `.dot()`, `.squaredNorm()`, small lazy products).
 *            Restated from upstream Eigen 3.4 Core/Redux.h as the reference's x86 builds use it (SSE2 packets:
 *            fast_gicp/CMakeLists.txt:13-15, ndt_omp/CMakeLists.txt:12-13, jueying_lio plain -O3; no FMA):
 *              - strided or runtime-size-<packet operands: sequential, left to right;
 *              - contiguous operands of n scalars, packet size P (4 float / 2 double), runtime size:
 *                two packet accumulators over [0, n - n % 2P), added, plus one more packet when n % 2P >= P,
 *                horizontal add of the packet (float: (p0+p2)+(p1+p3), double: p0+p1), then the n % P tail
 *                scalars one by one  (orc_redux_f / orc_redux_d below);
 *              - fixed-size contiguous operands: the same packets combined by a balanced tree, then the tail;
 *              - fixed-size strided operands (the unrolled triangular solves of LDLT::solve): balanced binary
 *                tree over the terms (redux_novec_unroller).
 *   [CORE-2] TriangularView::solveInPlace: fixed-size right-hand sides (LDLT, 6x1) run the row-oriented unrolled
 *            substitution (Core/SolveTriangular.h, triangular_solver_unroller); runtime-size ones
 *            (ColPivHouseholderQR::solve) the column-oriented axpy form of Core/products/TriangularSolverVector.h.
 *   [CORE-3] Transpositions * vector applies the swaps k = 0 .. n-1 in that order, the transpose in reverse.
 * None of [CORE-1..3] is pinned by a file in the tree or by a fixture: it changes results at rounding level only, and
 * the independent float64 / brute-force checks in tests/test_eigen_restatements.py bound every function against
 * numpy regardless of these choices.
 *
 * Matrices crossing function boundaries are ROW-MAJOR unless a comment says otherwise.
 */
#ifndef ORC_EIGEN_H
#define ORC_EIGEN_H

#include <math.h>
#include <string.h>
#include <float.h>

/* ---- [CORE-1] reductions -------------------------------------------------------------------- */
/* sum of n contiguous floats, runtime size: redux_impl<LinearVectorizedTraversal, NoUnrolling>, Packet4f */
static inline float orc_redux_f(const float *v, int n) {
  const int P = 4;
  const int asz = (n / P) * P, asz2 = (n / (2 * P)) * (2 * P);
  float res;
  if (asz) {
    float p0[4] = {v[0], v[1], v[2], v[3]};
    if (asz > P) {
      float p1[4] = {v[4], v[5], v[6], v[7]};
      for (int i = 2 * P; i < asz2; i += 2 * P)
        for (int l = 0; l < 4; l++) { p0[l] = p0[l] + v[i + l]; p1[l] = p1[l] + v[i + P + l]; }
      for (int l = 0; l < 4; l++) p0[l] = p0[l] + p1[l];
      if (asz > asz2) for (int l = 0; l < 4; l++) p0[l] = p0[l] + v[asz2 + l];
    }
    res = (p0[0] + p0[2]) + (p0[1] + p0[3]);   /* predux<Packet4f>, SSE2: add(a, movehl(a,a)) then add_ss with lane 1 */
    for (int i = asz; i < n; i++) res = res + v[i];
  } else {
    res = v[0];
    for (int i = 1; i < n; i++) res = res + v[i];
  }
  return res;
}
/* the same for doubles, Packet2d */
static inline double orc_redux_d(const double *v, int n) {
  const int P = 2;
  const int asz = (n / P) * P, asz2 = (n / (2 * P)) * (2 * P);
  double res;
  if (asz) {
    double p0[2] = {v[0], v[1]};
    if (asz > P) {
      double p1[2] = {v[2], v[3]};
      for (int i = 2 * P; i < asz2; i += 2 * P)
        for (int l = 0; l < 2; l++) { p0[l] = p0[l] + v[i + l]; p1[l] = p1[l] + v[i + P + l]; }
      for (int l = 0; l < 2; l++) p0[l] = p0[l] + p1[l];
      if (asz > asz2) for (int l = 0; l < 2; l++) p0[l] = p0[l] + v[asz2 + l];
    }
    res = p0[0] + p0[1];
    for (int i = asz; i < n; i++) res = res + v[i];
  } else {
    res = v[0];
    for (int i = 1; i < n; i++) res = res + v[i];
  }
  return res;
}
/* balanced tree over n terms: redux_novec_unroller<Start, Length>: f(run(Start, L/2), run(Start + L/2, L - L/2)) */
static inline double orc_redux_tree_d(const double *v, int n) {
  if (n == 1) return v[0];
  const int h = n / 2;
  return orc_redux_tree_d(v, h) + orc_redux_tree_d(v + h, n - h);
}
/* fixed-size contiguous doubles (n even, <= 8): packets combined by the same tree (redux_vec_unroller), then predux */
static inline double orc_redux_fixed_d(const double *v, int n) {
  double lo[4], hi[4];
  const int np = n / 2;
  for (int i = 0; i < np; i++) { lo[i] = v[2 * i]; hi[i] = v[2 * i + 1]; }
  double res = orc_redux_tree_d(lo, np) + orc_redux_tree_d(hi, np);
  for (int i = 2 * np; i < n; i++) res = res + v[i];
  return res;
}

/* =============================================================================================
 * LDLT<Matrix<double,6,6>, Lower>::compute + solve
 *   compute  E/Cholesky/LDLT.h:497-530 -> ldlt_inplace<Lower>::unblocked :297-390 (left-looking, diagonal pivoting)
 *   solve    E/Cholesky/LDLT.h:569-611 (_solve_impl_transposed<true>): P b, L^-1, pseudo-inverse of D with
 *            tolerance numeric_limits::min (:593-600), L^-T, P^T
 * Call sites: lsq_registration_impl.hpp:111,136  `Eigen::LDLT<Matrix<double,6,6>> solver(H); d = solver.solve(-b)`.
 * Only the LOWER triangle of A is read (m_matrix is used through its Lower view only).
 * ============================================================================================= */
static inline void orc_eig_ldlt6_solve(const double A_in[36], const double rhs[6], double x[6]) {
  enum { N = 6 };
  double m[N][N], temp[N];
  int tr[N];
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) m[i][j] = A_in[i * N + j];
  for (int k = 0; k < N; k++) tr[k] = k;
  for (int k = 0; k < N; k++) {
    /* :320-322 largest |diagonal| of the trailing block (maxCoeff: first of equal maxima) */
    int big = k;
    double best = fabs(m[k][k]);
    for (int i = k + 1; i < N; i++) if (fabs(m[i][i]) > best) { best = fabs(m[i][i]); big = i; }
    tr[k] = big;
    if (k != big) {   /* :325-341 transposition on the lower triangle only */
      for (int j = 0; j < k; j++) { const double t = m[k][j]; m[k][j] = m[big][j]; m[big][j] = t; }
      for (int i = big + 1; i < N; i++) { const double t = m[i][k]; m[i][k] = m[i][big]; m[i][big] = t; }
      { const double t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
      for (int i = k + 1; i < big; i++) { const double t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
    }
    /* :352-358  temp = D(0:k) .* A10^T ;  A(k,k) -= A10 . temp ;  A21 -= A20 * temp   (runtime-size blocks of a
     * column-major matrix: A10 and the rows of A20 are strided -> sequential sums, [CORE-1]) */
    const int rs = N - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = m[j][j] * m[k][j];
      double s = m[k][0] * temp[0];
      for (int j = 1; j < k; j++) s = s + m[k][j] * temp[j];
      m[k][k] -= s;
      for (int i = k + 1; i < N; i++) {
        double t = m[i][0] * temp[0];
        for (int j = 1; j < k; j++) t = t + m[i][j] * temp[j];
        m[i][k] -= t;
      }
    }
    /* :364-384 */
    const double akk = m[k][k];
    const int valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) { for (int j = 0; j < N; j++) tr[j] = j; break; }   /* :367-378 whole diagonal zero */
    if (rs > 0 && valid) for (int i = k + 1; i < N; i++) m[i][k] /= akk;
  }
  /* ---- solve :569-611 ---- */
  double y[N], prod[N];
  for (int i = 0; i < N; i++) y[i] = rhs[i];
  for (int k = 0; k < N; k++) if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }   /* [CORE-3] */
  /* matrixL().solveInPlace: unit lower, fixed size 6 -> unrolled row-oriented substitution [CORE-2]; the row of a
   * column-major matrix is strided -> tree sum [CORE-1] */
  for (int i = 1; i < N; i++) {
    for (int j = 0; j < i; j++) prod[j] = m[i][j] * y[j];
    y[i] -= orc_redux_tree_d(prod, i);
  }
  for (int i = 0; i < N; i++) {   /* :593-600 */
    if (fabs(m[i][i]) > DBL_MIN) y[i] /= m[i][i];
    else y[i] = 0.0;
  }
  /* matrixL().transpose().solveInPlace: unit upper view of the transpose; its rows are the columns of L below the
   * diagonal, i.e. contiguous -> packets where the segment is long enough [CORE-1] (segment<LoopIndex>, fixed size) */
  for (int l = 1; l < N; l++) {
    const int i = N - l - 1;   /* DiagIndex, StartIndex = i + 1, LoopIndex = l */
    for (int j = 0; j < l; j++) prod[j] = m[i + 1 + j][i] * y[i + 1 + j];
    y[i] -= (l >= 2) ? orc_redux_fixed_d(prod, l) : prod[0];
  }
  for (int k = N - 1; k >= 0; k--) if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < N; i++) x[i] = y[i];
}

/* =============================================================================================
 * ColPivHouseholderQR<Matrix<T, rows, 3>>(A).solve(b)
 *   computeInPlace  E/QR/ColPivHouseholderQR.h:482-580
 *   makeHouseholder E/Householder/Householder.h:66-98; applyHouseholderOnTheLeft :116-137
 *   solve           E/QR/ColPivHouseholderQR.h:585-608 (nonzeroPivots(), householderQ().setLength().adjoint() applied
 *                   reflector by reflector: E/Householder/HouseholderSequence.h:402-413), back substitution [CORE-2]
 * Call sites: jueying_lio/include/common_lib.h:208 (Matrix<float,5,3>, fixed) and :223 (Matrix<double,Dynamic,3>).
 * Storage here is COLUMN-major a[col][row] like the reference's matrices, because the reductions run along
 * contiguous column segments [CORE-1].  Both instantiations have runtime-size segments (tail(), bottomRightCorner())
 * except the initial column norms of the fixed 5x3 float matrix, whose fixed size 5 gives the same order.
 * ============================================================================================= */
#define ORC_EIG_QR_MAXR 32
#define ORC_DEF_EIG_COLPIVQR(NAME, T, SQRT, FABS, EPS, TMIN, REDUX)                                        \
  static inline void NAME(const T *A_rowmajor, int rows, const T *b_in, T x[3]) {                           \
    const int cols = 3;                                                                                     \
    const int size = rows < cols ? rows : cols;                                                             \
    T a[3][ORC_EIG_QR_MAXR], prod[ORC_EIG_QR_MAXR], c[ORC_EIG_QR_MAXR];                                     \
    T hcoef[3] = {0, 0, 0}, nu[3], nd[3], tmpv[3];                                                          \
    int trn[3] = {0, 1, 2};                                                                                 \
    for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) a[j][i] = A_rowmajor[i * 3 + j];          \
    T maxnorm = 0;                                                                                          \
    for (int j = 0; j < cols; j++) {   /* :503-508 col(k).norm() */                                        \
      for (int i = 0; i < rows; i++) prod[i] = a[j][i] * a[j][i];                                           \
      nd[j] = nu[j] = SQRT(REDUX(prod, rows));                                                              \
      if (nu[j] > maxnorm) maxnorm = nu[j];                                                                 \
    }                                                                                                       \
    const T thr_helper = ((maxnorm * EPS) * (maxnorm * EPS)) / (T)rows;   /* :510 abs2(max * eps) / rows */ \
    const T downdate_thr = SQRT(EPS);                                      /* :511 */                       \
    int nonzero = size;                                                                                     \
    for (int k = 0; k < size; k++) {                                                                        \
      int big = k;                                                         /* :519-521 first of equal maxima */ \
      T bign = nu[k];                                                                                       \
      for (int j = k + 1; j < cols; j++) if (nu[j] > bign) { bign = nu[j]; big = j; }                       \
      const T big_sq = bign * bign;                                                                         \
      if (nonzero == size && big_sq < thr_helper * (T)(rows - k)) nonzero = k;   /* :525-526 */             \
      trn[k] = big;                                                                                         \
      if (k != big) {                                                      /* :529-535 */                   \
        for (int i = 0; i < rows; i++) { const T t = a[k][i]; a[k][i] = a[big][i]; a[big][i] = t; }         \
        T t = nu[k]; nu[k] = nu[big]; nu[big] = t;                                                          \
        t = nd[k]; nd[k] = nd[big]; nd[big] = t;                                                            \
      }                                                                                                     \
      /* :538-539 makeHouseholderInPlace on col(k).tail(rows-k)   Householder.h:66-98 */                    \
      const int tl = rows - k - 1;                                                                          \
      T tail_sq = 0;                                                                                        \
      if (tl > 0) { for (int i = 0; i < tl; i++) prod[i] = a[k][k + 1 + i] * a[k][k + 1 + i]; tail_sq = REDUX(prod, tl); } \
      const T c0 = a[k][k];                                                                                 \
      T beta, tau;                                                                                          \
      if (tail_sq <= TMIN) {                                                                                \
        tau = 0; beta = c0;                                                                                 \
        for (int i = k + 1; i < rows; i++) a[k][i] = 0;                                                     \
      } else {                                                                                              \
        beta = SQRT(c0 * c0 + tail_sq);                                                                     \
        if (c0 >= 0) beta = -beta;                                                                          \
        const T den = c0 - beta;                                                                            \
        for (int i = k + 1; i < rows; i++) a[k][i] = a[k][i] / den;                                         \
        tau = (beta - c0) / beta;                                                                           \
      }                                                                                                     \
      a[k][k] = beta;                                                      /* :542 */                       \
      hcoef[k] = tau;                                                                                       \
      /* :548-549 bottomRightCorner(rows-k, cols-k-1).applyHouseholderOnTheLeft   Householder.h:116-137 */   \
      if (cols - k - 1 > 0) {                                                                               \
        if (rows - k == 1) {                                                                                \
          for (int j = k + 1; j < cols; j++) a[j][k] *= ((T)1 - tau);                                       \
        } else if (tau != 0) {                                                                              \
          for (int j = k + 1; j < cols; j++) {                             /* tmp = essential^T * bottom */ \
            for (int i = 0; i < tl; i++) prod[i] = a[k][k + 1 + i] * a[j][k + 1 + i];                       \
            tmpv[j] = REDUX(prod, tl);                                                                      \
          }                                                                                                 \
          for (int j = k + 1; j < cols; j++) tmpv[j] += a[j][k];           /* tmp += row(0) */              \
          for (int j = k + 1; j < cols; j++) a[j][k] -= tau * tmpv[j];     /* row(0) -= tau * tmp */        \
          for (int j = k + 1; j < cols; j++)                               /* bottom -= (tau * essential) * tmp */ \
            for (int i = k + 1; i < rows; i++) a[j][i] -= tmpv[j] * (tau * a[k][i]);                        \
        }                                                                                                   \
      }                                                                                                     \
      /* :552-571 norm down-date */                                                                         \
      for (int j = k + 1; j < cols; j++) {                                                                  \
        if (nu[j] != 0) {                                                                                   \
          T temp = FABS(a[j][k]) / nu[j];                                                                   \
          temp = ((T)1 + temp) * ((T)1 - temp);                                                             \
          temp = temp < 0 ? (T)0 : temp;                                                                    \
          const T r = nu[j] / nd[j];                                                                        \
          const T temp2 = temp * (r * r);                                                                   \
          if (temp2 <= downdate_thr) {                                                                      \
            T s = 0;                                                                                        \
            if (tl > 0) { for (int i = 0; i < tl; i++) prod[i] = a[j][k + 1 + i] * a[j][k + 1 + i]; s = REDUX(prod, tl); } \
            nd[j] = nu[j] = SQRT(s);                                                                        \
          } else {                                                                                          \
            nu[j] *= SQRT(temp);                                                                            \
          }                                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
    }                                                                                                       \
    /* :574-576 permutation from the transpositions: indices(k) <-> indices(trn[k]) for k = 0..size-1 */     \
    int perm[3] = {0, 1, 2};                                                                                \
    for (int k = 0; k < size; k++) { const int t = perm[k]; perm[k] = perm[trn[k]]; perm[trn[k]] = t; }     \
    x[0] = x[1] = x[2] = 0;                                                                                 \
    if (nonzero == 0) return;                                              /* :589-593 */                   \
    for (int i = 0; i < rows; i++) c[i] = b_in[i];                                                          \
    for (int k = 0; k < nonzero; k++) {                                    /* :597  H_0, H_1, ... in turn */ \
      const T tau = hcoef[k];                                                                               \
      const int tl = rows - k - 1;                                                                          \
      if (rows - k == 1) { c[k] *= ((T)1 - tau); continue; }                                                \
      if (tau == 0) continue;                                                                               \
      for (int i = 0; i < tl; i++) prod[i] = a[k][k + 1 + i] * c[k + 1 + i];                                \
      T tmp = REDUX(prod, tl);                                                                              \
      tmp += c[k];                                                                                          \
      c[k] -= tau * tmp;                                                                                    \
      for (int i = k + 1; i < rows; i++) c[i] -= tmp * (tau * a[k][i]);                                     \
    }                                                                                                       \
    /* :599-601 upper-triangular solve, runtime size, column-major: column-oriented [CORE-2] */              \
    for (int i = nonzero - 1; i >= 0; i--) {                                                                \
      if (c[i] != 0) {                                                                                      \
        c[i] = c[i] / a[i][i];                                                                              \
        for (int r = 0; r < i; r++) c[r] -= c[i] * a[i][r];                                                 \
      }                                                                                                     \
    }                                                                                                       \
    for (int i = 0; i < nonzero; i++) x[perm[i]] = c[i];                   /* :603-604 */                   \
  }
ORC_DEF_EIG_COLPIVQR(orc_eig_colpivqr3f, float, sqrtf, fabsf, FLT_EPSILON, FLT_MIN, orc_redux_f)
ORC_DEF_EIG_COLPIVQR(orc_eig_colpivqr3d, double, sqrt, fabs, DBL_EPSILON, DBL_MIN, orc_redux_d)

/* =============================================================================================
 * JacobiRotation  E/Jacobi/Jacobi.h
 * ============================================================================================= */
/* makeJacobi(x, y, z) :92-125 : J with J^T [x y; y z] J diagonal */
static inline int orc_eig_make_jacobi(double x, double y, double z, double *c, double *s) {
  const double deno = 2.0 * fabs(y);
  if (deno < DBL_MIN) { *c = 1.0; *s = 0.0; return 0; }
  const double tau = (x - z) / deno;
  const double w = sqrt(tau * tau + 1.0);
  const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0.0 ? 1.0 : -1.0;
  const double n = 1.0 / sqrt(t * t + 1.0);
  *s = -sign_t * (y / fabs(y)) * fabs(t) * n;
  *c = n;
  return 1;
}
/* makeGivens(p, q) :228-262 (real) : G^T (p, q)^T = (r, 0)^T */
static inline void orc_eig_make_givens(double p, double q, double *c, double *s) {
  if (q == 0.0) { *c = p < 0.0 ? -1.0 : 1.0; *s = 0.0; }
  else if (p == 0.0) { *c = 0.0; *s = q < 0.0 ? 1.0 : -1.0; }
  else if (fabs(p) > fabs(q)) {
    const double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    *c = 1.0 / u;
    *s = -t * *c;
  } else {
    const double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    *s = -1.0 / u;
    *c = -t * *s;
  }
}
/* apply_rotation_in_the_plane :329-340 (and its packet form, same arithmetic per element): x' = c x + s y ; y' = -s x + c y */
#define ORC_EIG_ROT(xv, yv, c, s) do { const double xi_ = (xv), yi_ = (yv); (xv) = (c) * xi_ + (s) * yi_; (yv) = -(s) * xi_ + (c) * yi_; } while (0)

/* =============================================================================================
 * JacobiSVD<Matrix<double,n,n>> (n <= 6, square: no QR preconditioner step runs), ComputeFullU | ComputeFullV
 *   compute            E/SVD/JacobiSVD.h:667-797 (two-sided Jacobi sweeps p = 1..n-1, q = 0..p-1)
 *   2x2 real SVD       E/misc/RealSvd2x2.h:19-51
 *   precondition (real scalars: only the maxDiagEntry / threshold part) E/SVD/JacobiSVD.h:350-358
 * Outputs row-major U (n x n), S descending, V.  U or V may be NULL.
 * Call sites: fast_gicp_impl.hpp:273 (3x3 covariances), ndt_omp_impl.hpp:112 (6x6 Hessian),
 *             gicp_omp_impl.hpp:110 (3x3).
 * ============================================================================================= */
static inline void orc_eig_jacobi_svd(int n, const double *A, double *Uo, double *So, double *Vo) {
  double w[6][6], U[6][6], V[6][6], sv[6];
  const double precision = 2.0 * DBL_EPSILON, considerAsZero = DBL_MIN;
  double scale = 0.0;
  for (int i = 0; i < n * n; i++) { const double a = fabs(A[i]); if (a > scale || a != a) scale = a; }   /* :680 maxCoeff<PropagateNaN> */
  if (!isfinite(scale)) {   /* :681-685 InvalidInput: the members keep whatever they held; report zeros */
    for (int i = 0; i < n; i++) { So[i] = 0.0; for (int j = 0; j < n; j++) { if (Uo) Uo[i * n + j] = i == j; if (Vo) Vo[i * n + j] = i == j; } }
    return;
  }
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { w[i][j] = A[i * n + j] / scale; U[i][j] = V[i][j] = (i == j) ? 1.0 : 0.0; }   /* :697-702 */
  double maxDiag = 0.0;
  for (int i = 0; i < n; i++) if (fabs(w[i][i]) > maxDiag) maxDiag = fabs(w[i][i]);   /* :706 */
  int finished = 0;
  while (!finished) {
    finished = 1;
    for (int p = 1; p < n; p++) {
      for (int q = 0; q < p; q++) {
        double thr = precision * maxDiag; if (considerAsZero > thr) thr = considerAsZero;   /* :722 */
        if (fabs(w[p][q]) > thr || fabs(w[q][p]) > thr) {
          finished = 0;
          /* svd_precondition_2x2_block_to_be_real, real scalar :350-358 -> returns true; the IsComplex=true body
           * (:361-424) is not instantiated for double */
          {
            /* real_2x2_jacobi_svd  RealSvd2x2.h:19-51 */
            double m00 = w[p][p], m01 = w[p][q], m10 = w[q][p], m11 = w[q][q];
            double r1c, r1s;
            const double t = m00 + m11, d = m10 - m01;
            if (fabs(d) < DBL_MIN) { r1s = 0.0; r1c = 1.0; }
            else { const double u = t / d; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
            /* m.applyOnTheLeft(0, 1, rot1): rows 0 and 1 */
            ORC_EIG_ROT(m00, m10, r1c, r1s);
            ORC_EIG_ROT(m01, m11, r1c, r1s);
            double jrc, jrs;
            orc_eig_make_jacobi(m00, m01, m11, &jrc, &jrs);           /* j_right->makeJacobi(m, 0, 1) */
            /* *j_left = rot1 * j_right->transpose()  Jacobi.h:49-55: (c1 c2 - s1 s2', ...) with j_right^T = (c, -s) */
            const double tc = jrc, ts = -jrs;
            const double jlc = r1c * tc - r1s * ts;
            const double jls = r1c * ts + r1s * tc;
            /* :732-739 */
            for (int j = 0; j < n; j++) ORC_EIG_ROT(w[p][j], w[q][j], jlc, jls);            /* work.applyOnTheLeft(p,q,j_left) */
            for (int i = 0; i < n; i++) ORC_EIG_ROT(U[i][p], U[i][q], jlc, jls);            /* U.applyOnTheRight(p,q,j_left.transpose()): applies (j_left^T)^T = j_left to the columns */
            for (int i = 0; i < n; i++) ORC_EIG_ROT(w[i][p], w[i][q], jrc, -jrs);           /* work.applyOnTheRight(p,q,j_right): j_right.transpose() on the columns */
            for (int i = 0; i < n; i++) ORC_EIG_ROT(V[i][p], V[i][q], jrc, -jrs);
            double mx = fabs(w[p][p]) > fabs(w[q][q]) ? fabs(w[p][p]) : fabs(w[q][q]);   /* :742 */
            if (mx > maxDiag) maxDiag = mx;
          }
        }
      }
    }
  }
  for (int i = 0; i < n; i++) {   /* :750-769 */
    const double a = w[i][i];
    sv[i] = fabs(a);
    if (a < 0.0) for (int r = 0; r < n; r++) U[r][i] = -U[r][i];
  }
  for (int i = 0; i < n; i++) sv[i] *= scale;   /* :771 */
  for (int i = 0; i < n; i++) {   /* :775-793 selection sort, descending (maxCoeff: first of equal maxima) */
    int pos = i;
    double mxv = sv[i];
    for (int j = i + 1; j < n; j++) if (sv[j] > mxv) { mxv = sv[j]; pos = j; }
    if (mxv == 0.0) break;
    if (pos != i) {
      const double t = sv[i]; sv[i] = sv[pos]; sv[pos] = t;
      for (int r = 0; r < n; r++) { double u = U[r][pos]; U[r][pos] = U[r][i]; U[r][i] = u; u = V[r][pos]; V[r][pos] = V[r][i]; V[r][i] = u; }
    }
  }
  for (int i = 0; i < n; i++) { So[i] = sv[i]; for (int j = 0; j < n; j++) { if (Uo) Uo[i * n + j] = U[i][j]; if (Vo) Vo[i * n + j] = V[i][j]; } }
}

/* JacobiSVD<Matrix<double,6,6>>(H, ComputeFullU | ComputeFullV).solve(b)
 *   rank()       E/SVD/SVDBase.h:148-157 with threshold() :198-205 = diagSize * epsilon
 *   _solve_impl  E/SVD/SVDBase.h:308-318:  tmp = U(:, :rank)^T b ; tmp = S^-1 tmp ; x = V(:, :rank) tmp
 * Call site: ndt_omp_impl.hpp:112-114. */
static inline void orc_eig_svd_solve6(const double H[36], const double b[6], double x[6]) {
  double U[36], V[36], S[6], tmp[6], prod[6];
  orc_eig_jacobi_svd(6, H, U, S, V);
  double pre = S[0] * (6.0 * DBL_EPSILON);
  if (DBL_MIN > pre) pre = DBL_MIN;
  int nz = 6;
  for (int i = 0; i < 6; i++) if (S[i] == 0.0) { nz = i; break; }   /* m_nonzeroSingularValues  JacobiSVD.h:775-783 */
  int rank = nz;
  while (rank > 0 && S[rank - 1] < pre) rank--;
  for (int j = 0; j < rank; j++) {   /* column of U (contiguous, fixed size 6) . b  [CORE-1] */
    for (int k = 0; k < 6; k++) prod[k] = U[k * 6 + j] * b[k];
    tmp[j] = orc_redux_fixed_d(prod, 6);
  }
  for (int j = 0; j < rank; j++) tmp[j] = (1.0 / S[j]) * tmp[j];   /* asDiagonal().inverse() * tmp */
  for (int i = 0; i < 6; i++) {      /* row of V (strided), runtime length rank: sequential [CORE-1] */
    double s = 0.0;
    if (rank > 0) { s = V[i * 6 + 0] * tmp[0]; for (int j = 1; j < rank; j++) s = s + V[i * 6 + j] * tmp[j]; }
    x[i] = s;
  }
}

/* =============================================================================================
 * SelfAdjointEigenSolver<Matrix3d>::compute(A)   (iterative: tridiagonalisation + implicit symmetric QR)
 *   compute                       E/Eigenvalues/SelfAdjointEigenSolver.h:412-462 (lower triangle, scaled to [-1, 1])
 *   tridiagonalization (3x3 real) E/Eigenvalues/Tridiagonalization.h:464-503
 *   computeFromTridiagonal_impl   E/Eigenvalues/SelfAdjointEigenSolver.h:498-566 (deflation test, max 30 n iterations, sort)
 *   tridiagonal_qr_step           E/Eigenvalues/SelfAdjointEigenSolver.h:838-895 (Wilkinson shift, bulge chasing)
 * Eigenvalues ascending in w, eigenvectors in the COLUMNS of V (row-major storage).  Returns 1 on Success.
 * Call site: voxel_grid_covariance_omp_impl.hpp:333.
 * ============================================================================================= */
static inline int orc_eig_selfadjoint3(const double A[9], double w[3], double V[9]) {
  double mat[3][3], diag[3], sub[2];
  /* mat = lower triangle (:443), scale by max |coeff| (:444-446) */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mat[i][j] = (j <= i) ? A[i * 3 + j] : 0.0;
  double scale = 0.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (fabs(mat[i][j]) > scale) scale = fabs(mat[i][j]);
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++) mat[i][j] /= scale;
  /* Tridiagonalization.h:464-503 */
  double Q[3][3];
  {
    diag[0] = mat[0][0];
    const double v1norm2 = mat[2][0] * mat[2][0];
    if (v1norm2 <= DBL_MIN) {
      diag[1] = mat[1][1]; diag[2] = mat[2][2]; sub[0] = mat[1][0]; sub[1] = mat[2][1];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] = i == j;
    } else {
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
  }
  /* computeFromTridiagonal_impl :498-566 */
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
    /* tridiagonal_qr_step :838-895 */
    {
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
        orc_eig_make_givens(xx, z, &c, &s);
        const double sdk = s * diag[k] + c * sub[k];
        const double dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k] = c * sdk - s * dkp1;
        if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
        xx = sub[k];
        if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
        /* q.applyOnTheRight(k, k+1, rot): rot.transpose() = (c, -s) on columns k, k+1 */
        for (int r = 0; r < 3; r++) ORC_EIG_ROT(Q[r][k], Q[r][k + 1], c, -s);
      }
    }
  }
  const int ok = iter <= maxIterations * n;
  if (ok) {   /* :548-563 selection sort ascending (minCoeff: first of equal minima) */
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
  for (int i = 0; i < 3; i++) { w[i] = diag[i] * scale; for (int j = 0; j < 3; j++) V[i * 3 + j] = Q[i][j]; }   /* :455 */
  return ok;
}

/* =============================================================================================
 * SelfAdjointEigenSolver<Matrix3f>::computeDirect(A)   (closed form; the reference's CUDA kernels)
 *   run / computeRoots / extract_kernel   E/Eigenvalues/SelfAdjointEigenSolver.h:577-733
 * float throughout, device math replaced by libm's correctly-named single-precision functions.
 * Eigenvalues ascending in w, eigenvectors in the COLUMNS of V (row-major).  Reads the lower triangle.
 * Call site: covariance_regularization.cu:18-20 (svd_kernel), :42,93.
 * 3-term sums (trace, squaredNorm, dot) are fixed-size 3 floats: below the float packet size -> sequential [CORE-1].
 * ============================================================================================= */
static inline void orc_eig_direct3f_kernel(float t[3][3], float res[3], float rep[3]) {   /* extract_kernel :634-653 */
  int i0 = 0;
  float best = fabsf(t[0][0]);
  for (int i = 1; i < 3; i++) if (fabsf(t[i][i]) > best) { best = fabsf(t[i][i]); i0 = i; }
  for (int r = 0; r < 3; r++) rep[r] = t[r][i0];
  const int i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
  float c0[3], c1[3];
  /* cross(a, b) = (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0)   E/Geometry/OrthoMethods.h */
  c0[0] = rep[1] * t[2][i1] - rep[2] * t[1][i1]; c0[1] = rep[2] * t[0][i1] - rep[0] * t[2][i1]; c0[2] = rep[0] * t[1][i1] - rep[1] * t[0][i1];
  c1[0] = rep[1] * t[2][i2] - rep[2] * t[1][i2]; c1[1] = rep[2] * t[0][i2] - rep[0] * t[2][i2]; c1[2] = rep[0] * t[1][i2] - rep[1] * t[0][i2];
  const float n0 = (c0[0] * c0[0] + c0[1] * c0[1]) + c0[2] * c0[2];
  const float n1 = (c1[0] * c1[0] + c1[1] * c1[1]) + c1[2] * c1[2];
  if (n0 > n1) { const float s = sqrtf(n0); for (int r = 0; r < 3; r++) res[r] = c0[r] / s; }
  else { const float s = sqrtf(n1); for (int r = 0; r < 3; r++) res[r] = c1[r] / s; }
}
static inline void orc_eig_direct3f(const float A[9], float w[3], float V[9]) {
  float sm[3][3], ev[3], vec[3][3];   /* vec[r][c]: column c = eigenvector c */
  const float shift = ((A[0] + A[4]) + A[8]) / 3.0f;                               /* :669 */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sm[i][j] = A[(i > j ? i : j) * 3 + (i > j ? j : i)];   /* :671 selfadjointView<Lower> */
  for (int i = 0; i < 3; i++) sm[i][i] -= shift;
  float scale = 0.0f;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (fabsf(sm[i][j]) > scale) scale = fabsf(sm[i][j]);
  if (scale > 0.0f) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sm[i][j] /= scale;
  {   /* computeRoots :589-631 */
    const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = sqrtf(3.0f);
#define M_(i, j) sm[i][j]
    const float c0 = M_(0,0) * M_(1,1) * M_(2,2) + 2.0f * M_(1,0) * M_(2,0) * M_(2,1) - M_(0,0) * M_(2,1) * M_(2,1) - M_(1,1) * M_(2,0) * M_(2,0) - M_(2,2) * M_(1,0) * M_(1,0);
    const float c1 = M_(0,0) * M_(1,1) - M_(1,0) * M_(1,0) + M_(0,0) * M_(2,2) - M_(2,0) * M_(2,0) + M_(1,1) * M_(2,2) - M_(2,1) * M_(2,1);
    const float c2 = M_(0,0) + M_(1,1) + M_(2,2);
#undef M_
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
    a_over_3 = a_over_3 > 0.0f ? a_over_3 : 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    q = q > 0.0f ? q : 0.0f;
    const float rho = sqrtf(a_over_3);
    /* atan2 / cos / sin: evaluated in double and rounded to float on both sides of the parity tests (the reference runs
     * CUDA's device libm here, whose last-ulp behaviour no host library reproduces) */
    const float theta = (float)atan2((double)sqrtf(q), (double)half_b) * s_inv3;
    const float cos_theta = (float)cos((double)theta), sin_theta = (float)sin((double)theta);
    ev[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    ev[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    ev[2] = c2_over_3 + 2.0f * rho * cos_theta;
  }
  if ((ev[2] - ev[0]) <= FLT_EPSILON) {   /* :681-685 */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) vec[i][j] = i == j;
  } else {
    float tmp[3][3], colk[3], coll[3];
    float d0 = ev[2] - ev[1], d1 = ev[1] - ev[0];
    int k = 0, l = 2;
    if (d0 > d1) { k = 2; l = 0; d0 = d1; }
    memcpy(tmp, sm, sizeof(tmp));
    for (int i = 0; i < 3; i++) tmp[i][i] -= ev[k];
    orc_eig_direct3f_kernel(tmp, colk, coll);   /* :702-706: eivecs.col(k) = kernel, eivecs.col(l) = representative */
    if (d0 <= 2.0f * FLT_EPSILON * d1) {   /* :709-715 */
      const float dt = (colk[0] * coll[0] + colk[1] * coll[1]) + colk[2] * coll[2];
      for (int r = 0; r < 3; r++) coll[r] -= dt * coll[r];
      const float nn = sqrtf((coll[0] * coll[0] + coll[1] * coll[1]) + coll[2] * coll[2]);
      for (int r = 0; r < 3; r++) coll[r] /= nn;
    } else {                               /* :716-723 */
      float dummy[3];
      memcpy(tmp, sm, sizeof(tmp));
      for (int i = 0; i < 3; i++) tmp[i][i] -= ev[l];
      orc_eig_direct3f_kernel(tmp, coll, dummy);
    }
    for (int r = 0; r < 3; r++) { vec[r][k] = colk[r]; vec[r][l] = coll[r]; }
    /* :726 col(1) = col(2).cross(col(0)).normalized() */
    float cr[3] = {vec[1][2] * vec[2][0] - vec[2][2] * vec[1][0], vec[2][2] * vec[0][0] - vec[0][2] * vec[2][0], vec[0][2] * vec[1][0] - vec[1][2] * vec[0][0]};
    const float n2 = (cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2];
    if (n2 > 0.0f) { const float nn = sqrtf(n2); for (int r = 0; r < 3; r++) cr[r] /= nn; }   /* normalized(): divides only when the norm is > 0 */
    for (int r = 0; r < 3; r++) vec[r][1] = cr[r];
  }
  for (int i = 0; i < 3; i++) { w[i] = ev[i] * scale + shift; for (int j = 0; j < 3; j++) V[i * 3 + j] = vec[i][j]; }   /* :730-731 */
}

/* =============================================================================================
 * Matrix3 inverse (fixed size): compute_inverse<MatrixType, ResultType, 3>   E/LU/InverseImpl.h:125-176
 *   cofactors of column 0, det = (cofactors_col0 .* col(0)).sum() (3 terms: [CORE-1] (c0 m00 + c1 m10) + c2 m20 for
 *   double, Packet2d + tail, and sequential for float, below the packet size: the same order), invdet = 1 / det,
 *   result(r, c) = cofactor<c, r> * invdet.
 * ============================================================================================= */
#define ORC_DEF_EIG_INV3(NAME, T)                                                                     \
  static inline void NAME(const T m[9], T inv[9]) {                                                  \
    /* cofactor_3x3<i,j> = m(i1,j1) m(i2,j2) - m(i1,j2) m(i2,j1), i1 = (i+1)%3 ... :127-137 */        \
    T cof[3][3];                                                                                     \
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {                                        \
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;              \
      cof[i][j] = m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];                 \
    }                                                                                                \
    const T det = (cof[0][0] * m[0] + cof[1][0] * m[3]) + cof[2][0] * m[6];                          \
    const T invdet = (T)1 / det;                                                                     \
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) inv[r * 3 + c] = cof[c][r] * invdet;     \
  }
ORC_DEF_EIG_INV3(orc_eig_inv3d, double)
ORC_DEF_EIG_INV3(orc_eig_inv3f, float)

/* =============================================================================================
 * Matrix4d::inverse(): compute_inverse_size4<Architecture::Target, double, ...>  (the Packet2d form every x86-64
 * build of the reference uses)   E/LU/arch/InverseSize4.h:166-351, column-major operands (StorageOrdersMatch).
 * Packet helpers (Core/arch/SSE, not in the tree): swizzle2(a, b, mask) = (a[mask & 1], b[mask >> 1]),
 * duplane(a, p) = (a[p], a[p]).
 * Call site: fast_gicp_impl.hpp:146-150 (RCR.inverse() with RCR(3,3) = 1).
 * ============================================================================================= */
typedef struct { double v[2]; } orc_p2d;
static inline orc_p2d orc_p2(double a, double b) { orc_p2d r; r.v[0] = a; r.v[1] = b; return r; }
static inline orc_p2d orc_pmul(orc_p2d a, orc_p2d b) { return orc_p2(a.v[0] * b.v[0], a.v[1] * b.v[1]); }
static inline orc_p2d orc_padd(orc_p2d a, orc_p2d b) { return orc_p2(a.v[0] + b.v[0], a.v[1] + b.v[1]); }
static inline orc_p2d orc_psub(orc_p2d a, orc_p2d b) { return orc_p2(a.v[0] - b.v[0], a.v[1] - b.v[1]); }
static inline orc_p2d orc_swz(orc_p2d a, orc_p2d b, int mask) { return orc_p2(a.v[mask & 1], b.v[(mask >> 1) & 1]); }
static inline orc_p2d orc_dup(orc_p2d a, int p) { return orc_p2(a.v[p], a.v[p]); }
static inline void orc_eig_inv4d(const double M[16] /* row-major */, double R[16] /* row-major */) {
  double cm[16], res[16];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) cm[j * 4 + i] = M[i * 4 + j];   /* column-major data() */
  /* :199-206: A1 = data[0..1] ... -- the names follow the file; with column-major data "rows" are columns */
  orc_p2d A1 = orc_p2(cm[0], cm[1]), B1 = orc_p2(cm[2], cm[3]), A2 = orc_p2(cm[4], cm[5]), B2 = orc_p2(cm[6], cm[7]);
  orc_p2d C1 = orc_p2(cm[8], cm[9]), D1 = orc_p2(cm[10], cm[11]), C2 = orc_p2(cm[12], cm[13]), D2 = orc_p2(cm[14], cm[15]);
  orc_p2d dA, dB, dC, dD;   /* :236-253 */
  dA = orc_swz(A2, A2, 1); dA = orc_pmul(A1, dA); dA = orc_psub(dA, orc_dup(dA, 1));
  dB = orc_swz(B2, B2, 1); dB = orc_pmul(B1, dB); dB = orc_psub(dB, orc_dup(dB, 1));
  dC = orc_swz(C2, C2, 1); dC = orc_pmul(C1, dC); dC = orc_psub(dC, orc_dup(dC, 1));
  dD = orc_swz(D2, D2, 1); dD = orc_pmul(D1, dD); dD = orc_psub(dD, orc_dup(dD, 1));
  orc_p2d DC1, DC2, AB1, AB2;   /* :257-268 */
  AB1 = orc_pmul(B1, orc_dup(A2, 1));
  AB2 = orc_pmul(B2, orc_dup(A1, 0));
  AB1 = orc_psub(AB1, orc_pmul(B2, orc_dup(A1, 1)));
  AB2 = orc_psub(AB2, orc_pmul(B1, orc_dup(A2, 0)));
  DC1 = orc_pmul(C1, orc_dup(D2, 1));
  DC2 = orc_pmul(C2, orc_dup(D1, 0));
  DC1 = orc_psub(DC1, orc_pmul(C2, orc_dup(D1, 1)));
  DC2 = orc_psub(DC2, orc_pmul(C1, orc_dup(D2, 0)));
  orc_p2d d1, d2, det, rd;   /* :270-290 */
  d1 = orc_pmul(AB1, orc_swz(DC1, DC2, 0));
  d2 = orc_pmul(AB2, orc_swz(DC1, DC2, 3));
  rd = orc_padd(d1, d2);
  rd = orc_padd(rd, orc_dup(rd, 1));
  d1 = orc_pmul(dA, dD);
  d2 = orc_pmul(dB, dC);
  det = orc_padd(d1, d2);
  det = orc_psub(det, rd);
  det = orc_dup(det, 0);
  rd = orc_p2(1.0 / det.v[0], 1.0 / det.v[1]);
  orc_p2d iA1, iA2, iB1, iB2, iC1, iC2, iD1, iD2;   /* :292-330 */
  iD1 = orc_pmul(AB1, orc_dup(C1, 0));
  iD2 = orc_pmul(AB1, orc_dup(C2, 0));
  iD1 = orc_padd(iD1, orc_pmul(AB2, orc_dup(C1, 1)));
  iD2 = orc_padd(iD2, orc_pmul(AB2, orc_dup(C2, 1)));
  dA = orc_dup(dA, 0);
  iD1 = orc_psub(orc_pmul(D1, dA), iD1);
  iD2 = orc_psub(orc_pmul(D2, dA), iD2);
  iA1 = orc_pmul(DC1, orc_dup(B1, 0));
  iA2 = orc_pmul(DC1, orc_dup(B2, 0));
  iA1 = orc_padd(iA1, orc_pmul(DC2, orc_dup(B1, 1)));
  iA2 = orc_padd(iA2, orc_pmul(DC2, orc_dup(B2, 1)));
  dD = orc_dup(dD, 0);
  iA1 = orc_psub(orc_pmul(A1, dD), iA1);
  iA2 = orc_psub(orc_pmul(A2, dD), iA2);
  iB1 = orc_pmul(D1, orc_swz(AB2, AB1, 1));
  iB2 = orc_pmul(D2, orc_swz(AB2, AB1, 1));
  iB1 = orc_psub(iB1, orc_pmul(orc_swz(D1, D1, 1), orc_swz(AB2, AB1, 2)));
  iB2 = orc_psub(iB2, orc_pmul(orc_swz(D2, D2, 1), orc_swz(AB2, AB1, 2)));
  dB = orc_dup(dB, 0);
  iB1 = orc_psub(orc_pmul(C1, dB), iB1);
  iB2 = orc_psub(orc_pmul(C2, dB), iB2);
  iC1 = orc_pmul(A1, orc_swz(DC2, DC1, 1));
  iC2 = orc_pmul(A2, orc_swz(DC2, DC1, 1));
  iC1 = orc_psub(iC1, orc_pmul(orc_swz(A1, A1, 1), orc_swz(DC2, DC1, 2)));
  iC2 = orc_psub(iC2, orc_pmul(orc_swz(A2, A2, 1), orc_swz(DC2, DC1, 2)));
  dC = orc_dup(dC, 0);
  iC1 = orc_psub(orc_pmul(B1, dC), iC1);
  iC2 = orc_psub(orc_pmul(B2, dC), iC2);
  d1 = orc_p2(rd.v[0], -rd.v[1]);   /* :332-337 pxor with the sign masks (+, -) and (-, +) */
  d2 = orc_p2(-rd.v[0], rd.v[1]);
  orc_p2d o;                         /* :339-348, res_stride = 4 */
#define ORC_ST(off, val) do { o = (val); res[(off)] = o.v[0]; res[(off) + 1] = o.v[1]; } while (0)
  ORC_ST(0, orc_pmul(orc_swz(iA2, iA1, 3), d1));
  ORC_ST(4, orc_pmul(orc_swz(iA2, iA1, 0), d2));
  ORC_ST(2, orc_pmul(orc_swz(iB2, iB1, 3), d1));
  ORC_ST(6, orc_pmul(orc_swz(iB2, iB1, 0), d2));
  ORC_ST(8, orc_pmul(orc_swz(iC2, iC1, 3), d1));
  ORC_ST(12, orc_pmul(orc_swz(iC2, iC1, 0), d2));
  ORC_ST(10, orc_pmul(orc_swz(iD2, iD1, 3), d1));
  ORC_ST(14, orc_pmul(orc_swz(iD2, iD1, 0), d2));
#undef ORC_ST
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) R[i * 4 + j] = res[j * 4 + i];
}

#endif /* ORC_EIGEN_H */

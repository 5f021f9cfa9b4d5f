/*
 * oracle/orc_pclndt.c -- pclomp::NormalDistributionsTransform of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Restates (paths relative to /root/reference/src/pointcloud_match/ndt_omp/include/pclomp):
 *   VoxelGridCovariance::applyFilter ........ voxel_grid_covariance_omp_impl.hpp:48-370
 *   getNeighborhoodAtPoint{,7,1} ............ voxel_grid_covariance_omp_impl.hpp:373-442
 *   computeTransformation ................... ndt_omp_impl.hpp:69-156
 *   computeDerivatives / updateDerivatives .. ndt_omp_impl.hpp:168-267, 451-495   (float inner products)
 *   computeAngleDerivatives ................. ndt_omp_impl.hpp:270-366
 *   computePointDerivatives (float, double) . ndt_omp_impl.hpp:369-448
 *   computeHessian / updateHessian .......... ndt_omp_impl.hpp:498-590            (double)
 *   updateIntervalMT / trialValueSelectionMT  ndt_omp_impl.hpp:593-690
 *   computeStepLengthMT ..................... ndt_omp_impl.hpp:693-833
 * Third-party pieces that are not in the tree, restated from their published algorithms:
 *   Eigen::JacobiSVD<6x6>::solve   -> one-sided Jacobi SVD, rank threshold 6 eps sigma_max (SVDBase::threshold)
 *   Eigen::SelfAdjointEigenSolver  -> cyclic Jacobi (orc_linalg.h)
 *   Matrix3f::eulerAngles(0,1,2)   -> Eigen/src/Geometry/EulerAngles.h; Transform::rotation() (an SVD polar
 *                                     decomposition in Eigen) is taken as the guess's linear part itself
 *   pcl::transformPointCloud       -> x' = m00 x + (m01 y + (m02 z + m03)) per row, float (PCL >= 1.10 SSE order)
 * Floating-point order inside the float products is the plain left-to-right sum over k.
 */
#include "orc_internal.h"

#include <float.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct orc_leaf { double mean[3], cov[9], icov[9], evals[3]; double sum[3], sxx[9]; float csum[3], centroid[3]; int n, in_centroids; } orc_leaf;

typedef struct orc_pclndt_state {
  orc_vhash h;
  orc_leaf *leaf;
  long nleaf;
  int valid;
  float leaf_size, inv_leaf;
  /* computeAngleDerivatives products */
  double j_ang_d[8][3], h_ang_d[15][3];
  float j_ang[8][4], h_ang[16][4];
  double gauss_d1, gauss_d2, gauss_d3;
  float final_T[16];   /* final_transformation_ (row-major) */
  int n_deriv, n_hess;
} orc_pclndt_state;

static orc_pclndt_state *ps(oracle *o) {
  if (!o->pclndt) o->pclndt = calloc(1, sizeof(orc_pclndt_state));
  return (orc_pclndt_state *)o->pclndt;
}

void orc_pclndt_invalidate(oracle *o) { if (o->pclndt) ((orc_pclndt_state *)o->pclndt)->valid = 0; }

void orc_pclndt_free(oracle *o) {
  if (!o->pclndt) return;
  orc_pclndt_state *s = (orc_pclndt_state *)o->pclndt;
  orc_vhash_free(&s->h);
  free(s->leaf);
  free(s);
  o->pclndt = NULL;
}

/* ---- VoxelGridCovariance::applyFilter  voxel_grid_covariance_omp_impl.hpp:206-368 ------------- */
static void build_leaves(oracle *o) {
  orc_pclndt_state *s = ps(o);
  if (s->valid && s->leaf_size == (float)o->cfg.voxel_resolution) return;
  orc_vhash_free(&s->h);
  free(s->leaf);
  s->leaf_size = (float)o->cfg.voxel_resolution;
  s->inv_leaf = 1.0f / s->leaf_size;                       /* inverse_leaf_size_ = 1 / leaf_size_ (float) */
  orc_vhash_init(&s->h, o->tgt.n);
  s->leaf = (orc_leaf *)calloc((size_t)o->tgt.n + 1, sizeof(orc_leaf));
  for (long i = 0; i < o->tgt.n; i++) {                    /* first pass  :206-259 */
    const float *p = o->tgt.xyz + 3 * i;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    const int c0 = (int)floorf(p[0] * s->inv_leaf), c1 = (int)floorf(p[1] * s->inv_leaf), c2 = (int)floorf(p[2] * s->inv_leaf);
    orc_leaf *l = &s->leaf[orc_vhash_insert(&s->h, c0, c1, c2)];
    /* Leaf(): cov_ starts as the IDENTITY (voxel_grid_covariance_omp.h:103-110) and the first pass adds x x^T onto it (:236) */
    if (l->n == 0) l->sxx[0] = l->sxx[4] = l->sxx[8] = 1.0;
    const double x[3] = {p[0], p[1], p[2]};
    for (int a = 0; a < 3; a++) { l->sum[a] += x[a]; for (int b = 0; b < 3; b++) l->sxx[a * 3 + b] += x[a] * x[b]; }
    for (int a = 0; a < 3; a++) l->csum[a] += p[a];          /* leaf.centroid += pt (Vector4f, float)  :241-242 */
    l->n++;
  }
  s->nleaf = s->h.count;
  const int min_points = 6;                                /* voxel_grid_covariance_omp.h:210 */
  const double min_covar_eigvalue_mult = 0.01;             /* :211 */
  for (long v = 0; v < s->nleaf; v++) {                    /* second pass  :262-366 */
    orc_leaf *l = &s->leaf[v];
    for (int a = 0; a < 3; a++) l->centroid[a] = l->csum[a] / (float)l->n;     /* :275 */
    for (int a = 0; a < 3; a++) l->mean[a] = l->sum[a] / l->n;
    if (l->n < min_points) continue;
    l->in_centroids = 1;                                      /* pushed to the centroid cloud before the eigen checks  :288-318 */
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
      l->cov[a * 3 + b] = (l->sxx[a * 3 + b] - 2 * (l->sum[a] * l->mean[b])) / l->n + l->mean[a] * l->mean[b];      /* :323 */
    for (int a = 0; a < 9; a++) l->cov[a] *= (l->n - 1.0) / l->n;                                                 /* :324 */
    double w[3], V[9], sym[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sym[a * 3 + b] = l->cov[(a > b ? a : b) * 3 + (a > b ? b : a)];   /* self-adjoint view: lower triangle */
    orc_eig_selfadjoint3(sym, w, V);   /* eigensolver.compute(leaf.cov_)  :327 -> Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h:412-462 (orc_eigen.h) */
    if (w[0] < 0 || w[1] < 0 || w[2] <= 0) { l->n = -1; continue; }                                              /* :331-335 */
    const double min_ev = min_covar_eigvalue_mult * w[2];
    if (w[0] < min_ev) {                                                                                          /* :339-349 */
      w[0] = min_ev;
      if (w[1] < min_ev) w[1] = min_ev;
      double Vi[9], VW[9];
      orc_inv3d(V, Vi);
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) VW[a * 3 + b] = V[a * 3 + b] * w[b];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) l->cov[a * 3 + b] = VW[a * 3 + 0] * Vi[0 * 3 + b] + VW[a * 3 + 1] * Vi[1 * 3 + b] + VW[a * 3 + 2] * Vi[2 * 3 + b];
    }
    memcpy(l->evals, w, sizeof(w));
    orc_inv3d(l->cov, l->icov);
    double mx = -DBL_MAX, mn = DBL_MAX;
    for (int a = 0; a < 9; a++) { if (l->icov[a] > mx) mx = l->icov[a]; if (l->icov[a] < mn) mn = l->icov[a]; }
    if (mx == (double)INFINITY || mn == -(double)INFINITY) l->n = -1;                                             /* :353-357 */
  }
  s->valid = 1;
}

static int neighbor_offsets(int nn, int out[27][3]) {
  if (nn == 0) return 0;   /* KDTREE */
  if (nn == 1) { out[0][0] = out[0][1] = out[0][2] = 0; return 1; }
  if (nn == 7) {   /* getNeighborhoodAtPoint7  :414-428 */
    static const int o7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    memcpy(out, o7, sizeof(o7));
    return 7;
  }
  int t = 0;       /* DIRECT26: pcl::getAllNeighborCellIndices (all 27 cells of the 3x3x3 block) */
  for (int i = -1; i <= 1; i++) for (int j = -1; j <= 1; j++) for (int k = -1; k <= 1; k++) { out[t][0] = i; out[t][1] = j; out[t][2] = k; t++; }
  return 27;
}

/* leaves around a transformed point  :373-405 (the bounding-box test only guards the linear index).
 * nO == 0: KDTREE -- radiusSearch(point, resolution) over the centroid cloud (voxel_grid_covariance_omp.h:476-505): every leaf
 * that entered the centroid cloud (>= 6 points, even if its covariance was rejected afterwards) whose float centroid lies
 * strictly within the radius (FLANN RadiusResultSet: dist < radius^2, L2_Simple in float).  A centroid lies inside its own
 * cell, so the 27 cells around the query hold every candidate. */
static int neighborhood(const orc_pclndt_state *s, const float xt[3], int nO, int offs[27][3], const orc_leaf **out) {
  const int c0 = (int)floorf(xt[0] / s->leaf_size), c1 = (int)floorf(xt[1] / s->leaf_size), c2 = (int)floorf(xt[2] / s->leaf_size);
  int m = 0;
  if (nO == 0) {
    const float r2 = (float)((double)s->leaf_size * (double)s->leaf_size);
    for (int i = -1; i <= 1; i++) for (int j = -1; j <= 1; j++) for (int k = -1; k <= 1; k++) {
      const int v = orc_vhash_find(&s->h, c0 + i, c1 + j, c2 + k);
      if (v < 0 || !s->leaf[v].in_centroids) continue;
      const float *c = s->leaf[v].centroid;
      float d2 = 0.0f;
      for (int a = 0; a < 3; a++) { const float df = xt[a] - c[a]; d2 += df * df; }
      if (d2 < r2) out[m++] = &s->leaf[v];
    }
    return m;
  }
  for (int k = 0; k < nO; k++) {
    const int v = orc_vhash_find(&s->h, c0 + offs[k][0], c1 + offs[k][1], c2 + offs[k][2]);
    if (v >= 0 && s->leaf[v].n >= 6) out[m++] = &s->leaf[v];
  }
  return m;
}

/* ---- computeAngleDerivatives  ndt_omp_impl.hpp:270-366 --------------------------------------- */
static void angle_derivatives(orc_pclndt_state *s, const double p[6]) {
  double cx, cy, cz, sx, sy, sz;
  if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
  if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
  if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
  const double J[8][3] = {{(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},
                          {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
                          {(-sy * cz), sy * sz, cy},
                          {sx * cy * cz, (-sx * cy * sz), sx * sy},
                          {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
                          {(-cy * sz), (-cy * cz), 0},
                          {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},
                          {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
  for (int r = 0; r < 8; r++) { for (int c = 0; c < 3; c++) { s->j_ang_d[r][c] = J[r][c]; s->j_ang[r][c] = (float)J[r][c]; } s->j_ang[r][3] = 0.0f; }
  /* double vectors h_ang_a2_ .. h_ang_f3_  :317-337 */
  const double Hd[15][3] = {{(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},      /* a2 */
                            {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},   /* a3 */
                            {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                         /* b2 */
                            {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                         /* b3 */
                            {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},             /* c2 */
                            {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},             /* c3 */
                            {(-cy * cz), (cy * sz), (-sy)},                                       /* d1 */
                            {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                         /* d2 */
                            {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                        /* d3 */
                            {(sy * sz), (sy * cz), 0},                                            /* e1 */
                            {(-sx * cy * sz), (-sx * cy * cz), 0},                                /* e2 */
                            {(cx * cy * sz), (cx * cy * cz), 0},                                  /* e3 */
                            {(-cy * cz), (cy * sz), 0},                                           /* f1 */
                            {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},            /* f2 */
                            {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};           /* f3 */
  memcpy(s->h_ang_d, Hd, sizeof(Hd));
  /* float matrix h_ang  :339-364: identical except row 6 (d1), whose z entry is (sy), not (-sy) */
  for (int r = 0; r < 15; r++) { for (int c = 0; c < 3; c++) s->h_ang[r][c] = (float)Hd[r][c]; s->h_ang[r][3] = 0.0f; }
  s->h_ang[6][2] = (float)(sy);
  for (int c = 0; c < 4; c++) s->h_ang[15][c] = 0.0f;
}

/* Eigen::AngleAxis<float>::toRotationMatrix() for a unit coordinate axis (Eigen/src/Geometry/AngleAxis.h) */
static void angle_axis_matrix(float angle, int axis, float R[9]) {
  float ax[3] = {0.0f, 0.0f, 0.0f};
  ax[axis] = 1.0f;
  const float sn = sinf(angle), c = cosf(angle);
  const float sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]}, c1[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
  float tmp;
  tmp = c1[0] * ax[1]; R[0 * 3 + 1] = tmp - sa[2]; R[1 * 3 + 0] = tmp + sa[2];
  tmp = c1[0] * ax[2]; R[0 * 3 + 2] = tmp + sa[1]; R[2 * 3 + 0] = tmp - sa[1];
  tmp = c1[1] * ax[2]; R[1 * 3 + 2] = tmp - sa[0]; R[2 * 3 + 1] = tmp + sa[0];
  for (int a = 0; a < 3; a++) R[a * 3 + a] = c1[a] * ax[a] + c;
}

/* final_transformation_ = Translation(x_t[0..2]) * AngleAxis(x) * AngleAxis(y) * AngleAxis(z), all float  :723-724 */
static void pose_from_p(const double p[6], float T[16]) {
  float Rx[9], Ry[9], Rz[9], A[9], R[9];
  angle_axis_matrix((float)p[3], 0, Rx);
  angle_axis_matrix((float)p[4], 1, Ry);
  angle_axis_matrix((float)p[5], 2, Rz);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i * 3 + j] = (Rx[i * 3 + 0] * Ry[0 * 3 + j] + Rx[i * 3 + 1] * Ry[1 * 3 + j]) + Rx[i * 3 + 2] * Ry[2 * 3 + j];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = (A[i * 3 + 0] * Rz[0 * 3 + j] + A[i * 3 + 1] * Rz[1 * 3 + j]) + A[i * 3 + 2] * Rz[2 * 3 + j];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j]; T[i * 4 + 3] = (float)p[i]; }
  T[12] = T[13] = T[14] = 0.0f; T[15] = 1.0f;
}

static inline void transform_point(const float T[16], const float *p, float out[3]) {   /* pcl::transformPointCloud, float */
  for (int a = 0; a < 3; a++) out[a] = T[a * 4 + 0] * p[0] + (T[a * 4 + 1] * p[1] + (T[a * 4 + 2] * p[2] + T[a * 4 + 3]));
}

/* ---- computeDerivatives  :168-267 with updateDerivatives :451-495 ---------------------------- */
static double derivatives(oracle *o, const float T[16], const double p[6], int compute_hessian, double g[6], double H[36]) {
  orc_pclndt_state *s = ps(o);
  angle_derivatives(s, p);
  s->n_deriv++;
  int offs[27][3];
  const int nO = neighbor_offsets(o->cfg.num_neighbors, offs);
  const float gauss_d2 = (float)s->gauss_d2;
#ifdef _OPENMP
  const int nth = o->cfg.num_threads > 0 ? o->cfg.num_threads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
  double *acc = (double *)calloc((size_t)nth * 44, sizeof(double));
#pragma omp parallel num_threads(nth)
  {
#ifdef _OPENMP
    double *A = acc + (size_t)omp_get_thread_num() * 44;
#else
    double *A = acc;
#endif
#pragma omp for schedule(dynamic, 64)
    for (long idx = 0; idx < o->src.n; idx++) {
      const float *xp = o->src.xyz + 3 * idx;
      float xt[3];
      transform_point(T, xp, xt);
      const orc_leaf *nb[27];
      const int m = neighborhood(s, xt, nO, offs, nb);
      if (!m) continue;
      /* computePointDerivatives (float)  :369-412 */
      const float x4[4] = {xp[0], xp[1], xp[2], 0.0f};
      float xj[8], xh[16];
      for (int r = 0; r < 8; r++) xj[r] = ((s->j_ang[r][0] * x4[0] + s->j_ang[r][1] * x4[1]) + s->j_ang[r][2] * x4[2]) + s->j_ang[r][3] * x4[3];
      for (int r = 0; r < 16; r++) xh[r] = ((s->h_ang[r][0] * x4[0] + s->h_ang[r][1] * x4[1]) + s->h_ang[r][2] * x4[2]) + s->h_ang[r][3] * x4[3];
      float pg[4][6];
      memset(pg, 0, sizeof(pg));
      pg[0][0] = pg[1][1] = pg[2][2] = 1.0f;
      pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
      float ph[24][6];
      memset(ph, 0, sizeof(ph));
      const float va[4] = {0, xh[0], xh[1], 0}, vb[4] = {0, xh[2], xh[3], 0}, vc[4] = {0, xh[4], xh[5], 0}, vd[4] = {xh[6], xh[7], xh[8], 0}, ve[4] = {xh[9], xh[10], xh[11], 0},
                  vf[4] = {xh[12], xh[13], xh[14], 0};
      for (int r = 0; r < 4; r++) {
        ph[12 + r][3] = va[r]; ph[16 + r][3] = vb[r]; ph[20 + r][3] = vc[r];
        ph[12 + r][4] = vb[r]; ph[16 + r][4] = vd[r]; ph[20 + r][4] = ve[r];
        ph[12 + r][5] = vc[r]; ph[16 + r][5] = ve[r]; ph[20 + r][5] = vf[r];
      }
      double score_pt = 0.0, g_pt[6] = {0, 0, 0, 0, 0, 0}, H_pt[36];
      memset(H_pt, 0, sizeof(H_pt));
      for (int c = 0; c < m; c++) {
        const orc_leaf *l = nb[c];
        /* x_trans = (double)x_trans_pt - mean; then cast to float  :239-243,453 */
        float xt4[4];
        for (int a = 0; a < 3; a++) xt4[a] = (float)((double)xt[a] - l->mean[a]);
        xt4[3] = 0.0f;
        float ci[4][4];
        memset(ci, 0, sizeof(ci));
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) ci[a][b] = (float)l->icov[a * 3 + b];
        float xc[4];   /* x_trans4 * c_inv4 */
        for (int b = 0; b < 4; b++) xc[b] = ((xt4[0] * ci[0][b] + xt4[1] * ci[1][b]) + xt4[2] * ci[2][b]) + xt4[3] * ci[3][b];
        const float q = ((xt4[0] * xc[0] + xt4[1] * xc[1]) + xt4[2] * xc[2]) + xt4[3] * xc[3];
        float e = expf(-gauss_d2 * q * 0.5f);
        const float score_inc = (float)(-s->gauss_d1 * (double)e);
        e = gauss_d2 * e;
        if (e > 1 || e < 0 || e != e) continue;                /* return (0): the score is not counted either */
        e = (float)((double)e * s->gauss_d1);
        score_pt += (double)score_inc;
        float cg[4][6];   /* c_inv4 * point_gradient4 */
        for (int a = 0; a < 4; a++) for (int j = 0; j < 6; j++) cg[a][j] = ((ci[a][0] * pg[0][j] + ci[a][1] * pg[1][j]) + ci[a][2] * pg[2][j]) + ci[a][3] * pg[3][j];
        float xg[6];      /* x_trans4 * (c_inv4 * point_gradient4) */
        for (int j = 0; j < 6; j++) xg[j] = ((xt4[0] * cg[0][j] + xt4[1] * cg[1][j]) + xt4[2] * cg[2][j]) + xt4[3] * cg[3][j];
        for (int j = 0; j < 6; j++) g_pt[j] += (double)(e * xg[j]);
        if (!compute_hessian) continue;
        float gg[6][6];   /* point_gradient4^T * (c_inv4 * point_gradient4) */
        for (int a = 0; a < 6; a++) for (int j = 0; j < 6; j++) gg[a][j] = ((pg[0][a] * cg[0][j] + pg[1][a] * cg[1][j]) + pg[2][a] * cg[2][j]) + pg[3][a] * cg[3][j];
        for (int i = 0; i < 6; i++) {
          float xh6[6];   /* x_trans4_x_c_inv4 * point_hessian_.block<4,6>(4 i, 0) */
          for (int j = 0; j < 6; j++) xh6[j] = ((xc[0] * ph[4 * i + 0][j] + xc[1] * ph[4 * i + 1][j]) + xc[2] * ph[4 * i + 2][j]) + xc[3] * ph[4 * i + 3][j];
          for (int j = 0; j < 6; j++) H_pt[i * 6 + j] += (double)(e * ((-gauss_d2 * xg[i] * xg[j] + xh6[j]) + gg[j][i]));
        }
      }
      A[42] += score_pt;
      for (int j = 0; j < 6; j++) A[36 + j] += g_pt[j];
      for (int j = 0; j < 36; j++) A[j] += H_pt[j];
    }
  }
  double score = 0.0;
  memset(g, 0, 6 * sizeof(double));
  memset(H, 0, 36 * sizeof(double));
  for (int k = 0; k < nth; k++) {
    const double *A = acc + (size_t)k * 44;
    score += A[42];
    for (int j = 0; j < 6; j++) g[j] += A[36 + j];
    for (int j = 0; j < 36; j++) H[j] += A[j];
  }
  free(acc);
  return score;
}

/* ---- computeHessian / updateHessian  :498-590 (double; the angle tables of the last computeDerivatives) ---- */
static void hessian_only(oracle *o, const float T[16], double H[36]) {
  orc_pclndt_state *s = ps(o);
  s->n_hess++;
  int offs[27][3];
  const int nO = neighbor_offsets(o->cfg.num_neighbors, offs);
  memset(H, 0, 36 * sizeof(double));
  for (long idx = 0; idx < o->src.n; idx++) {   /* serial in the reference */
    const float *xp = o->src.xyz + 3 * idx;
    float xt[3];
    transform_point(T, xp, xt);
    const orc_leaf *nb[27];
    const int m = neighborhood(s, xt, nO, offs, nb);
    if (!m) continue;
    const double x[3] = {xp[0], xp[1], xp[2]};
    double pg[3][6], ph[18][6];
    memset(pg, 0, sizeof(pg));
    memset(ph, 0, sizeof(ph));
    pg[0][0] = pg[1][1] = pg[2][2] = 1.0;
#define DOT3(v) ((x[0] * (v)[0] + x[1] * (v)[1]) + x[2] * (v)[2])
    pg[1][3] = DOT3(s->j_ang_d[0]); pg[2][3] = DOT3(s->j_ang_d[1]); pg[0][4] = DOT3(s->j_ang_d[2]); pg[1][4] = DOT3(s->j_ang_d[3]);
    pg[2][4] = DOT3(s->j_ang_d[4]); pg[0][5] = DOT3(s->j_ang_d[5]); pg[1][5] = DOT3(s->j_ang_d[6]); pg[2][5] = DOT3(s->j_ang_d[7]);
    const double va[3] = {0, DOT3(s->h_ang_d[0]), DOT3(s->h_ang_d[1])}, vb[3] = {0, DOT3(s->h_ang_d[2]), DOT3(s->h_ang_d[3])}, vc[3] = {0, DOT3(s->h_ang_d[4]), DOT3(s->h_ang_d[5])},
                 vd[3] = {DOT3(s->h_ang_d[6]), DOT3(s->h_ang_d[7]), DOT3(s->h_ang_d[8])}, ve[3] = {DOT3(s->h_ang_d[9]), DOT3(s->h_ang_d[10]), DOT3(s->h_ang_d[11])},
                 vf[3] = {DOT3(s->h_ang_d[12]), DOT3(s->h_ang_d[13]), DOT3(s->h_ang_d[14])};
#undef DOT3
    for (int r = 0; r < 3; r++) {
      ph[9 + r][3] = va[r]; ph[12 + r][3] = vb[r]; ph[15 + r][3] = vc[r];
      ph[9 + r][4] = vb[r]; ph[12 + r][4] = vd[r]; ph[15 + r][4] = ve[r];
      ph[9 + r][5] = vc[r]; ph[12 + r][5] = ve[r]; ph[15 + r][5] = vf[r];
    }
    for (int c = 0; c < m; c++) {
      const orc_leaf *l = nb[c];
      double xt3[3], cx[3];
      for (int a = 0; a < 3; a++) xt3[a] = (double)xt[a] - l->mean[a];
      for (int a = 0; a < 3; a++) cx[a] = (l->icov[a * 3 + 0] * xt3[0] + l->icov[a * 3 + 1] * xt3[1]) + l->icov[a * 3 + 2] * xt3[2];
      double e = s->gauss_d2 * exp(-s->gauss_d2 * ((xt3[0] * cx[0] + xt3[1] * cx[1]) + xt3[2] * cx[2]) / 2);
      if (e > 1 || e < 0 || e != e) continue;
      e *= s->gauss_d1;
      double cg[3][6], xg[6];   /* c_inv * point_gradient_.col(j); x_trans . that */
      for (int j = 0; j < 6; j++) {
        for (int a = 0; a < 3; a++) cg[a][j] = (l->icov[a * 3 + 0] * pg[0][j] + l->icov[a * 3 + 1] * pg[1][j]) + l->icov[a * 3 + 2] * pg[2][j];
        xg[j] = (xt3[0] * cg[0][j] + xt3[1] * cg[1][j]) + xt3[2] * cg[2][j];
      }
      for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) {
          double chx[3];
          for (int a = 0; a < 3; a++) chx[a] = (l->icov[a * 3 + 0] * ph[3 * i + 0][j] + l->icov[a * 3 + 1] * ph[3 * i + 1][j]) + l->icov[a * 3 + 2] * ph[3 * i + 2][j];
          const double t2 = (xt3[0] * chx[0] + xt3[1] * chx[1]) + xt3[2] * chx[2];
          const double t3 = (pg[0][j] * cg[0][i] + pg[1][j] * cg[1][i]) + pg[2][j] * cg[2][i];
          H[i * 6 + j] += e * ((-s->gauss_d2 * xg[i] * xg[j] + t2) + t3);
        }
      }
    }
  }
}

/* ---- Eigen::JacobiSVD<Matrix6d>(H, FullU | FullV).solve(b)  :112-114: two-sided Jacobi of Eigen/src/SVD/JacobiSVD.h
 *      and SVDBase::_solve_impl, restated in orc_eigen.h ---------------------------------------- */
static void svd_solve6(const double Hin[36], const double b[6], double x[6]) { orc_eig_svd_solve6(Hin, b, x); }

/* ---- More-Thuente  :593-690 ------------------------------------------------------------------- */
static int update_interval(double *a_l, double *f_l, double *g_l, double *a_u, double *f_u, double *g_u, double a_t, double f_t, double g_t) {
  if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
  else if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  else if (g_t * (*a_l - a_t) < 0) { *a_u = *a_l; *f_u = *f_l; *g_u = *g_l; *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  return 1;
}

static double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_t_next = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    if (a_t > a_l) return fmin(a_t + 0.66 * (a_u - a_t), a_t_next);
    return fmax(a_t + 0.66 * (a_u - a_t), a_t_next);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  const double w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

/* computeStepLengthMT  :693-833; psi / dpsi: ndt_omp.h auxiliaryFunction_PsiMT / dPsiMT */
static double step_length_mt(oracle *o, const double x[6], double dir[6], double step_init, double step_max, double step_min, double *score, double g[6], double H[36]) {
  orc_pclndt_state *s = ps(o);
  const double phi_0 = -*score;
  double d_phi_0 = -(((((g[0] * dir[0] + g[1] * dir[1]) + g[2] * dir[2]) + g[3] * dir[3]) + g[4] * dir[4]) + g[5] * dir[5]);
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;
    d_phi_0 *= -1;
    for (int i = 0; i < 6; i++) dir[i] *= -1;
  }
  const int max_step_iterations = 10;
  int step_iterations = 0;
  const double mu = 1.e-4, nu = 0.9;
  double a_l = 0, a_u = 0;
#define PSI(a, f_a, f_0, g_0) ((f_a) - (f_0) - mu * (g_0) * (a))
#define DPSI(g_a, g_0) ((g_a) - mu * (g_0))
  double f_l = PSI(a_l, phi_0, phi_0, d_phi_0), g_l = DPSI(d_phi_0, d_phi_0);
  double f_u = PSI(a_u, phi_0, phi_0, d_phi_0), g_u = DPSI(d_phi_0, d_phi_0);
  int interval_converged = (step_max - step_min) < 0, open_interval = 1;
  double a_t = step_init;
  a_t = fmin(a_t, step_max);
  a_t = fmax(a_t, step_min);
  double x_t[6];
  for (int i = 0; i < 6; i++) x_t[i] = x[i] + dir[i] * a_t;
  pose_from_p(x_t, s->final_T);
  *score = derivatives(o, s->final_T, x_t, 1, g, H);
  double phi_t = -*score;
  double d_phi_t = -(((((g[0] * dir[0] + g[1] * dir[1]) + g[2] * dir[2]) + g[3] * dir[3]) + g[4] * dir[4]) + g[5] * dir[5]);
  double psi_t = PSI(a_t, phi_t, phi_0, d_phi_0), d_psi_t = DPSI(d_phi_t, d_phi_0);
  while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    if (open_interval) a_t = trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else a_t = trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = fmin(a_t, step_max);
    a_t = fmax(a_t, step_min);
    for (int i = 0; i < 6; i++) x_t[i] = x[i] + dir[i] * a_t;
    pose_from_p(x_t, s->final_T);
    *score = derivatives(o, s->final_T, x_t, 0, g, H);
    phi_t = -*score;
    d_phi_t = -(((((g[0] * dir[0] + g[1] * dir[1]) + g[2] * dir[2]) + g[3] * dir[3]) + g[4] * dir[4]) + g[5] * dir[5]);
    psi_t = PSI(a_t, phi_t, phi_0, d_phi_0);
    d_psi_t = DPSI(d_phi_t, d_phi_0);
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = 0;
      f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
      f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
    }
    if (open_interval) interval_converged = update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
    else interval_converged = update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
#undef PSI
#undef DPSI
  if (step_iterations) hessian_only(o, s->final_T, H);
  return a_t;
}

static void gauss_params(oracle *o) {   /* eq. 6.8  :77-82 */
  orc_pclndt_state *s = ps(o);
  const double gauss_c1 = 10 * (1 - o->cfg.ndt_outlier_ratio);
  const double gauss_c2 = o->cfg.ndt_outlier_ratio / pow((double)(float)o->cfg.voxel_resolution, 3);
  s->gauss_d3 = -log(gauss_c2);
  s->gauss_d1 = -log(gauss_c1 + gauss_c2) - s->gauss_d3;
  s->gauss_d2 = -2 * log((-log(gauss_c1 * exp(-0.5) + gauss_c2) - s->gauss_d3) / s->gauss_d1);
}

/* Matrix3f::eulerAngles(0, 1, 2)  Eigen/src/Geometry/EulerAngles.h (float) */
static void euler_012(const float R[9], float res[3]) {
  const int i = 0, j = 1, k = 2;   /* odd = 0 */
  res[0] = atan2f(R[j * 3 + k], R[k * 3 + k]);
  const float c2 = sqrtf(R[i * 3 + i] * R[i * 3 + i] + R[i * 3 + j] * R[i * 3 + j]);
  if (res[0] > 0.0f) {
    if (res[0] > 0.0f) res[0] -= (float)M_PI; else res[0] += (float)M_PI;
    res[1] = atan2f(-R[i * 3 + k], -c2);
  } else {
    res[1] = atan2f(-R[i * 3 + k], c2);
  }
  const float s1 = sinf(res[0]), c1 = cosf(res[0]);
  res[2] = atan2f(s1 * R[k * 3 + i] - c1 * R[j * 3 + i], c1 * R[j * 3 + j] - s1 * R[k * 3 + j]);
  res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];
}

/* computeTransformation  :69-156 */
int orc_pclndt_align(oracle *o, const float guess[16], orc_result *out) {
  orc_pclndt_state *s = ps(o);
  if (o->src.n <= 0 || o->tgt.n <= 0) return -1;
  build_leaves(o);
  gauss_params(o);
  s->n_deriv = s->n_hess = 0;
  int nr_iterations = 0, converged = 0;
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(s->final_T, ident, sizeof(ident));
  if (memcmp(guess, ident, sizeof(ident)) != 0) memcpy(s->final_T, guess, sizeof(ident));
  float R[9], eul[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = s->final_T[i * 4 + j];
  euler_012(R, eul);
  double p[6] = {s->final_T[3], s->final_T[7], s->final_T[11], eul[0], eul[1], eul[2]};
  double g[6], H[36], delta[6];
  double score = derivatives(o, s->final_T, p, 1, g, H);
  const double step_size = o->cfg.ndt_step_size, eps = o->cfg.translation_eps;
  while (!converged) {
    double mg[6];
    for (int i = 0; i < 6; i++) mg[i] = -g[i];
    svd_solve6(H, mg, delta);
    double nrm = 0;
    for (int i = 0; i < 6; i++) nrm += delta[i] * delta[i];
    nrm = sqrt(nrm);
    if (nrm == 0 || nrm != nrm) { converged = nrm == nrm; goto done; }   /* :117-121 */
    for (int i = 0; i < 6; i++) delta[i] /= nrm;
    nrm = step_length_mt(o, p, delta, nrm, step_size, eps / 2, &score, g, H);
    for (int i = 0; i < 6; i++) { delta[i] *= nrm; p[i] += delta[i]; }
    if (nr_iterations > o->cfg.max_iterations || (nr_iterations && (fabs(nrm) < eps))) converged = 1;
    nr_iterations++;
  }
done:
  if (out) {
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < 16; i++) { out->T[i] = s->final_T[i]; out->T64[i] = (double)s->final_T[i]; }
    memcpy(out->H, H, sizeof(out->H));
    out->cost = score;                       /* trans_probability_ * N */
    out->iterations = nr_iterations;
    out->converged = converged;
    out->num_linearize = s->n_deriv;
    out->num_compute_error = s->n_hess;
  }
  return 0;
}

/* calculateScore  ndt_omp_impl.hpp:835-880: sum over the neighbour cells of (-d1 e - d3) / #cells, divided by N; double, serial */
double orc_pclndt_score(void *h, const float T[16]) {
  oracle *o = (oracle *)h;
  orc_pclndt_state *s = ps(o);
  build_leaves(o);
  gauss_params(o);
  int offs[27][3];
  const int nO = neighbor_offsets(o->cfg.num_neighbors, offs);
  double score = 0.0;
  for (long idx = 0; idx < o->src.n; idx++) {
    float xt[3];
    transform_point(T, o->src.xyz + 3 * idx, xt);
    const orc_leaf *nb[27];
    const int m = neighborhood(s, xt, nO, offs, nb);
    for (int c = 0; c < m; c++) {
      const orc_leaf *l = nb[c];
      double x[3], cx[3];
      for (int a = 0; a < 3; a++) x[a] = (double)xt[a] - l->mean[a];
      for (int a = 0; a < 3; a++) cx[a] = (l->icov[a * 3 + 0] * x[0] + l->icov[a * 3 + 1] * x[1]) + l->icov[a * 3 + 2] * x[2];
      const double e = exp(-s->gauss_d2 * ((x[0] * cx[0] + x[1] * cx[1]) + x[2] * cx[2]) / 2);
      const double inc = -s->gauss_d1 * e - s->gauss_d3;
      score += inc / m;
    }
  }
  return score / (double)o->src.n;
}

/* ---- unit hooks ------------------------------------------------------------------------------- */
double orc_pclndt_derivatives(void *h, const double p[6], int compute_hessian, double g[6], double H[36]) {
  oracle *o = (oracle *)h;
  build_leaves(o);
  gauss_params(o);
  float T[16];
  pose_from_p(p, T);
  return derivatives(o, T, p, compute_hessian, g, H);
}

void orc_pclndt_hessian(void *h, const double p[6], double H[36]) {   /* after orc_pclndt_derivatives at the same p */
  oracle *o = (oracle *)h;
  float T[16];
  pose_from_p(p, T);
  hessian_only(o, T, H);
}

int orc_pclndt_leaf(void *h, const float pt[3], double mean[3], double icov[9], int *n) {
  oracle *o = (oracle *)h;
  orc_pclndt_state *s = ps(o);
  build_leaves(o);
  const int v = orc_vhash_find(&s->h, (int)floorf(pt[0] / s->leaf_size), (int)floorf(pt[1] / s->leaf_size), (int)floorf(pt[2] / s->leaf_size));
  if (v < 0) return 0;
  memcpy(mean, s->leaf[v].mean, 3 * sizeof(double));
  memcpy(icov, s->leaf[v].icov, 9 * sizeof(double));
  *n = s->leaf[v].n;
  return 1;
}

void orc_pclndt_pose(const double p[6], float T[16]) { pose_from_p(p, T); }
void orc_pclndt_euler(const float R[9], float e[3]) { euler_012(R, e); }
void orc_pclndt_svd_solve(const double H[36], const double b[6], double x[6]) { svd_solve6(H, b, x); }

/*
 * oracle/orc_gicp.c -- GICP and VGICP residual models of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * What is restated here, and from where (paths relative to
 * /root/reference/src/pointcloud_match/fast_gicp/include/fast_gicp/gicp):
 *   covariance estimation ....... impl/fast_gicp_impl.hpp:239-298 (kNN k=20, XX^T/k, regularisation)
 *   FastGICP correspondences ..... impl/fast_gicp_impl.hpp:114-152 (float transform, exact 1-NN, RCR^-1)
 *   FastGICP linearize / error ... impl/fast_gicp_impl.hpp:154-237
 *   VGICP voxel map .............. fast_vgicp_voxel.hpp:10-44,94-182 (ADDITIVE accumulation, floor(x/res-0.5))
 *   FastVGICP correspondences .... impl/fast_vgicp_impl.hpp:72-124
 *   FastVGICP linearize / error .. impl/fast_vgicp_impl.hpp:126-204
 * The kd-tree (pcl::search::KdTree -> FLANN, not in the tree) returns the exact Euclidean k
 * nearest neighbours; here the same set is found with a uniform grid and an exactness test per
 * ring (verified against brute force in tests/test_oracle_kat.py).  JacobiSVD of the symmetric
 * PSD covariance = its eigen-decomposition (cyclic Jacobi, orc_linalg.h).
 */
#include "orc_internal.h"

#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- uniform grid for exact kNN ------------------------------------------------------------- */
typedef struct orc_grid {
  orc_vhash h;
  int *start, *pts;
  long ncell;
  double cell;
} orc_grid;

static void grid_free(orc_grid *g) { orc_vhash_free(&g->h); free(g->start); free(g->pts); memset(g, 0, sizeof(*g)); }

static inline void grid_key(const orc_grid *g, const float p[3], int k[3]) {
  for (int a = 0; a < 3; a++) k[a] = (int)floor((double)p[a] / g->cell);
}

static void grid_build(orc_grid *g, const orc_cloud *c, double cell) {
  grid_free(g);
  g->cell = cell;
  orc_vhash_init(&g->h, c->n);
  int *pv = (int *)malloc(sizeof(int) * (size_t)(c->n + 1));
  for (long i = 0; i < c->n; i++) { int k[3]; grid_key(g, c->xyz + 3 * i, k); pv[i] = orc_vhash_insert(&g->h, k[0], k[1], k[2]); }
  g->ncell = g->h.count;
  g->start = (int *)calloc((size_t)g->ncell + 1, sizeof(int));
  g->pts = (int *)malloc(sizeof(int) * (size_t)(c->n + 1));
  for (long i = 0; i < c->n; i++) g->start[pv[i] + 1]++;
  for (long j = 0; j < g->ncell; j++) g->start[j + 1] += g->start[j];
  int *fill = (int *)malloc(sizeof(int) * ((size_t)g->ncell + 1));
  memcpy(fill, g->start, sizeof(int) * ((size_t)g->ncell + 1));
  for (long i = 0; i < c->n; i++) g->pts[fill[pv[i]]++] = (int)i;
  free(fill); free(pv);
}

/* insert point pi into the sorted top-k (distance, then index); returns the new count */
static inline int topk_insert(const orc_cloud *c, const float q[3], int pi, int k, int n, int *idx, float *d2) {
  const float *p = c->xyz + 3 * (long)pi;
  const float ex = p[0] - q[0], ey = p[1] - q[1], ez = p[2] - q[2];
  const float dd = ex * ex + ey * ey + ez * ez;
  int pos = n < k ? n : k;
  while (pos > 0 && (dd < d2[pos - 1] || (dd == d2[pos - 1] && pi < idx[pos - 1]))) pos--;
  if (pos >= k) return n;
  const int last = n < k ? n : k - 1;
  for (int t = last; t > pos; t--) { d2[t] = d2[t - 1]; idx[t] = idx[t - 1]; }
  d2[pos] = dd; idx[pos] = pi;
  return n < k ? n + 1 : n;
}

/* exact k nearest neighbours of q (float squared distances like FLANN's L2_Simple<float>);
 * out sorted ascending (distance, then index).  Returns the number found (< k only for tiny clouds). */
static int grid_knn(const orc_grid *g, const orc_cloud *c, const float q[3], int k, double max_d2, int *idx, float *d2) {
  int key[3], n = 0;
  grid_key(g, q, key);
  const int RMAX = 12;
  int exact = 0;
  for (int R = 0; R <= RMAX && !exact; R++) {
    for (int dx = -R; dx <= R; dx++) for (int dy = -R; dy <= R; dy++) for (int dz = -R; dz <= R; dz++) {
      const int m = abs(dx) > abs(dy) ? abs(dx) : abs(dy);
      if ((m > abs(dz) ? m : abs(dz)) != R) continue;   /* shell only */
      const int v = orc_vhash_find(&g->h, key[0] + dx, key[1] + dy, key[2] + dz);
      if (v < 0) continue;
      for (int s = g->start[v]; s < g->start[v + 1]; s++) n = topk_insert(c, q, g->pts[s], k, n, idx, d2);
    }
    /* every unseen point is farther than R * cell from q: exact once the k-th distance is inside, or nothing within max_d2 remains */
    const double reach = (double)R * g->cell;
    if ((n == k && (double)d2[k - 1] <= reach * reach) || reach * reach > max_d2) exact = 1;
  }
  if (!exact) {   /* sparse neighbourhood: brute force over the cloud */
    n = 0;
    for (long i = 0; i < c->n; i++) n = topk_insert(c, q, (int)i, k, n, idx, d2);
  }
  return n;
}

/* ---- per-point covariances  fast_gicp_impl.hpp:239-298 --------------------------------------- */
static void regularize(int method, const double cov[9], double out[9]) {
  if (method == ORC_REG_NONE) { memcpy(out, cov, 9 * sizeof(double)); return; }
  if (method == ORC_REG_FROBENIUS) {   /* :266-271 */
    double C[9], Ci[9], N[9], sq[9];
    memcpy(C, cov, sizeof(C));
    C[0] += 1e-3; C[4] += 1e-3; C[8] += 1e-3;
    orc_inv3d(C, Ci);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sq[b * 3 + a] = Ci[a * 3 + b] * Ci[a * 3 + b];   /* Matrix3d::norm(): fixed-size sum over the column-major coefficients */
    const double nrm = sqrt(orc_redux_fixed_d(sq, 9));
    for (int a = 0; a < 9; a++) N[a] = Ci[a] / nrm;
    orc_inv3d(N, out);
    return;
  }
  /* Eigen::JacobiSVD<Matrix3d>(cov, ComputeFullU | ComputeFullV)  :273, restated from Eigen/src/SVD/JacobiSVD.h (orc_eigen.h) */
  double U[9], S[3], V[9], val[3];
  orc_eig_jacobi_svd(3, cov, U, S, V);
  if (method == ORC_REG_PLANE) { val[0] = 1.0; val[1] = 1.0; val[2] = 1e-3; }                          /* :280 */
  else if (method == ORC_REG_MIN_EIG) { for (int k = 0; k < 3; k++) val[k] = S[k] > 1e-3 ? S[k] : 1e-3; }   /* :283 */
  else { for (int k = 0; k < 3; k++) { const double v = S[k] / S[0]; val[k] = v > 1e-3 ? v : 1e-3; } }      /* :286-287 NORMALIZED_MIN_EIG (maxCoeff = S[0]) */
  /* svd.matrixU() * values.asDiagonal() * svd.matrixV().transpose()  :292 */
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
    out[a * 3 + b] = ((U[a * 3 + 0] * val[0]) * V[b * 3 + 0] + (U[a * 3 + 1] * val[1]) * V[b * 3 + 1]) + (U[a * 3 + 2] * val[2]) * V[b * 3 + 2];
}

void orc_calc_covariances(const oracle *o, const orc_cloud *c, double *covs /* 9 per point */) {
  orc_grid g;
  memset(&g, 0, sizeof(g));
  grid_build(&g, c, o->cfg.voxel_resolution);
  const int k = o->cfg.k_correspondences;
#ifdef _OPENMP
  const int nth = o->cfg.num_threads > 0 ? o->cfg.num_threads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
#pragma omp parallel for num_threads(nth) schedule(dynamic, 64)
  for (long i = 0; i < c->n; i++) {
    int idx[64];
    float d2[64];
    const int m = grid_knn(&g, c, c->xyz + 3 * i, k, 1e300, idx, d2);
    double mean[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (o->cfg.regularization == ORC_REG_PCLOMP) {
      /* pclomp computeCovariances (ndt_omp/include/pclomp/gicp_omp_impl.hpp:48-122): raw second moments with FLOAT products
       * added to doubles (nearest neighbour first), cov = S / k - mean mean^T on the lower triangle, mirrored; then the
       * singular values are replaced by (1, 1, gicp_epsilon_ = 0.001): cov = sum_k v_k u_k u_k^T, largest first. */
      for (int j = 0; j < m; j++) {
        const float *pt = c->xyz + 3 * (long)idx[j];
        for (int a = 0; a < 3; a++) {
          mean[a] += (double)pt[a];
          for (int b = 0; b <= a; b++) cov[a * 3 + b] += (double)(pt[a] * pt[b]);
        }
      }
      for (int a = 0; a < 3; a++) mean[a] /= (double)k;
      for (int a = 0; a < 3; a++) for (int b = 0; b <= a; b++) {
        cov[a * 3 + b] /= (double)k;
        cov[a * 3 + b] -= mean[a] * mean[b];
        cov[b * 3 + a] = cov[a * 3 + b];
      }
      double U[9], S[3], *out = covs + 9 * i;
      orc_eig_jacobi_svd(3, cov, U, S, NULL);   /* JacobiSVD<Matrix3d>(cov, ComputeFullU)  gicp_omp_impl.hpp:110 */
      const double val[3] = {1.0, 1.0, 0.001};
      for (int a = 0; a < 9; a++) out[a] = 0.0;
      for (int kk = 0; kk < 3; kk++)      /* cov += v * col * col.transpose()  :114-120 */
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) out[a * 3 + b] += (val[kk] * U[a * 3 + kk]) * U[b * 3 + kk];
      continue;
    }
    for (int j = 0; j < m; j++) for (int a = 0; a < 3; a++) mean[a] += (double)c->xyz[3 * (long)idx[j] + a];
    for (int a = 0; a < 3; a++) mean[a] /= (double)k;              /* rowwise().mean() over k columns */
    for (int j = 0; j < m; j++) {
      double d[3];
      for (int a = 0; a < 3; a++) d[a] = (double)c->xyz[3 * (long)idx[j] + a] - mean[a];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov[a * 3 + b] += d[a] * d[b];
    }
    for (int a = 0; a < 9; a++) cov[a] /= (double)k;
    regularize(o->cfg.regularization, cov, covs + 9 * i);
  }
  grid_free(&g);
}

/* ---- per-point covariances as the CUDA core computes them (float) ------------------------------
 * covariance_estimation.cu:16-42 (sum pt, sum pt pt^T over the k neighbours, cov = S / k - mean mean^T)
 * covariance_regularization.cu:14-121 (PLANE: V diag(1e-3, 1, 1) V^-1 with the eigenvectors of the self-adjoint solver,
 * MIN_EIG, FROBENIUS; the other methods are "unimplemented" there and leave the covariance as it is).
 * Neighbours: the CPU kd-tree of FastVGICPCuda (fast_vgicp_cuda_impl.hpp:97-101,152-170), k nearest incl. the point itself.
 * computeDirect is the closed-form float solver of Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h:577-733 (orc_eigen.h). */
static void regularize_f(int method, float c[9]) {
  if (method == ORC_REG_FROBENIUS) {
    float C[9], Ci[9], N[9];
    memcpy(C, c, sizeof(C));
    C[0] += 1e-3f; C[4] += 1e-3f; C[8] += 1e-3f;
    orc_inv3f(C, Ci);
    float nrm = 0.0f;
    for (int a = 0; a < 9; a++) nrm += Ci[a] * Ci[a];
    nrm = sqrtf(nrm);
    for (int a = 0; a < 9; a++) N[a] = Ci[a] / nrm;
    orc_inv3f(N, c);
    return;
  }
  if (method != ORC_REG_PLANE && method != ORC_REG_MIN_EIG) return;
  /* SelfAdjointEigenSolver<Matrix3f>::computeDirect  covariance_regularization.cu:57-58,84-85 (orc_eigen.h) */
  float w[3], Vf[9], Vi[9], val[3], VD[9];
  orc_eig_direct3f(c, w, Vf);
  if (method == ORC_REG_PLANE) { val[0] = 1e-3f; val[1] = 1.0f; val[2] = 1.0f; }      /* :62 */
  else for (int k = 0; k < 3; k++) val[k] = fmaxf(1e-3f, w[k]);                        /* :88-90 */
  orc_inv3f(Vf, Vi);                                                                    /* eigenvectors().inverse() */
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) VD[a * 3 + b] = Vf[a * 3 + b] * val[b];
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) c[a * 3 + b] = (VD[a * 3 + 0] * Vi[0 * 3 + b] + VD[a * 3 + 1] * Vi[1 * 3 + b]) + VD[a * 3 + 2] * Vi[2 * 3 + b];
}

/* covariance_estimation_rbf  (/root/reference/src/pointcloud_match/fast_gicp/src/fast_gicp/cuda/covariance_estimation_rbf.cu:59-151):
 * every point sums over the whole cloud, in blocks of 512 points of the input order (the last block padded with points at the
 * origin, :126-129), w = expf(-kernel_width * |x - p|^2) for |x - p|^2 <= max_dist^2 (:76-82), one float accumulator per block
 * (:41-45), the blocks added in order (:108-111), finalize (:47-53), covariance_regularization (fast_vgicp_cuda.cu:210,218).
 * Plain O(N^2) as the reference runs it.  Not bit-comparable with the real reference (CUDA's expf and nvcc's fused multiply-adds
 * are not restated): "parity unpinned"; pinned here against a float64 statement of the same sums (tests/test_gpu_rbf.py). */
static void rbf_covariances_f(const oracle *o, const orc_cloud *c, float *covs) {
  const float exp_factor = (float)o->cfg.rbf_kernel_width, max_dist = (float)o->cfg.rbf_max_dist;   /* thrust::device_vector<float> constants  :118-121 */
  const float max_dist_sq = max_dist * max_dist;
  const long n = c->n, nblocks = (n + 511) / 512;
#ifdef _OPENMP
  const int nth = o->cfg.num_threads > 0 ? o->cfg.num_threads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
#pragma omp parallel for num_threads(nth) schedule(dynamic, 16)
  for (long i = 0; i < n; i++) {
    const float *x = c->xyz + 3 * i;
    float sw = 0.f, sm[3] = {0, 0, 0}, sc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long b = 0; b < nblocks; b++) {
      float w0 = 0.f, m0[3] = {0, 0, 0}, c0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (long j = 0; j < 512; j++) {
        const long k = b * 512 + j;
        float pj[3] = {0.f, 0.f, 0.f};
        if (k < n) { pj[0] = c->xyz[3 * k]; pj[1] = c->xyz[3 * k + 1]; pj[2] = c->xyz[3 * k + 2]; }
        const float dx = x[0] - pj[0], dy = x[1] - pj[1], dz = x[2] - pj[2];
        const float sq_d = (dx * dx + dy * dy) + dz * dz;
        if (sq_d > max_dist_sq) continue;
        const float w = expf(-exp_factor * sq_d);
        w0 += w;
        for (int a = 0; a < 3; a++) {
          const float wp = w * pj[a];
          m0[a] += wp;
          for (int q = 0; q < 3; q++) c0[a * 3 + q] += wp * pj[q];
        }
      }
      sw += w0;
      for (int a = 0; a < 3; a++) sm[a] += m0[a];
      for (int a = 0; a < 9; a++) sc[a] += c0[a];
    }
    float mean[3], cov[9];
    for (int a = 0; a < 3; a++) mean[a] = sm[a] / sw;
    for (int a = 0; a < 3; a++) for (int q = 0; q < 3; q++) cov[a * 3 + q] = (sc[a * 3 + q] - mean[a] * sm[q]) / sw;
    regularize_f(o->cfg.regularization, cov);
    memcpy(covs + 9 * i, cov, sizeof(cov));
  }
}

void orc_calc_covariances_f(const oracle *o, const orc_cloud *c, float *covs /* 9 per point */) {
  if (o->cfg.rbf_kernel_width > 0.0) { rbf_covariances_f(o, c, covs); return; }
  orc_grid g;
  memset(&g, 0, sizeof(g));
  grid_build(&g, c, o->cfg.voxel_resolution);
  const int k = o->cfg.k_correspondences;
#ifdef _OPENMP
  const int nth = o->cfg.num_threads > 0 ? o->cfg.num_threads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
#pragma omp parallel for num_threads(nth) schedule(dynamic, 64)
  for (long i = 0; i < c->n; i++) {
    int idx[64];
    float d2[64];
    const int m = grid_knn(&g, c, c->xyz + 3 * i, k, 1e300, idx, d2);
    float mean[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < m; j++) {
      const float *pt = c->xyz + 3 * (long)idx[j];
      for (int a = 0; a < 3; a++) { mean[a] += pt[a]; for (int b = 0; b < 3; b++) cov[a * 3 + b] += pt[a] * pt[b]; }
    }
    for (int a = 0; a < 3; a++) mean[a] /= (float)k;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov[a * 3 + b] = cov[a * 3 + b] / (float)k - mean[a] * mean[b];
    regularize_f(o->cfg.regularization, cov);
    memcpy(covs + 9 * i, cov, sizeof(cov));
  }
  grid_free(&g);
}

/* ---- model state ---------------------------------------------------------------------------- */
typedef struct orc_vgvox { double mean[3], cov[9]; int n; } orc_vgvox;

typedef struct orc_gicp_state {
  double *src_cov, *tgt_cov;    /* 9 doubles per point */
  long src_n, tgt_n;
  orc_grid tgt_grid;            /* exact 1-NN of GICP */
  int tgt_grid_valid;
  int *corr;                    /* GICP: target index per source point; VGICP: voxel per (point, offset) */
  double *maha;                 /* 9 doubles per correspondence */
  long corr_cap;
  /* VGICP voxel map */
  orc_vhash vh;
  orc_vgvox *vox;
  long nvox;
  int vmap_valid;
} orc_gicp_state;

static orc_gicp_state *st(oracle *o) {
  if (!o->gicp) o->gicp = calloc(1, sizeof(orc_gicp_state));
  return (orc_gicp_state *)o->gicp;
}

void orc_gicp_invalidate(oracle *o, int target) {
  if (!o->gicp) return;
  orc_gicp_state *s = (orc_gicp_state *)o->gicp;
  if (target) { free(s->tgt_cov); s->tgt_cov = NULL; s->tgt_n = 0; s->tgt_grid_valid = 0; s->vmap_valid = 0; }
  else { free(s->src_cov); s->src_cov = NULL; s->src_n = 0; }
}

void orc_gicp_swap(oracle *o) {
  if (!o->gicp) return;
  orc_gicp_state *s = (orc_gicp_state *)o->gicp;
  double *t = s->src_cov; s->src_cov = s->tgt_cov; s->tgt_cov = t;
  long n = s->src_n; s->src_n = s->tgt_n; s->tgt_n = n;
  s->tgt_grid_valid = 0;
  s->vmap_valid = 0;
}

void orc_gicp_free(oracle *o) {
  if (!o->gicp) return;
  orc_gicp_state *s = (orc_gicp_state *)o->gicp;
  free(s->src_cov); free(s->tgt_cov); grid_free(&s->tgt_grid); free(s->corr); free(s->maha);
  orc_vhash_free(&s->vh); free(s->vox);
  free(s);
  o->gicp = NULL;
}

/* GaussianVoxelMap::create_voxelmap (ADDITIVE)  fast_vgicp_voxel.hpp:129-156 */
static void vgicp_build_voxelmap(oracle *o, orc_gicp_state *s) {
  orc_vhash_free(&s->vh);
  free(s->vox);
  const double res = o->cfg.voxel_resolution;
  orc_vhash_init(&s->vh, o->tgt.n);
  s->vox = (orc_vgvox *)calloc((size_t)o->tgt.n + 1, sizeof(orc_vgvox));
  const int mult = o->cfg.voxel_mode == 2;   /* MultiplicativeGaussianVoxel  fast_vgicp_voxel.hpp:79-102 */
  for (long i = 0; i < o->tgt.n; i++) {
    const float *p = o->tgt.xyz + 3 * i;
    int c[3];
    for (int a = 0; a < 3; a++) c[a] = (int)floor((double)p[a] / res - 0.5);       /* voxel_coord on Vector4d  :158-160 */
    orc_vgvox *v = &s->vox[orc_vhash_insert(&s->vh, c[0], c[1], c[2])];
    v->n++;
    if (mult) {   /* cov_inv = cov_ with (3,3) = 1, inverted (block inverse = the 3x3 inverse); cov += cov_inv; mean += cov_inv * mean_ */
      double Ci[9];
      orc_inv3d(s->tgt_cov + 9 * i, Ci);
      for (int a = 0; a < 9; a++) v->cov[a] += Ci[a];
      for (int a = 0; a < 3; a++) v->mean[a] += (Ci[a * 3 + 0] * (double)p[0] + Ci[a * 3 + 1] * (double)p[1]) + Ci[a * 3 + 2] * (double)p[2];
    } else {      /* AdditiveGaussianVoxel (ADDITIVE and ADDITIVE_WEIGHTED)  :104-122 */
      for (int a = 0; a < 3; a++) v->mean[a] += (double)p[a];
      for (int a = 0; a < 9; a++) v->cov[a] += s->tgt_cov[9 * i + a];
    }
  }
  s->nvox = s->vh.count;
  for (long j = 0; j < s->nvox; j++) {                                               /* finalize() */
    if (mult) {   /* cov = cov^-1 ; mean = cov * mean  :96-101 */
      double C[9], m[3];
      orc_inv3d(s->vox[j].cov, C);
      for (int a = 0; a < 3; a++) m[a] = (C[a * 3 + 0] * s->vox[j].mean[0] + C[a * 3 + 1] * s->vox[j].mean[1]) + C[a * 3 + 2] * s->vox[j].mean[2];
      memcpy(s->vox[j].cov, C, sizeof(C));
      memcpy(s->vox[j].mean, m, sizeof(m));
    } else {
      for (int a = 0; a < 3; a++) s->vox[j].mean[a] /= s->vox[j].n;
      for (int a = 0; a < 9; a++) s->vox[j].cov[a] /= s->vox[j].n;
    }
  }
  s->vmap_valid = 1;
}

void orc_gicp_prepare(oracle *o) {
  orc_gicp_state *s = st(o);
  if (!s->src_cov || s->src_n != o->src.n) {          /* computeTransformation: lazy covariances  fast_gicp_impl.hpp:102-110 */
    free(s->src_cov);
    s->src_cov = (double *)malloc(sizeof(double) * 9 * (size_t)(o->src.n + 1));
    orc_calc_covariances(o, &o->src, s->src_cov);
    s->src_n = o->src.n;
  }
  if (!s->tgt_cov || s->tgt_n != o->tgt.n) {
    free(s->tgt_cov);
    s->tgt_cov = (double *)malloc(sizeof(double) * 9 * (size_t)(o->tgt.n + 1));
    orc_calc_covariances(o, &o->tgt, s->tgt_cov);
    s->tgt_n = o->tgt.n;
    s->tgt_grid_valid = 0;
    s->vmap_valid = 0;
  }
  if (o->cfg.model == ORC_MODEL_GICP && !s->tgt_grid_valid) { grid_build(&s->tgt_grid, &o->tgt, o->cfg.voxel_resolution); s->tgt_grid_valid = 1; }
  if (o->cfg.model == ORC_MODEL_VGICP && !s->vmap_valid) vgicp_build_voxelmap(o, s);   /* FastVGICP rebuilds it every align; same content */
}

static const int VG_DIRECT7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};

static void vg_offsets(int n, int out[27][3]) {
  if (n == 27) { int t = 0; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { out[t][0] = i - 1; out[t][1] = j - 1; out[t][2] = k - 1; t++; } }
  else for (int t = 0; t < n && t < 7; t++) memcpy(out[t], VG_DIRECT7[t], sizeof(int) * 3);
}

/* M = (cov_B + T cov_A T^T)^-1 with (3,3) forced to 1 before and 0 after: the reference inverts the 4x4 matrix
 * (fast_gicp_impl.hpp:146-150), i.e. Eigen's Packet2d 4x4 inverse (Eigen/src/LU/arch/InverseSize4.h, orc_eig_inv4d);
 * the 3x3 block of the result is what the cost reads. */
static void maha3(const double CB[9], const double CA[9], const double T[16], double M[9]) {
  double RC[9], S4[16], I4[16];
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) RC[a * 3 + b] = T[a * 4 + 0] * CA[0 * 3 + b] + T[a * 4 + 1] * CA[1 * 3 + b] + T[a * 4 + 2] * CA[2 * 3 + b];
  for (int a = 0; a < 16; a++) S4[a] = 0.0;
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) S4[a * 4 + b] = CB[a * 3 + b] + (RC[a * 3 + 0] * T[b * 4 + 0] + RC[a * 3 + 1] * T[b * 4 + 1] + RC[a * 3 + 2] * T[b * 4 + 2]);
  S4[15] = 1.0;
  orc_eig_inv4d(S4, I4);
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) M[a * 3 + b] = I4[a * 4 + b];
}

static double gicp_pass(oracle *o, const double T[16], int update, double *H, double *b) {
  orc_gicp_state *s = st(o);
  orc_gicp_prepare(o);
  const int vg = o->cfg.model == ORC_MODEL_VGICP;
  const int nO = vg ? (o->cfg.num_neighbors == 27 ? 27 : (o->cfg.num_neighbors == 7 ? 7 : 1)) : 1;
  int offs[27][3];
  vg_offsets(nO, offs);
  const long n = o->src.n;
  if (s->corr_cap < n * nO) {
    free(s->corr); free(s->maha);
    s->corr = (int *)malloc(sizeof(int) * (size_t)(n * nO + 1));
    s->maha = (double *)malloc(sizeof(double) * 9 * (size_t)(n * nO + 1));
    s->corr_cap = n * nO;
  }
  float Rf[9], tf[3];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rf[i * 3 + j] = (float)T[i * 4 + j]; tf[i] = (float)T[i * 4 + 3]; }
  const double thr2 = o->cfg.max_corr_dist * o->cfg.max_corr_dist;
  const double res = o->cfg.voxel_resolution;
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
    for (long i = 0; i < n; i++) {
      const float *p = o->src.xyz + 3 * i;
      const double pa[3] = {p[0], p[1], p[2]};
      double q[3];
      for (int a = 0; a < 3; a++) q[a] = T[a * 4 + 0] * pa[0] + T[a * 4 + 1] * pa[1] + T[a * 4 + 2] * pa[2] + T[a * 4 + 3];
      for (int k = 0; k < nO; k++) {
        int j;
        if (update) {
          if (vg) {   /* voxel_coord(trans * mean_A) + offset  fast_vgicp_impl.hpp:86-95 */
            int c[3];
            for (int a = 0; a < 3; a++) c[a] = (int)floor(q[a] / res - 0.5);
            j = orc_vhash_find(&s->vh, c[0] + offs[k][0], c[1] + offs[k][1], c[2] + offs[k][2]);
          } else {    /* pt = trans_f * p (float); kd-tree 1-NN; d2 < corr_dist^2  fast_gicp_impl.hpp:128-136 */
            float qf[3];
            for (int a = 0; a < 3; a++) qf[a] = (Rf[a * 3 + 0] * p[0] + Rf[a * 3 + 1] * p[1]) + Rf[a * 3 + 2] * p[2] + tf[a];
            int id; float d2;
            const int m = grid_knn(&s->tgt_grid, &o->tgt, qf, 1, thr2, &id, &d2);
            j = (m == 1 && (double)d2 < thr2) ? id : -1;
          }
          s->corr[i * nO + k] = j;
          if (j >= 0) maha3(vg ? s->vox[j].cov : s->tgt_cov + 9 * (long)j, s->src_cov + 9 * i, T, s->maha + 9 * (i * nO + k));
        } else {
          j = s->corr[i * nO + k];
        }
        if (j < 0) continue;
        const double *M = s->maha + 9 * (i * nO + k);
        double mb[3], w = 1.0;
        if (vg) { memcpy(mb, s->vox[j].mean, sizeof(mb)); w = sqrt((double)s->vox[j].n); }   /* fast_vgicp_impl.hpp:149 */
        else for (int a = 0; a < 3; a++) mb[a] = (double)o->tgt.xyz[3 * (long)j + a];
        double e[3], Me[3];
        for (int a = 0; a < 3; a++) e[a] = mb[a] - q[a];
        for (int a = 0; a < 3; a++) Me[a] = M[a * 3 + 0] * e[0] + M[a * 3 + 1] * e[1] + M[a * 3 + 2] * e[2];
        A[42] += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
        A[43] += 1.0;
        if (!H) continue;
        const double J[3][6] = {{0, -q[2], q[1], -1, 0, 0}, {q[2], 0, -q[0], 0, -1, 0}, {-q[1], q[0], 0, 0, 0, -1}};   /* [skew(Tp), -I] */
        double JtM[6][3];
        for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) JtM[r][c] = J[0][r] * M[0 * 3 + c] + J[1][r] * M[1 * 3 + c] + J[2][r] * M[2 * 3 + c];
        for (int r = 0; r < 6; r++) {
          for (int c = 0; c < 6; c++) A[r * 6 + c] += w * (JtM[r][0] * J[0][c] + JtM[r][1] * J[1][c] + JtM[r][2] * J[2][c]);
          A[36 + r] += w * (JtM[r][0] * e[0] + JtM[r][1] * e[1] + JtM[r][2] * e[2]);
        }
      }
    }
  }
  double cost = 0.0, cnt = 0.0;
  if (H) memset(H, 0, 36 * sizeof(double));
  if (b) memset(b, 0, 6 * sizeof(double));
  for (int k = 0; k < nth; k++) {
    const double *A = acc + (size_t)k * 44;
    if (H) for (int a = 0; a < 36; a++) H[a] += A[a];
    if (b) for (int a = 0; a < 6; a++) b[a] += A[36 + a];
    cost += A[42]; cnt += A[43];
  }
  free(acc);
  o->num_inliers = (int)cnt;
  return cost;
}

double orc_gicp_linearize(oracle *o, const double T[16], double *H, double *b) { return gicp_pass(o, T, 1, H, b); }
double orc_gicp_compute_error(oracle *o, const double T[16]) { return gicp_pass(o, T, 0, NULL, NULL); }

/* unit hooks */
int orc_test_knn_exact(void *h, const float q[3], int k, int *idx, float *d2) {
  oracle *o = (oracle *)h;
  orc_grid g;
  memset(&g, 0, sizeof(g));
  grid_build(&g, &o->tgt, o->cfg.voxel_resolution);
  const int m = grid_knn(&g, &o->tgt, q, k, 1e300, idx, d2);
  grid_free(&g);
  return m;
}

void orc_test_covariances_f(void *h, int target, float *covs) {
  oracle *o = (oracle *)h;
  orc_calc_covariances_f(o, target ? &o->tgt : &o->src, covs);
}

void orc_test_covariances(void *h, int target, double *covs) {
  oracle *o = (oracle *)h;
  orc_calc_covariances(o, target ? &o->tgt : &o->src, covs);
}

/* ---- pcl::Registration::getFitnessScore(max_range)  (PCL is not in the reference tree; restated from pcl/registration/impl/
 * registration.hpp as the call sites use it: jueying_slam/src/localization.cpp:325-326, mapOptmization.cpp:693,719,
 * fast_gicp/src/align.cpp:63).  The input cloud is transformed by final_transformation_ in float (pcl::transformPointCloud),
 * every point looks up its exact nearest target point (kd-tree 1-NN), and the SQUARED distance is compared with max_range
 * itself (PCL's own quirk) and averaged over the points that pass; no point passes -> numeric_limits<double>::max(). ---- */
double orc_fitness_score(void *h, const float T[16], double max_range) {
  oracle *o = (oracle *)h;
  if (o->src.n <= 0 || o->tgt.n <= 0) return DBL_MAX;
  orc_grid g;
  memset(&g, 0, sizeof(g));
  grid_build(&g, &o->tgt, o->cfg.voxel_resolution > 0 ? o->cfg.voxel_resolution : 1.0);
  double sum = 0.0;
  long nr = 0;
#pragma omp parallel for reduction(+ : sum, nr) schedule(dynamic, 256)
  for (long i = 0; i < o->src.n; i++) {
    const float *p = o->src.xyz + 3 * i;
    float q[3];
    for (int a = 0; a < 3; a++) q[a] = T[a * 4 + 0] * p[0] + (T[a * 4 + 1] * p[1] + (T[a * 4 + 2] * p[2] + T[a * 4 + 3]));
    int idx[1];
    float d2[1];
    if (grid_knn(&g, &o->tgt, q, 1, 1e300, idx, d2) < 1) continue;
    if ((double)d2[0] <= max_range) { sum += (double)d2[0]; nr++; }
  }
  grid_free(&g);
  return nr > 0 ? sum / (double)nr : DBL_MAX;
}

/* pclomp GICP-BFGS: the correspondence step of computeTransformation  (ndt_omp/include/pclomp/gicp_omp_impl.hpp:405-472).
 * The handle is a GICP oracle whose source / target are *input_ / *target_ (covariances with ORC_REG_PCLOMP = pclomp's
 * computeCovariances).  Returns the pairs in source order and mahalanobis_[source] as 9 floats (row-major 3x3 block). */
int orc_gicp_bfgs_correspondences(void *h, const float transformation[16], const float guess[16], int *idx_src, int *idx_tgt, float *maha9, long *m_out) {
  oracle *o = (oracle *)h;
  if (o->cfg.model != ORC_MODEL_GICP) return -1;
  orc_gicp_state *s = st(o);
  orc_gicp_prepare(o);
  const double thr2 = o->cfg.max_corr_dist * o->cfg.max_corr_dist;   /* dist_threshold = corr_dist_threshold_^2  :397 */
  double R[9];
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {           /* transform_R(i,j) += double(transformation_(i,k)) * double(guess(k,j))  :416-419 */
    double v = 0.0;
    for (int k = 0; k < 4; k++) v += (double)transformation[a * 4 + k] * (double)guess[k * 4 + b];
    R[a * 3 + b] = v;
  }
  long m = 0;
  for (long i = 0; i < o->src.n; i++) {   /* serial: the result is sorted by source index anyway (:466-472) */
    const float *p = o->src.xyz + 3 * i;
    float out[3], q[3];
    for (int a = 0; a < 3; a++) out[a] = guess[a * 4 + 0] * p[0] + (guess[a * 4 + 1] * p[1] + (guess[a * 4 + 2] * p[2] + guess[a * 4 + 3]));   /* pcl::transformPointCloud(output, output, guess)  :398 */
    /* query = transformation_ * output[i]: Matrix4f * Vector4f, accumulated column by column (etor_product_packet_impl; CORE-1)  :428 */
    for (int a = 0; a < 3; a++) q[a] = ((transformation[a * 4 + 0] * out[0] + transformation[a * 4 + 1] * out[1]) + transformation[a * 4 + 2] * out[2]) + transformation[a * 4 + 3] * 1.f;
    int id; float d2;
    const int found = grid_knn(&s->tgt_grid, &o->tgt, q, 1, thr2, &id, &d2);
    if (!(found == 1 && (double)d2 < thr2)) continue;     /* nn_dists[0] < dist_threshold  :437 */
    const double *C1 = s->src_cov + 9 * i, *C2 = s->tgt_cov + 9 * (long)id;
    double M[9], tmp[9], inv[9];
    /* M = R * C1; temp = M * R^T; temp += C2  (:445-449): Matrix3d lazy products, each coefficient the fixed-size 3-term tree t0 + (t1 + t2) */
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) M[a * 3 + b] = R[a * 3 + 0] * C1[0 * 3 + b] + (R[a * 3 + 1] * C1[1 * 3 + b] + R[a * 3 + 2] * C1[2 * 3 + b]);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) tmp[a * 3 + b] = (M[a * 3 + 0] * R[b * 3 + 0] + (M[a * 3 + 1] * R[b * 3 + 1] + M[a * 3 + 2] * R[b * 3 + 2])) + C2[a * 3 + b];
    orc_eig_inv3d(tmp, inv);                                          /* temp.inverse()  :451 */
    if (idx_src) idx_src[m] = (int)i;
    if (idx_tgt) idx_tgt[m] = id;
    if (maha9) for (int a = 0; a < 9; a++) maha9[9 * m + a] = (float)inv[a];   /* M.cast<float>()  :452 */
    m++;
  }
  if (m_out) *m_out = m;
  return 0;
}

/* setSourceCovariances / setTargetCovariances  (impl/fast_gicp_impl.hpp:93-100): replace the per-point covariances (9 doubles each,
 * row-major, input order); they stay until the cloud is set again. */
int orc_set_covariances(void *h, int target, const double *cov9, long n) {
  oracle *o = (oracle *)h;
  orc_gicp_state *s = st(o);
  orc_gicp_prepare(o);
  if (n != (target ? o->tgt.n : o->src.n)) return -1;
  memcpy(target ? s->tgt_cov : s->src_cov, cov9, sizeof(double) * 9 * (size_t)n);
  if (target) s->vmap_valid = 0;   /* FastVGICP builds its voxel distributions from target_covs_ */
  return 0;
}

/*
 * oracle/orc_models_gauss.c -- Gaussian-voxel residual models of the CPU oracle
 * (TEST INFRASTRUCTURE ONLY): NDT point-to-distribution and distribution-to-
 * distribution as computed by the reference's CUDA core.
 *
 * What is restated here, and from where (paths relative to
 * /root/reference/src/pointcloud_match/fast_gicp):
 *   voxel coordinate ......... include/fast_gicp/cuda/vector3_hash.cuh:35-38   floor(x/res - 0.5), float
 *   voxel statistics ......... src/fast_gicp/cuda/gaussian_voxelmap.cu:122-148,178-198 (sum x, sum x x^T; mean, cov)
 *   MIN_EIG regularisation ... src/fast_gicp/cuda/covariance_regularization.cu:83-97 (eigenvalues clamped to >= 1e-3)
 *   neighbour offsets ........ src/fast_gicp/cuda/ndt_cuda.cu:35-88 (DIRECT1 / DIRECT7 / DIRECT27)
 *   correspondences .......... src/fast_gicp/cuda/find_voxel_correspondences.cu:31-63 (one per (source, offset) hit)
 *   P2D / D2D derivatives .... src/fast_gicp/cuda/ndt_compute_derivatives.cu:15-18,33-102,104-175
 *   NDTCudaCore .............. src/fast_gicp/cuda/ndt_cuda.cu:13-177 (D2D source = source-voxel means, :156-158)
 *
 * Deliberate, documented differences from the reference's arithmetic:
 *   - The reference accumulates voxel sums with float atomics and reduces the 43-float
 *     derivative tuples with an unordered float tree (thrust): both are run-to-run
 *     order dependent.  Here the per-correspondence terms are formed in float exactly
 *     as written in the reference, but every SUM (voxel statistics, H, b, cost) is
 *     carried in double in input order: the order-independent limit of that arithmetic.
 *   - The reference's bucket table silently drops a voxel after 10 failed probes
 *     (gaussian_voxelmap.cu:36-51); here every voxel is kept.
 *   - 3x3 symmetric eigen-decomposition: cyclic Jacobi in double instead of Eigen's
 *     float computeDirect; V diag(max(l,1e-3)) V^T is the same matrix up to rounding.
 */
#include "orc_internal.h"

#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct orc_gvox {
  float mean[3];
  float cov[9];   /* row-major, regularised */
  int n;
} orc_gvox;

typedef struct orc_gmap {
  orc_vhash h;
  orc_gvox *vox;
  long nvox;
  int valid;
} orc_gmap;

typedef struct orc_gauss_state {
  orc_gmap tgt, src;
  float *src_cov_f, *tgt_cov_f;   /* VGICP_CUDA: per-point float covariances (9 each) */
  long src_cov_n, tgt_cov_n;
  int *corr;          /* [n_elems * n_offsets] target voxel index or -1, from the last linearize */
  long corr_cap;
  float lin_R[9];     /* linearized_x rotation (R_eval of the D2D kernel) */
} orc_gauss_state;

static const int ORC_DIRECT7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};

#define ORC_MAX_OFFSETS 343   /* DIRECT_RADIUS up to radius 3 */

/* NeighborSearchMethod::DIRECT_RADIUS  ndt_cuda.cu:70-83: i, j, k loops over [-range, range], kept when |offset| <= radius + 1e-3 */
static int offsets_radius(double radius, int out[ORC_MAX_OFFSETS][3]) {
  const int range = (int)ceil(radius);
  int t = 0;
  for (int i = -range; i <= range; i++)
    for (int j = -range; j <= range; j++)
      for (int k = -range; k <= range; k++)
        if (sqrt((double)(i * i + j * j + k * k)) <= radius + 1e-3 && t < ORC_MAX_OFFSETS) { out[t][0] = i; out[t][1] = j; out[t][2] = k; t++; }
  return t;
}

static void offsets_for(int n, int out[ORC_MAX_OFFSETS][3]) {
  if (n == 27) {
    int t = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { out[t][0] = i - 1; out[t][1] = j - 1; out[t][2] = k - 1; t++; }
  } else {
    for (int t = 0; t < n && t < 7; t++) { out[t][0] = ORC_DIRECT7[t][0]; out[t][1] = ORC_DIRECT7[t][1]; out[t][2] = ORC_DIRECT7[t][2]; }
  }
}

/* calc_voxel_coord: (x.array() / resolution - 0.5).floor().cast<int>() on Vector3f */
static inline void gauss_coord(float res, const float p[3], int c[3]) {
  for (int a = 0; a < 3; a++) c[a] = (int)floorf(p[a] / res - 0.5f);
}

static void gmap_free(orc_gmap *m) {
  orc_vhash_free(&m->h);
  free(m->vox);
  memset(m, 0, sizeof(*m));
}

/* create_voxelmap(points) + covariance_regularization(MIN_EIG)  (ndt_cuda.cu:120-140) */
static void gmap_build(orc_gmap *m, const orc_cloud *cl, float res) {
  gmap_free(m);
  const long n = cl->n;
  orc_vhash_init(&m->h, n);
  int *pv = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (long i = 0; i < n; i++) {
    int c[3];
    gauss_coord(res, cl->xyz + 3 * i, c);
    pv[i] = orc_vhash_insert(&m->h, c[0], c[1], c[2]);
  }
  m->nvox = m->h.count;
  double *sx = (double *)calloc((size_t)m->nvox * 3 + 1, sizeof(double));
  double *sxx = (double *)calloc((size_t)m->nvox * 9 + 1, sizeof(double));
  int *cnt = (int *)calloc((size_t)m->nvox + 1, sizeof(int));
  for (long i = 0; i < n; i++) {   /* input order; the float products x x^T are formed as in the reference */
    const float *p = cl->xyz + 3 * i;
    const int v = pv[i];
    cnt[v]++;
    for (int a = 0; a < 3; a++) {
      sx[v * 3 + a] += (double)p[a];
      for (int b = 0; b < 3; b++) sxx[v * 9 + a * 3 + b] += (double)(p[a] * p[b]);
    }
  }
  m->vox = (orc_gvox *)calloc((size_t)m->nvox + 1, sizeof(orc_gvox));
  for (long v = 0; v < m->nvox; v++) {
    orc_gvox *g = &m->vox[v];
    const double nn = (double)cnt[v];
    double mean[3], cov[9];
    g->n = cnt[v];
    for (int a = 0; a < 3; a++) mean[a] = sx[v * 3 + a] / nn;
    /* voxel_covs = (sum x x^T - mean * sum x^T) / n   gaussian_voxelmap.cu:193-194 */
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov[a * 3 + b] = (sxx[v * 9 + a * 3 + b] - mean[a] * sx[v * 3 + b]) / nn;
    /* covariance_regularization_mineig (covariance_regularization.cu:83-97) on the float voxel covariance: computeDirect
     * (closed form, reads the lower triangle; orc_eigen.h), eigenvalues clamped at 1e-3, V diag V^-1 */
    float cf[9], w[3], Vf[9], Vi[9], VD[9];
    for (int a = 0; a < 9; a++) cf[a] = (float)cov[a];
    orc_eig_direct3f(cf, w, Vf);
    for (int k = 0; k < 3; k++) w[k] = fmaxf(1e-3f, w[k]);
    orc_inv3f(Vf, Vi);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) VD[a * 3 + b] = Vf[a * 3 + b] * w[b];
    for (int a = 0; a < 3; a++) {
      g->mean[a] = (float)mean[a];
      for (int b = 0; b < 3; b++) g->cov[a * 3 + b] = (VD[a * 3 + 0] * Vi[0 * 3 + b] + VD[a * 3 + 1] * Vi[1 * 3 + b]) + VD[a * 3 + 2] * Vi[2 * 3 + b];
    }
  }
  free(sx); free(sxx); free(cnt); free(pv);
  m->valid = 1;
}

/* GaussianVoxelMap::create_voxelmap(points, covariances): voxel mean = mean of the points, voxel covariance = mean of
 * the point covariances (accumulate_points_kernel + finalize_voxels_kernel, gaussian_voxelmap.cu:75-148,150-169) */
static void gmap_build_vgc(orc_gmap *m, const orc_cloud *cl, float res, const float *covs) {
  gmap_free(m);
  const long n = cl->n;
  orc_vhash_init(&m->h, n);
  int *pv = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (long i = 0; i < n; i++) {
    int c[3];
    gauss_coord(res, cl->xyz + 3 * i, c);
    pv[i] = orc_vhash_insert(&m->h, c[0], c[1], c[2]);
  }
  m->nvox = m->h.count;
  double *sx = (double *)calloc((size_t)m->nvox * 3 + 1, sizeof(double));
  double *sc = (double *)calloc((size_t)m->nvox * 9 + 1, sizeof(double));
  int *cnt = (int *)calloc((size_t)m->nvox + 1, sizeof(int));
  for (long i = 0; i < n; i++) {
    const int v = pv[i];
    cnt[v]++;
    for (int a = 0; a < 3; a++) sx[v * 3 + a] += (double)cl->xyz[3 * i + a];
    for (int a = 0; a < 9; a++) sc[v * 9 + a] += (double)covs[9 * i + a];
  }
  m->vox = (orc_gvox *)calloc((size_t)m->nvox + 1, sizeof(orc_gvox));
  for (long v = 0; v < m->nvox; v++) {
    orc_gvox *g = &m->vox[v];
    g->n = cnt[v];
    for (int a = 0; a < 3; a++) g->mean[a] = (float)(sx[v * 3 + a] / (double)cnt[v]);
    for (int a = 0; a < 9; a++) g->cov[a] = (float)(sc[v * 9 + a] / (double)cnt[v]);
  }
  free(sx); free(sc); free(cnt); free(pv);
  m->valid = 1;
}

static orc_gauss_state *gs(oracle *o) {
  if (!o->gauss) o->gauss = (orc_gauss_state *)calloc(1, sizeof(orc_gauss_state));
  return o->gauss;
}

void orc_gauss_invalidate(oracle *o, int target) {
  if (!o->gauss) return;
  if (target) { o->gauss->tgt.valid = 0; o->gauss->tgt_cov_n = -1; } else { o->gauss->src.valid = 0; o->gauss->src_cov_n = -1; }
}

void orc_gauss_swap(oracle *o) {
  if (!o->gauss) return;
  orc_gmap t = o->gauss->tgt; o->gauss->tgt = o->gauss->src; o->gauss->src = t;
  o->gauss->tgt.valid = 0; o->gauss->src.valid = 0; o->gauss->src_cov_n = -1; o->gauss->tgt_cov_n = -1;
  /* a map that was never built for the other role is simply rebuilt lazily */
}

void orc_gauss_free(oracle *o) {
  if (!o->gauss) return;
  gmap_free(&o->gauss->tgt);
  gmap_free(&o->gauss->src);
  free(o->gauss->corr);
  free(o->gauss->src_cov_f); free(o->gauss->tgt_cov_f);
  free(o->gauss);
  o->gauss = NULL;
}

/* NDTCudaCore::create_voxelmaps  (ndt_cuda.cu:115-140) */
void orc_gauss_prepare(oracle *o) {
  orc_gauss_state *g = gs(o);
  const float res = (float)o->cfg.voxel_resolution;
  if (o->cfg.model == ORC_MODEL_VGICP_CUDA) {   /* FastVGICPCuda: covariances of both clouds, voxel map of the target */
    if (g->src_cov_n != o->src.n) {
      free(g->src_cov_f);
      g->src_cov_f = (float *)malloc(sizeof(float) * 9 * (size_t)(o->src.n + 1));
      orc_calc_covariances_f(o, &o->src, g->src_cov_f);
      g->src_cov_n = o->src.n;
    }
    if (g->tgt_cov_n != o->tgt.n || !g->tgt.valid) {
      free(g->tgt_cov_f);
      g->tgt_cov_f = (float *)malloc(sizeof(float) * 9 * (size_t)(o->tgt.n + 1));
      orc_calc_covariances_f(o, &o->tgt, g->tgt_cov_f);
      g->tgt_cov_n = o->tgt.n;
      gmap_build_vgc(&g->tgt, &o->tgt, res, g->tgt_cov_f);
    }
    return;
  }
  if (!g->tgt.valid) gmap_build(&g->tgt, &o->tgt, res);
  if (o->cfg.model == ORC_MODEL_NDT_D2D && !g->src.valid) gmap_build(&g->src, &o->src, res);
}

static inline float cauchy(float k, float x) {   /* ndt_compute_derivatives.cu:15-18 */
  const float k_sq = k * k;
  return k_sq / (k_sq + x * x);
}

/* number of source elements: points (P2D) or source voxels (D2D) */
static long n_elems(const oracle *o) { return o->cfg.model == ORC_MODEL_NDT_D2D ? o->gauss->src.nvox : o->src.n; }

/* One pass over the correspondences.  update = 1: find them at `T` (update_correspondences) and
 * remember them; update = 0: re-use the remembered list (compute_error contract). */
static double ndt_pass(oracle *o, const double T[16], int update, double *H, double *b) {
  orc_gauss_state *g = gs(o);
  orc_gauss_prepare(o);
  const int d2d = o->cfg.model == ORC_MODEL_NDT_D2D;
  const int vgc = o->cfg.model == ORC_MODEL_VGICP_CUDA;   /* compute_derivatives.cu:49-92: the D2D form with per-point cov_A, w = sqrt(n), no Cauchy weight */
  int nO = o->cfg.num_neighbors == 27 ? 27 : (o->cfg.num_neighbors == 1 ? 1 : 7);
  int offs[ORC_MAX_OFFSETS][3];
  if (o->nb_radius > 0.0) nO = offsets_radius(o->nb_radius, offs);
  else offsets_for(nO, offs);
  const float res = (float)o->cfg.voxel_resolution;
  float R[9], t[3];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[i * 3 + j] = (float)T[i * 4 + j]; t[i] = (float)T[i * 4 + 3]; }
  const long n = n_elems(o);
  if (update) {
    if (g->corr_cap < n * nO) { free(g->corr); g->corr = (int *)malloc(sizeof(int) * (size_t)(n * nO + 1)); g->corr_cap = n * nO; }
    memcpy(g->lin_R, R, sizeof(R));   /* linearized_x  ndt_cuda.cu:149 */
  }
  const float *Re = g->lin_R;
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
#pragma omp for schedule(static)
    for (long i = 0; i < n; i++) {
      const float *pa = d2d ? g->src.vox[i].mean : o->src.xyz + 3 * i;
      float q[3];
      for (int a = 0; a < 3; a++) q[a] = (R[a * 3 + 0] * pa[0] + R[a * 3 + 1] * pa[1]) + R[a * 3 + 2] * pa[2] + t[a];
      int c[3];
      if (update) gauss_coord(res, q, c);
      for (int k = 0; k < nO; k++) {
        int v;
        if (update) {
          v = orc_vhash_find(&g->tgt.h, c[0] + offs[k][0], c[1] + offs[k][1], c[2] + offs[k][2]);
          g->corr[i * nO + k] = v;
        } else {
          v = g->corr[i * nO + k];
        }
        if (v < 0) continue;
        const orc_gvox *B = &g->tgt.vox[v];
        if (vgc ? B->n <= 0 : B->n <= 6) continue;   /* ndt_compute_derivatives.cu:61,132 | compute_derivatives.cu:62-64 */
        float C[9], M[9];
        memcpy(C, B->cov, sizeof(C));
        if (d2d || vgc) {   /* RCR = R_eval * cov_A * R_eval^T ; M = (cov_B + RCR)^-1   :145-146 */
          const float *CA = vgc ? g->src_cov_f + 9 * i : g->src.vox[i].cov;
          float RC[9];
          for (int a = 0; a < 3; a++) for (int bb = 0; bb < 3; bb++) RC[a * 3 + bb] = (Re[a * 3 + 0] * CA[0 * 3 + bb] + Re[a * 3 + 1] * CA[1 * 3 + bb]) + Re[a * 3 + 2] * CA[2 * 3 + bb];
          for (int a = 0; a < 3; a++) for (int bb = 0; bb < 3; bb++) C[a * 3 + bb] += (RC[a * 3 + 0] * Re[bb * 3 + 0] + RC[a * 3 + 1] * Re[bb * 3 + 1]) + RC[a * 3 + 2] * Re[bb * 3 + 2];
        }
        orc_inv3f(C, M);
        float e[3];
        for (int a = 0; a < 3; a++) e[a] = B->mean[a] - q[a];
        const float en = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
        const float w = vgc ? sqrtf((float)B->n) : cauchy(res, en);   /* :78,150 | compute_derivatives.cu:78 */
        float Me[3];
        for (int a = 0; a < 3; a++) Me[a] = (M[a * 3 + 0] * e[0] + M[a * 3 + 1] * e[1]) + M[a * 3 + 2] * e[2];
        const float err = w * ((e[0] * Me[0] + e[1] * Me[1]) + e[2] * Me[2]);
        A[42] += (double)err;
        A[43] += 1.0;
        if (!H) continue;
        /* J = [skew(q), -I] (3x6);  H = w J^T M J ; b = w J^T M e */
        float J[3][6] = {{0.f, -q[2], q[1], -1.f, 0.f, 0.f}, {q[2], 0.f, -q[0], 0.f, -1.f, 0.f}, {-q[1], q[0], 0.f, 0.f, 0.f, -1.f}};
        float JtM[6][3];
        for (int r = 0; r < 6; r++) for (int cc = 0; cc < 3; cc++) JtM[r][cc] = (w * J[0][r] * M[0 * 3 + cc] + w * J[1][r] * M[1 * 3 + cc]) + w * J[2][r] * M[2 * 3 + cc];
        for (int r = 0; r < 6; r++) {
          for (int cc = 0; cc < 6; cc++) A[r * 6 + cc] += (double)((JtM[r][0] * J[0][cc] + JtM[r][1] * J[1][cc]) + JtM[r][2] * J[2][cc]);
          A[36 + r] += (double)((JtM[r][0] * e[0] + JtM[r][1] * e[1]) + JtM[r][2] * e[2]);
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
    cost += A[42];
    cnt += A[43];
  }
  free(acc);
  o->num_inliers = (int)cnt;
  return cost;
}

double orc_gauss_linearize(oracle *o, const double T[16], double *H, double *b) {
  double Hl[36], bl[6];
  if (o->cfg.model != ORC_MODEL_NDT_P2D && o->cfg.model != ORC_MODEL_NDT_D2D && o->cfg.model != ORC_MODEL_VGICP_CUDA) return orc_gicp_linearize(o, T, H, b);   /* orc_gicp.c */
  const double c = ndt_pass(o, T, 1, Hl, bl);
  if (H) memcpy(H, Hl, sizeof(Hl));
  if (b) memcpy(b, bl, sizeof(bl));
  return c;
}

double orc_gauss_compute_error(oracle *o, const double T[16]) {
  if (o->cfg.model != ORC_MODEL_NDT_P2D && o->cfg.model != ORC_MODEL_NDT_D2D && o->cfg.model != ORC_MODEL_VGICP_CUDA) return orc_gicp_compute_error(o, T);
  return ndt_pass(o, T, 0, NULL, NULL);
}

/* unit hook: voxel statistics of the target map at a point */
int orc_test_gauss_voxel(void *h, const float p[3], float mean[3], float cov[9], int *n) {
  oracle *o = (oracle *)h;
  orc_gauss_prepare(o);
  int c[3];
  gauss_coord((float)o->cfg.voxel_resolution, p, c);
  const int v = orc_vhash_find(&o->gauss->tgt.h, c[0], c[1], c[2]);
  if (v < 0) return 0;
  memcpy(mean, o->gauss->tgt.vox[v].mean, 3 * sizeof(float));
  memcpy(cov, o->gauss->tgt.vox[v].cov, 9 * sizeof(float));
  *n = o->gauss->tgt.vox[v].n;
  return 1;
}

/*
 * oracle/pcm_oracle.c -- CPU restatement of the reference's scan-to-submap
 * registration hot path (matiable/pointcloud-slam, V2.1.0).
 *
 * TEST INFRASTRUCTURE ONLY (see pcm_oracle.h): parity checker + reported CPU
 * baseline.  "parity unpinned" by the reference's own fixtures; self-pinned by
 * the analytic KATs in tests/test_oracle_*.py.
 *
 * What is restated here, and from where (paths relative to /root/reference/src):
 *   GN/LM outer loop ........ pointcloud_match/fast_gicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp:52-172
 *   so3_exp ................. pointcloud_match/fast_gicp/include/fast_gicp/so3/so3.hpp:58-77
 *   P2PLANE matcher ......... jueying_lio/src/laser_mapping.cc:592-701 (ObsModel)
 *   iVox 5-NN ............... jueying_lio/include/ivox3d/ivox3d.h:132-204,211-235,283-286
 *                             jueying_lio/include/ivox3d/ivox3d_node.hpp:13-16,140-205
 *   plane fit ............... jueying_lio/include/common_lib.h:186-243
 * Other residual models (GICP / VGICP / NDT) live in orc_models_gauss.c.
 */
#include "pcm_oracle.h"
#include "orc_internal.h"

#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* config                                                                     */
/* ------------------------------------------------------------------------- */
void orc_default_config(orc_config *c) {
  memset(c, 0, sizeof(*c));
  c->model = ORC_MODEL_P2PLANE;
  c->optimizer = ORC_OPT_LM;            /* lsq_registration_impl.hpp:15 */
  c->max_iterations = 64;               /* :11 */
  c->rotation_eps = 2e-3;               /* :12 */
  c->translation_eps = 5e-4;            /* :13 */
  c->lm_max_iterations = 10;            /* :17 */
  c->lm_init_lambda_factor = 1e-9;      /* :18 */
  c->voxel_resolution = 0.5;
  c->num_neighbors = 27;
  c->knn = 5;                           /* options.h:14 */
  c->min_knn = 3;                       /* options.h:15 */
  c->max_range = 5.0;                   /* ivox3d.h:80 */
  c->plane_threshold = 0.1;             /* options.cc:10 */
  c->max_corr_dist = (double)FLT_MAX;   /* fast_gicp_impl.hpp:18 */
  c->k_correspondences = 20;            /* fast_gicp_impl.hpp:16 */
  c->regularization = ORC_REG_PLANE;    /* fast_gicp_impl.hpp:20 */
  c->num_threads = 0;
  c->map_capacity = 1000000;            /* ivox3d.h:57 */
  c->ndt_step_size = 0.1;               /* ndt_omp_impl.hpp:48 */
  c->ndt_outlier_ratio = 0.55;          /* ndt_omp_impl.hpp:48 */
}

static int orc_threads(const oracle *o) {
#ifdef _OPENMP
  return o->cfg.num_threads > 0 ? o->cfg.num_threads : omp_get_max_threads();
#else
  (void)o;
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* voxel hash: semantics of std::unordered_map<Vector3i, ...> (exact lookups) */
/* ------------------------------------------------------------------------- */
/* bucket hash of jueying_lio/include/ivox3d/eigen_types.h:72-76 (only bucket
 * placement depends on it, never a result) */
static inline unsigned long orc_hash3(int x, int y, int z) {
  unsigned int hx = (unsigned int)x * 73856093u, hy = (unsigned int)y * 471943u, hz = (unsigned int)z * 83492791u;
  return (unsigned long)(size_t)(int)(hx ^ hy ^ hz) % 10000000ul;
}

void orc_vhash_init(orc_vhash *h, long expected) {
  long cap = 64;
  while (cap < expected * 2) cap <<= 1;
  h->cap = cap;
  h->keys = (int *)malloc(sizeof(int) * 3 * cap);
  h->val = (int *)malloc(sizeof(int) * cap);
  for (long i = 0; i < cap; i++) h->val[i] = -1;
  h->count = 0;
}

void orc_vhash_free(orc_vhash *h) {
  free(h->keys);
  free(h->val);
  memset(h, 0, sizeof(*h));
}

int orc_vhash_find(const orc_vhash *h, int x, int y, int z) {
  if (h->cap == 0) return -1;
  unsigned long i = orc_hash3(x, y, z) & (unsigned long)(h->cap - 1);
  for (;;) {
    if (h->val[i] < 0) return -1;
    const int *k = h->keys + 3 * i;
    if (k[0] == x && k[1] == y && k[2] == z) return h->val[i];
    i = (i + 1) & (unsigned long)(h->cap - 1);
  }
}

/* insert-or-get; returns the value (new values are assigned sequentially) */
int orc_vhash_insert(orc_vhash *h, int x, int y, int z) {
  unsigned long i = orc_hash3(x, y, z) & (unsigned long)(h->cap - 1);
  for (;;) {
    if (h->val[i] < 0) {
      int *k = h->keys + 3 * i;
      k[0] = x; k[1] = y; k[2] = z;
      h->val[i] = (int)h->count++;
      return h->val[i];
    }
    const int *k = h->keys + 3 * i;
    if (k[0] == x && k[1] == y && k[2] == z) return h->val[i];
    i = (i + 1) & (unsigned long)(h->cap - 1);
  }
}

/* ------------------------------------------------------------------------- */
/* clouds                                                                     */
/* ------------------------------------------------------------------------- */
static void cloud_set(orc_cloud *c, const float *xyz, long n, long stride) {
  free(c->xyz);
  c->n = n;
  c->xyz = (float *)malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
  for (long i = 0; i < n; i++) {
    c->xyz[3 * i + 0] = xyz[i * stride + 0];
    c->xyz[3 * i + 1] = xyz[i * stride + 1];
    c->xyz[3 * i + 2] = xyz[i * stride + 2];
  }
}

/* ------------------------------------------------------------------------- */
/* iVox (static build): ivox3d.h:256-286, IVoxNode::InsertPoint               */
/* ------------------------------------------------------------------------- */
/* Pos2Grid: (pt * inv_resolution).array().round().cast<int>()  ivox3d.h:283-286
 * inv_resolution_ = 1.0 / resolution_ stored as float (ivox3d.h:55,67) */
void orc_ivox_key(const oracle *o, const float p[3], int key[3]) {
  float res = (float)o->cfg.voxel_resolution;
  float inv = (float)(1.0 / res);
  for (int a = 0; a < 3; a++) key[a] = (int)roundf(p[a] * inv);
}

static void ivox_free(orc_ivox *v) {
  orc_vhash_free(&v->h);
  free(v->vox_start);
  free(v->vox_pts);
  memset(v, 0, sizeof(*v));
}

static void ivox_build(oracle *o) {
  orc_ivox *v = &o->tgt_ivox;
  ivox_free(v);
  const long n = o->tgt.n;
  orc_vhash_init(&v->h, n);
  int *pv = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
  for (long i = 0; i < n; i++) {
    int k[3];
    orc_ivox_key(o, o->tgt.xyz + 3 * i, k);
    pv[i] = orc_vhash_insert(&v->h, k[0], k[1], k[2]);
  }
  v->nvox = v->h.count;
  v->vox_start = (int *)calloc((size_t)v->nvox + 1, sizeof(int));
  v->vox_pts = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
  for (long i = 0; i < n; i++) v->vox_start[pv[i] + 1]++;
  for (long j = 0; j < v->nvox; j++) v->vox_start[j + 1] += v->vox_start[j];
  int *fill = (int *)malloc(sizeof(int) * ((size_t)v->nvox + 1));
  memcpy(fill, v->vox_start, sizeof(int) * ((size_t)v->nvox + 1));
  for (long i = 0; i < n; i++) v->vox_pts[fill[pv[i]]++] = (int)i; /* insertion order = points_ order */
  free(fill);
  free(pv);
  v->valid = 1;
}

/* nearby_grids_ in the reference's order: ivox3d.h:211-235 */
static const int ORC_NEARBY[27][3] = {
  {0, 0, 0},   {-1, 0, 0},  {1, 0, 0},   {0, 1, 0},   {0, -1, 0},  {0, 0, -1},  {0, 0, 1},
  {1, 1, 0},   {-1, 1, 0},  {1, -1, 0},  {-1, -1, 0}, {1, 0, 1},   {-1, 0, 1},  {1, 0, -1},
  {-1, 0, -1}, {0, 1, 1},   {0, -1, 1},  {0, 1, -1},  {0, -1, -1}, {1, 1, 1},   {-1, 1, 1},
  {1, -1, 1},  {1, 1, -1},  {-1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};

typedef struct { double dist; int idx; } orc_distpt;

/* keep the K smallest of cand[lo..*n) (std::nth_element + resize); stable in
 * visit order for equal distances. */
static void sort_distpts(orc_distpt *c, int n);
static void keep_k_smallest(orc_distpt *cand, int lo, int *n, int K) {
  int m = *n - lo;
  if (m <= K) return;
  sort_distpts(cand + lo, m);
  *n = lo + K;
}

static void sort_distpts(orc_distpt *c, int n) {
  for (int i = 1; i < n; i++) {
    orc_distpt x = c[i];
    int j = i;
    while (j > 0 && x.dist < c[j - 1].dist) { c[j] = c[j - 1]; j--; }
    c[j] = x;
  }
}

/*
 * IVox::GetClosestPoint(pt, closest, max_num, max_range)  ivox3d.h:132-204 with
 * IVoxNode::KNNPointByCondition  ivox3d_node.hpp:140-205.
 * Output: up to K target indices.  ORC_KNN_ORDER_ASCENDING (default, what the HIP kernels produce): ascending distance,
 * equal distances in visit order.  ORC_KNN_ORDER_LIBSTDCXX: the order the reference's three std::nth_element calls leave
 * with libstdc++ (orc_knn_libstdcxx.cpp) -- the row order of the matrix esti_plane factorises, which moves the float plane
 * fit at rounding level (tests/test_knn_order.py measures by how much).
 */
void orc_std_nth_element(orc_distpt *first, int nth, int n);
void orc_set_knn_order(void *h, int order) { ((oracle *)h)->knn_order = order; }
int orc_ivox_knn(const oracle *o, const float q[3], int *idx_out, float *d2_out, orc_distpt_buf *buf) {
  const orc_ivox *v = &o->tgt_ivox;
  const int K = o->cfg.knn;
  const double max_r2 = o->cfg.max_range * o->cfg.max_range;
  int key[3];
  orc_ivox_key(o, q, key);
  int n = 0;
  orc_distpt *cand = (orc_distpt *)buf->data;
  for (int g = 0; g < o->cfg.num_neighbors; g++) {
    int vi = orc_vhash_find(&v->h, key[0] + ORC_NEARBY[g][0], key[1] + ORC_NEARBY[g][1], key[2] + ORC_NEARBY[g][2]);
    if (vi < 0) continue;
    int old = n;
    for (int s = v->vox_start[vi]; s < v->vox_start[vi + 1]; s++) {
      int pi = v->vox_pts[s];
      const float *p = o->tgt.xyz + 3 * (long)pi;
      /* distance2(): (pt1 - pt2).squaredNorm() in float, returned as double  ivox3d_node.hpp:13-16 */
      float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
      float d2 = dx * dx + dy * dy + dz * dz;
      double d = (double)d2;
      if (d < max_r2) {
        if ((size_t)(n + 1) * sizeof(orc_distpt) > buf->bytes) {
          buf->bytes = buf->bytes ? buf->bytes * 2 : 4096;
          buf->data = realloc(buf->data, buf->bytes);
          cand = (orc_distpt *)buf->data;
        }
        cand[n].dist = d;
        cand[n].idx = pi;
        n++;
      }
    }
    if (o->knn_order == ORC_KNN_ORDER_LIBSTDCXX) {   /* ivox3d_node.hpp:176-181 */
      if (n - old > K) { orc_std_nth_element(cand + old, K - 1, n - old); n = old + K; }
    } else {
      keep_k_smallest(cand, old, &n, K); /* per-voxel nth_element + resize */
    }
  }
  if (n == 0) return 0;
  if (o->knn_order == ORC_KNN_ORDER_LIBSTDCXX) {
    /* ivox3d.h:173-178, on the container's libstdc++ (orc_knn_libstdcxx.cpp) */
    if (n > K) { orc_std_nth_element(cand, K - 1, n); n = K; }
    orc_std_nth_element(cand, 0, n);
  } else {
    int lo = 0;
    keep_k_smallest(cand, lo, &n, K);
    sort_distpts(cand, n);
  }
  for (int i = 0; i < n; i++) {
    idx_out[i] = cand[i].idx;
    if (d2_out) d2_out[i] = (float)cand[i].dist;
  }
  return n;
}

/* ------------------------------------------------------------------------- */
/* common::esti_plane  common_lib.h:186-243                                   */
/* ------------------------------------------------------------------------- */
int orc_esti_plane(const float *pts, int n, int K, int min_pts, float threshold, float plane[4]) {
  if (n < min_pts) return 0;
  float normvec[3];
  if (n == K && K <= ORC_QR_MAXR) {
    float A[ORC_QR_MAXR * 3], b[ORC_QR_MAXR];
    for (int j = 0; j < n; j++) {
      A[j * 3 + 0] = pts[j * 3 + 0]; A[j * 3 + 1] = pts[j * 3 + 1]; A[j * 3 + 2] = pts[j * 3 + 2];
      b[j] = -1.0f;
    }
    orc_colpivqr3f(A, n, b, normvec);
  } else {
    double A[ORC_QR_MAXR * 3], b[ORC_QR_MAXR], x[3];
    for (int j = 0; j < n; j++) {
      A[j * 3 + 0] = pts[j * 3 + 0]; A[j * 3 + 1] = pts[j * 3 + 1]; A[j * 3 + 2] = pts[j * 3 + 2];
      b[j] = -1.0;
    }
    orc_colpivqr3d(A, n, b, x);
    normvec[0] = (float)x[0]; normvec[1] = (float)x[1]; normvec[2] = (float)x[2];
  }
  float nn = sqrtf(normvec[0] * normvec[0] + normvec[1] * normvec[1] + normvec[2] * normvec[2]);
  plane[0] = normvec[0] / nn;
  plane[1] = normvec[1] / nn;
  plane[2] = normvec[2] / nn;
  plane[3] = (float)(1.0 / (double)nn);
  for (int j = 0; j < n; j++) {
    float d = plane[0] * pts[j * 3 + 0] + plane[1] * pts[j * 3 + 1] + plane[2] * pts[j * 3 + 2] + plane[3];
    if (fabsf(d) > threshold) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------------- */
/* P2PLANE model: ObsModel under LsqRegistration::linearize                    */
/* ------------------------------------------------------------------------- */
static void iso_to_float(const double T[16], float R[9], float t[3]) {
  /* trans.cast<float>()  fast_gicp_impl.hpp:119 ; laser_mapping.cc:602-603 */
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[i * 3 + j] = (float)T[i * 4 + j];
    t[i] = (float)T[i * 4 + 3];
  }
}

static void p2plane_prepare(oracle *o) {
  if (!o->tgt_ivox.valid) ivox_build(o);
  if (o->plane_cap < o->src.n) {
    free(o->plane);
    free(o->selected);
    o->plane = (float *)malloc(sizeof(float) * 4 * (size_t)(o->src.n > 0 ? o->src.n : 1));
    o->selected = (unsigned char *)malloc((size_t)(o->src.n > 0 ? o->src.n : 1));
    o->plane_cap = o->src.n;
  }
}

/*
 * linearize for the P2PLANE model.
 *   per point (laser_mapping.cc:606-637):  p_w = R_f p + t_f (float);
 *   5-NN (:618) -> esti_plane (:621-622) -> pd2 = n.p_w + d (:627-629);
 *   valid iff ||p_body|| > 81 pd2^2 (:631).  Clean semantics:
 *   a point failing any test is dropped for this linearization.
 *   residual e = pd2; left-perturbation Jacobian (lsq_registration_impl.hpp:139-143
 *   update is delta * x0): de/d[w,v] = [(p_w x n)^T, n^T]; geometry in float,
 *   stored/accumulated in double like h_x (laser_mapping.cc:665,689-696;
 *   esekfom.hpp:1687).
 */
static double p2plane_linearize(oracle *o, const double T[16], double *H, double *b) {
  p2plane_prepare(o);
  float R[9], t[3];
  iso_to_float(T, R, t);
  const long n = o->src.n;
  const int K = o->cfg.knn;
  const int nth = orc_threads(o);
  double *Hs = (double *)calloc((size_t)nth * 43, sizeof(double));
  long *cnts = (long *)calloc((size_t)nth, sizeof(long));
#pragma omp parallel num_threads(nth)
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double *Ht = Hs + (size_t)tid * 43;
    orc_distpt_buf buf = {0, 0};
    int idx[ORC_QR_MAXR];
    float near[ORC_QR_MAXR * 3];
#pragma omp for schedule(static)
    for (long i = 0; i < n; i++) {
      const float *p = o->src.xyz + 3 * i;
      float q[3];
      for (int a = 0; a < 3; a++) q[a] = (R[a * 3 + 0] * p[0] + R[a * 3 + 1] * p[1]) + R[a * 3 + 2] * p[2] + t[a];
      int m = orc_ivox_knn(o, q, idx, NULL, &buf);
      float *pl = o->plane + 4 * i;
      int sel = m >= o->cfg.min_knn;
      if (sel) {
        for (int j = 0; j < m; j++) {
          near[j * 3 + 0] = o->tgt.xyz[3 * (long)idx[j] + 0];
          near[j * 3 + 1] = o->tgt.xyz[3 * (long)idx[j] + 1];
          near[j * 3 + 2] = o->tgt.xyz[3 * (long)idx[j] + 2];
        }
        sel = orc_esti_plane(near, m, K, o->cfg.min_knn, (float)o->cfg.plane_threshold, pl);
      }
      float pd2 = 0.f;
      if (sel) {
        pd2 = pl[0] * q[0] + pl[1] * q[1] + pl[2] * q[2] + pl[3];
        float pn = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        sel = pn > 81.f * pd2 * pd2;
      }
      o->selected[i] = (unsigned char)sel;
      if (!sel) continue;
      cnts[tid]++;
      float Jf[6];
      Jf[0] = q[1] * pl[2] - q[2] * pl[1];
      Jf[1] = q[2] * pl[0] - q[0] * pl[2];
      Jf[2] = q[0] * pl[1] - q[1] * pl[0];
      Jf[3] = pl[0]; Jf[4] = pl[1]; Jf[5] = pl[2];
      double J[6], e = (double)pd2;
      for (int a = 0; a < 6; a++) J[a] = (double)Jf[a];
      Ht[42] += e * e;
      for (int a = 0; a < 6; a++) {
        for (int c = 0; c < 6; c++) Ht[a * 6 + c] += J[a] * J[c];
        Ht[36 + a] += J[a] * e;
      }
    }
    free(buf.data);
  }
  double cost = 0.0;
  if (H) memset(H, 0, 36 * sizeof(double));
  if (b) memset(b, 0, 6 * sizeof(double));
  long cnt = 0;
  for (int tI = 0; tI < nth; tI++) { /* serial sum of per-thread partials  fast_gicp_impl.hpp:201-208 */
    const double *Ht = Hs + (size_t)tI * 43;
    if (H) for (int a = 0; a < 36; a++) H[a] += Ht[a];
    if (b) for (int a = 0; a < 6; a++) b[a] += Ht[36 + a];
    cost += Ht[42];
    cnt += cnts[tI];
  }
  o->num_inliers = (int)cnt;
  free(Hs);
  free(cnts);
  return cost;
}

/* compute_error: cost at a trial pose re-using the correspondences (here: the
 * selected set and fitted planes) of the last linearize -- same contract as
 * FastGICP::compute_error  fast_gicp_impl.hpp:213-237. */
static double p2plane_compute_error(oracle *o, const double T[16]) {
  float R[9], t[3];
  iso_to_float(T, R, t);
  const long n = o->src.n;
  double cost = 0.0;
  const int nth = orc_threads(o);
#pragma omp parallel for num_threads(nth) reduction(+ : cost) schedule(static)
  for (long i = 0; i < n; i++) {
    if (!o->selected[i]) continue;
    const float *p = o->src.xyz + 3 * i;
    const float *pl = o->plane + 4 * i;
    float q[3];
    for (int a = 0; a < 3; a++) q[a] = (R[a * 3 + 0] * p[0] + R[a * 3 + 1] * p[1]) + R[a * 3 + 2] * p[2] + t[a];
    float pd2 = pl[0] * q[0] + pl[1] * q[1] + pl[2] * q[2] + pl[3];
    cost += (double)pd2 * (double)pd2;
  }
  return cost;
}

/* ------------------------------------------------------------------------- */
/* jueying_lio ObsModel + IEKF reduction                                       */
/* ------------------------------------------------------------------------- */
/* Eigen::Quaternion product a * b, coefficients (x, y, z, w) */
static void quat_mul(const double a[4], const double b[4], double r[4]) {
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
/* Eigen QuaternionBase::_transformVector: uv = u x v; uv += uv; v + w*uv + u x uv */
#define ORC_DEF_QROT(NAME, T)                                                     \
  static void NAME(const T q[4], const T v[3], T r[3]) {                          \
    T uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]}; \
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];                               \
    const T c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]}; \
    for (int a = 0; a < 3; a++) r[a] = v[a] + q[3] * uv[a] + c[a];                \
  }
ORC_DEF_QROT(quat_rot_d, double)
ORC_DEF_QROT(quat_rot_f, float)
/* Eigen Quaternion::toRotationMatrix, coefficients (x, y, z, w) -> row-major 3x3 */
static void quat_to_rot_xyzw(const double q[4], double R[9]) {
  const double w[4] = {q[3], q[0], q[1], q[2]};
  orc_quat_to_rot(w, R);
}

/* residuals_.resize(cur_pts, 0); point_selected_surf_.resize(cur_pts, true); plane_coef_.resize(cur_pts, V4F::Zero())
 * laser_mapping.cc:337-339 -- std::vector::resize: the first min(old, new) entries survive, appended ones take the default */
static void lio_ref_resize(oracle *o, long n) {
  if (n > o->ref_cap) {
    long cap = n + n / 2 + 16;
    o->ref_plane = (float *)realloc(o->ref_plane, sizeof(float) * 4 * (size_t)cap);
    o->ref_resid = (float *)realloc(o->ref_resid, sizeof(float) * (size_t)cap);
    o->ref_sel = (unsigned char *)realloc(o->ref_sel, (size_t)cap);
    o->ref_cap = cap;
  }
  for (long i = o->ref_n; i < n; i++) {
    o->ref_plane[4 * i] = o->ref_plane[4 * i + 1] = o->ref_plane[4 * i + 2] = o->ref_plane[4 * i + 3] = 0.f;
    o->ref_resid[i] = 0.f;
    o->ref_sel[i] = 1;
  }
  o->ref_n = n;
}

void orc_set_neighbor_radius(void *h, double radius) {
  oracle *o = (oracle *)h;
  o->nb_radius = radius > 0.0 ? radius : 0.0;
}

void orc_set_lio_reference_semantics(void *h, int on) {
  oracle *o = (oracle *)h;
  o->lio_ref = on != 0;
  if (on) lio_ref_resize(o, o->src.n);
}

int orc_get_lio_members(void *h, float *plane4, float *resid, unsigned char *selected, long n) {
  oracle *o = (oracle *)h;
  if (!o->lio_ref || n != o->ref_n) return -1;
  memcpy(plane4, o->ref_plane, sizeof(float) * 4 * (size_t)n);
  memcpy(resid, o->ref_resid, sizeof(float) * (size_t)n);
  memcpy(selected, o->ref_sel, (size_t)n);
  return 0;
}

int orc_obs_model(void *h, const orc_lio_state *s, int extrinsic_est_en, int converge, double HTH[144], double HTh[12], int *n_eff, double *sum_h2) {
  oracle *o = (oracle *)h;
  p2plane_prepare(o);
  const int ref = o->lio_ref;
  if (ref) lio_ref_resize(o, o->src.n);
  /* R_wl = (s.rot * s.offset_R_L_I).cast<float>() ; t_wl = (s.rot * s.offset_T_L_I + s.pos).cast<float>()   :602-603 */
  double qwl[4], twl[3];
  quat_mul(s->rot, s->off_R, qwl);
  quat_rot_d(s->rot, s->off_T, twl);
  float qf[4], tf[3];
  for (int a = 0; a < 4; a++) qf[a] = (float)qwl[a];
  for (int a = 0; a < 3; a++) tf[a] = (float)(twl[a] + s->pos[a]);
  /* off_R, off_t, Rt as float matrices   :669-671 */
  double Rd[9], ORd[9];
  quat_to_rot_xyzw(s->rot, Rd);
  quat_to_rot_xyzw(s->off_R, ORd);
  float Rt[9], offR[9], offt[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rt[i * 3 + j] = (float)Rd[j * 3 + i]; offR[i * 3 + j] = (float)ORd[i * 3 + j]; }
  for (int a = 0; a < 3; a++) offt[a] = (float)s->off_T[a];
  const long n = o->src.n;
  const int K = o->cfg.knn;
  if (o->nn_cap < n) { free(o->nn); o->nn = (int *)malloc(sizeof(int) * 5 * (size_t)(n > 0 ? n : 1)); o->nn_cap = n; }
  double acc[92];
  memset(acc, 0, sizeof(acc));
  orc_distpt_buf buf = {0, 0};
  int idx[ORC_QR_MAXR];
  float near[ORC_QR_MAXR * 3];
  long cnt = 0;
  for (long i = 0; i < n; i++) {   /* serial: this hook is a checker, not the timed baseline */
    const float *p = o->src.xyz + 3 * i;
    float q[3];
    quat_rot_f(qf, p, q);
    for (int a = 0; a < 3; a++) q[a] = q[a] + tf[a];
    float *pl = ref ? o->ref_plane + 4 * i : o->plane + 4 * i;
    unsigned char *selp = ref ? o->ref_sel + i : o->selected + i;
    int sel;
    if (converge) {
      int m = orc_ivox_knn(o, q, idx, NULL, &buf);
      for (int j = 0; j < 5; j++) o->nn[i * 5 + j] = j < m ? idx[j] : -1;   /* nearest_points_[i] */
      sel = m >= o->cfg.min_knn;
      if (sel) {
        for (int j = 0; j < m; j++) for (int a = 0; a < 3; a++) near[j * 3 + a] = o->tgt.xyz[3 * (long)idx[j] + a];
        sel = orc_esti_plane(near, m, K, o->cfg.min_knn, (float)o->cfg.plane_threshold, pl);
      }
      *selp = (unsigned char)sel;   /* plane validity only; the residual test below is re-evaluated every call */
    } else {
      sel = *selp;
    }
    if (!sel) continue;
    float pd2 = pl[0] * q[0] + pl[1] * q[1] + pl[2] * q[2] + pl[3];
    const float pn = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    if (ref) {
      /* :631-635  the flag stays true either way; a failing point keeps the residual an earlier call stored at index i */
      if (pn > 81.f * pd2 * pd2) o->ref_resid[i] = pd2;
      pd2 = o->ref_resid[i];   /* corr_pts_[.][3] = residuals_[i]  :649 */
    } else if (!(pn > 81.f * pd2 * pd2)) {
      continue;   /* clean semantics: dropped for this call */
    }
    /* Jacobian row   :674-698 */
    float pthis[3];
    for (int a = 0; a < 3; a++) pthis[a] = (offR[a * 3 + 0] * p[0] + offR[a * 3 + 1] * p[1]) + offR[a * 3 + 2] * p[2] + offt[a];
    float C[3];
    for (int a = 0; a < 3; a++) C[a] = (Rt[a * 3 + 0] * pl[0] + Rt[a * 3 + 1] * pl[1]) + Rt[a * 3 + 2] * pl[2];
    /* A = skew(point_this) * C as a matrix-vector product (0*c0 first) */
    const float A[3] = {(0.f * C[0] + -pthis[2] * C[1]) + pthis[1] * C[2], (pthis[2] * C[0] + 0.f * C[1]) + -pthis[0] * C[2], (-pthis[1] * C[0] + pthis[0] * C[1]) + 0.f * C[2]};
    float row[12] = {pl[0], pl[1], pl[2], A[0], A[1], A[2], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (extrinsic_est_en) {
      /* B = (skew(p_be) * off_R^T) * C */
      const float S[9] = {0.f, -p[2], p[1], p[2], 0.f, -p[0], -p[1], p[0], 0.f};
      float SM[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) SM[a * 3 + b] = (S[a * 3 + 0] * offR[b * 3 + 0] + S[a * 3 + 1] * offR[b * 3 + 1]) + S[a * 3 + 2] * offR[b * 3 + 2];
      for (int a = 0; a < 3; a++) row[6 + a] = (SM[a * 3 + 0] * C[0] + SM[a * 3 + 1] * C[1]) + SM[a * 3 + 2] * C[2];
      for (int a = 0; a < 3; a++) row[9 + a] = C[a];
    }
    const double hh = -(double)pd2;   /* ekfom_data.h(i) = -residual  :697 */
    int t = 0;
    for (int a = 0; a < 12; a++) for (int b = a; b < 12; b++) acc[t++] += (double)row[a] * (double)row[b];
    for (int a = 0; a < 12; a++) acc[78 + a] += (double)row[a] * hh;
    acc[90] += hh * hh;
    cnt++;
  }
  free(buf.data);
  int t = 0;
  for (int a = 0; a < 12; a++) for (int b = a; b < 12; b++) { HTH[a * 12 + b] = acc[t]; HTH[b * 12 + a] = acc[t]; t++; }
  for (int a = 0; a < 12; a++) HTh[a] = acc[78 + a];
  if (sum_h2) *sum_h2 = acc[90];
  if (n_eff) *n_eff = (int)cnt;
  o->num_inliers = (int)cnt;
  return cnt > 0 ? 0 : -1;
}

/* ------------------------------------------------------------------------- */
/* model dispatch                                                             */
/* ------------------------------------------------------------------------- */
static double model_linearize(oracle *o, const double T[16], double *H, double *b) {
  double Hl[36], bl[6];
  double cost;
  o->num_linearize++;
  switch (o->cfg.model) {
    case ORC_MODEL_P2PLANE: cost = p2plane_linearize(o, T, Hl, bl); break;
    default: cost = orc_gauss_linearize(o, T, Hl, bl); break;
  }
  if (o->trace && o->trace_n < o->trace_max) {
    double *r = o->trace + (size_t)o->trace_n * 43;
    r[0] = cost;
    memcpy(r + 1, Hl, sizeof(Hl));
    memcpy(r + 37, bl, sizeof(bl));
    o->trace_n++;
  }
  if (H) memcpy(H, Hl, sizeof(Hl));
  if (b) memcpy(b, bl, sizeof(bl));
  return cost;
}

static double model_compute_error(oracle *o, const double T[16]) {
  double cost;
  o->num_compute_error++;
  switch (o->cfg.model) {
    case ORC_MODEL_P2PLANE: cost = p2plane_compute_error(o, T); break;
    default: cost = orc_gauss_compute_error(o, T); break;
  }
  if (o->trace && o->trace_n < o->trace_max) {   /* trial record: cost, then NaN in the H / b slots */
    double *r = o->trace + (size_t)o->trace_n * 43;
    r[0] = cost;
    for (int k = 1; k < 43; k++) r[k] = NAN;
    o->trace_n++;
  }
  return cost;
}

/* ------------------------------------------------------------------------- */
/* LsqRegistration  lsq_registration_impl.hpp                                 */
/* ------------------------------------------------------------------------- */
/* :81-91 */
static int is_converged(const oracle *o, const double delta[16]) {
  double rmax = 0.0, tmax = 0.0;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      double r = fabs(delta[i * 4 + j] - (i == j ? 1.0 : 0.0)) * (1.0 / o->cfg.rotation_eps);
      if (r > rmax) rmax = r;
    }
    double tt = fabs(delta[i * 4 + 3]) * (1.0 / o->cfg.translation_eps);
    if (tt > tmax) tmax = tt;
  }
  return (rmax > tmax ? rmax : tmax) < 1;
}

/* :105-122 */
static int step_gn(oracle *o, double x0[16], double delta[16]) {
  double H[36], b[6], nb[6], d[6];
  o->last_cost = model_linearize(o, x0, H, b);
  for (int i = 0; i < 6; i++) nb[i] = -b[i];
  orc_ldlt6_solve(H, nb, d);
  orc_delta_from_d(d, delta);
  orc_iso_mul(delta, x0, x0);
  memcpy(o->final_hessian, H, sizeof(H));
  return 1;
}

/* :124-172 */
static int step_lm(oracle *o, double x0[16], double delta[16]) {
  double H[36], b[6];
  double y0 = model_linearize(o, x0, H, b);
  o->last_cost = y0;
  if (o->lm_lambda < 0.0) {
    double mx = 0.0;
    for (int i = 0; i < 6; i++) if (fabs(H[i * 6 + i]) > mx) mx = fabs(H[i * 6 + i]);
    o->lm_lambda = o->cfg.lm_init_lambda_factor * mx;
  }
  double nu = 2.0;
  for (int i = 0; i < o->cfg.lm_max_iterations; i++) {
    double A[36], nb[6], d[6], xi[16];
    memcpy(A, H, sizeof(A));
    for (int k = 0; k < 6; k++) { A[k * 6 + k] += o->lm_lambda; nb[k] = -b[k]; }
    orc_ldlt6_solve(A, nb, d);
    orc_delta_from_d(d, delta);
    orc_iso_mul(delta, x0, xi);
    double yi = model_compute_error(o, xi);
    double dp[6];
    for (int k = 0; k < 6; k++) dp[k] = d[k] * (o->lm_lambda * d[k] - b[k]);
    const double den = orc_redux_fixed_d(dp, 6);   /* d.dot(lm_lambda_ * d - b)  :146: fixed size 6, Packet2d ([CORE-1], orc_eigen.h) */
    double rho = (y0 - yi) / den;
    if (rho < 0) {
      if (is_converged(o, delta)) return 1;
      o->lm_lambda = nu * o->lm_lambda;
      nu = 2 * nu;
      continue;
    }
    memcpy(x0, xi, sizeof(xi));
    double f = 1 - pow(2 * rho - 1, 3);   /* std::pow(2 * rho - 1, 3): libm, as the reference calls it (lsq_registration_impl.hpp:166) */
    o->lm_lambda = o->lm_lambda * (f > 1.0 / 3.0 ? f : 1.0 / 3.0);
    memcpy(o->final_hessian, H, sizeof(H));
    return 1;
  }
  return 0;
}

/* :52-79 */
int orc_align(void *h, const float guess[16], orc_result *out) {
  oracle *o = (oracle *)h;
  if (o->cfg.model == ORC_MODEL_NDT_OMP) return orc_pclndt_align(o, guess, out);   /* orc_pclndt.c */
  if (o->src.n <= 0 || o->tgt.n <= 0) return -1;
  double x0[16];
  for (int i = 0; i < 16; i++) x0[i] = (double)guess[i];
  o->lm_lambda = -1.0;
  o->num_linearize = o->num_compute_error = 0;
  int converged = 0, nr_iterations = 0;
  orc_prepare_model(o);
  for (int i = 0; i < o->cfg.max_iterations && !converged; i++) {
    nr_iterations = i;
    double delta[16];
    int ok = (o->cfg.optimizer == ORC_OPT_GN) ? step_gn(o, x0, delta) : step_lm(o, x0, delta);
    if (!ok) break; /* "lm not converged!!" :69-72 */
    converged = is_converged(o, delta);
  }
  if (out) {
    for (int i = 0; i < 16; i++) { out->T[i] = (float)x0[i]; out->T64[i] = x0[i]; }
    memcpy(out->H, o->final_hessian, sizeof(out->H));
    out->cost = o->last_cost;
    out->iterations = nr_iterations;
    out->converged = converged;
    out->num_linearize = o->num_linearize;
    out->num_compute_error = o->num_compute_error;
    out->num_inliers = o->num_inliers;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* object lifetime / setters                                                  */
/* ------------------------------------------------------------------------- */
void *orc_create(const orc_config *c) {
  oracle *o = (oracle *)calloc(1, sizeof(oracle));
  o->cfg = *c;
  o->lm_lambda = -1.0;
  for (int i = 0; i < 6; i++) o->final_hessian[i * 6 + i] = 1.0; /* :21 setIdentity */
  return o;
}

void orc_destroy(void *h) {
  oracle *o = (oracle *)h;
  if (!o) return;
  free(o->src.xyz);
  free(o->tgt.xyz);
  ivox_free(&o->tgt_ivox);
  free(o->plane);
  free(o->selected);
  free(o->ref_plane);
  free(o->ref_resid);
  free(o->ref_sel);
  free(o->nn);
  orc_lru_free(o);
  orc_gauss_free(o);
  orc_gicp_free(o);
  orc_pclndt_free(o);
  free(o);
}

int orc_set_target(void *h, const float *xyz, long n, long stride) {
  oracle *o = (oracle *)h;
  cloud_set(&o->tgt, xyz, n, stride);
  o->tgt_ivox.valid = 0;
  orc_lru_reset(o);
  orc_gauss_invalidate(o, 1);
  orc_gicp_invalidate(o, 1);
  orc_pclndt_invalidate(o);
  return 0;
}

int orc_set_source(void *h, const float *xyz, long n, long stride) {
  oracle *o = (oracle *)h;
  cloud_set(&o->src, xyz, n, stride);
  if (o->lio_ref) lio_ref_resize(o, n);   /* one resize per frame  laser_mapping.cc:335-339 */
  orc_gauss_invalidate(o, 0);
  orc_gicp_invalidate(o, 0);
  return 0;
}

/* FastGICP::swapSourceAndTarget  fast_gicp_impl.hpp:50-58 */
void orc_swap_source_and_target(void *h) {
  oracle *o = (oracle *)h;
  orc_cloud t = o->src; o->src = o->tgt; o->tgt = t;
  o->tgt_ivox.valid = 0;
  orc_lru_reset(o);
  orc_gauss_swap(o);
  orc_gicp_swap(o);
  orc_pclndt_invalidate(o);
}

double orc_linearize(void *h, const double T[16], double H[36], double b[6]) {
  oracle *o = (oracle *)h;
  orc_prepare_model(o);
  return model_linearize(o, T, H, b);
}

double orc_compute_error(void *h, const double T[16]) {
  oracle *o = (oracle *)h;
  return model_compute_error(o, T);
}

int orc_get_planes(void *h, float *planes, unsigned char *selected, long n) {
  oracle *o = (oracle *)h;
  if (n != o->src.n || !o->plane) return -1;
  memcpy(planes, o->plane, sizeof(float) * 4 * (size_t)n);
  memcpy(selected, o->selected, (size_t)n);
  return 0;
}

int orc_num_inliers(void *h) { return ((oracle *)h)->num_inliers; }

void orc_set_trace(void *h, double *buf, int max_records) {
  oracle *o = (oracle *)h;
  o->trace = buf;
  o->trace_max = max_records;
  o->trace_n = 0;
}

int orc_trace_count(void *h) { return ((oracle *)h)->trace_n; }

void orc_prepare_model(oracle *o) {
  if (o->cfg.model == ORC_MODEL_P2PLANE) p2plane_prepare(o);
  else if (o->cfg.model == ORC_MODEL_GICP || o->cfg.model == ORC_MODEL_VGICP) orc_gicp_prepare(o);
  else orc_gauss_prepare(o);
}

/* ------------------------------------------------------------------------- */
/* unit hooks                                                                 */
/* ------------------------------------------------------------------------- */
void orc_test_so3_exp(const double omega[3], double R[9]) {
  double q[4];
  orc_so3_exp(omega, q);
  orc_quat_to_rot(q, R);
}

void orc_test_ldlt6_solve(const double A[36], const double b[6], double x[6]) { orc_ldlt6_solve(A, b, x); }

int orc_test_esti_plane(const float *pts_xyz, int n, float threshold, float plane[4]) {
  return orc_esti_plane(pts_xyz, n, 5, 3, threshold, plane);
}

int orc_test_knn(void *h, const float q[3], int *idx_out, float *d2_out) {
  oracle *o = (oracle *)h;
  p2plane_prepare(o);
  orc_distpt_buf buf = {0, 0};
  int m = orc_ivox_knn(o, q, idx_out, d2_out, &buf);
  free(buf.data);
  return m;
}

long orc_test_voxel_key(void *h, const float p[3], int key[3]) {
  oracle *o = (oracle *)h;
  orc_ivox_key(o, p, key);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* batch hooks for the Eigen restatements of orc_eigen.h (tests/test_eigen_restatements.py checks every one of them  */
/* against float64 numpy on >= 10 000 random cases)                                                                   */
/* ------------------------------------------------------------------------- */
void orc_test_eig_ldlt6(long n, const double *A, const double *b, double *x) {
  for (long i = 0; i < n; i++) orc_eig_ldlt6_solve(A + 36 * i, b + 6 * i, x + 6 * i);
}
void orc_test_eig_colpivqr_f(long n, int rows, const float *A, const float *b, float *x) {
  for (long i = 0; i < n; i++) orc_eig_colpivqr3f(A + (long)rows * 3 * i, rows, b + (long)rows * i, x + 3 * i);
}
void orc_test_eig_colpivqr_d(long n, int rows, const double *A, const double *b, double *x) {
  for (long i = 0; i < n; i++) orc_eig_colpivqr3d(A + (long)rows * 3 * i, rows, b + (long)rows * i, x + 3 * i);
}
void orc_test_eig_jacobi_svd(long n, int dim, const double *A, double *U, double *S, double *V) {
  for (long i = 0; i < n; i++) orc_eig_jacobi_svd(dim, A + (long)dim * dim * i, U + (long)dim * dim * i, S + (long)dim * i, V + (long)dim * dim * i);
}
void orc_test_eig_svd_solve6(long n, const double *A, const double *b, double *x) {
  for (long i = 0; i < n; i++) orc_eig_svd_solve6(A + 36 * i, b + 6 * i, x + 6 * i);
}
void orc_test_eig_selfadjoint3(long n, const double *A, double *w, double *V, int *ok) {
  for (long i = 0; i < n; i++) ok[i] = orc_eig_selfadjoint3(A + 9 * i, w + 3 * i, V + 9 * i);
}
void orc_test_eig_direct3f(long n, const float *A, float *w, float *V) {
  for (long i = 0; i < n; i++) orc_eig_direct3f(A + 9 * i, w + 3 * i, V + 9 * i);
}
void orc_test_eig_inv3d(long n, const double *A, double *R) { for (long i = 0; i < n; i++) orc_eig_inv3d(A + 9 * i, R + 9 * i); }
void orc_test_eig_inv3f(long n, const float *A, float *R) { for (long i = 0; i < n; i++) orc_eig_inv3f(A + 9 * i, R + 9 * i); }
void orc_test_eig_inv4d(long n, const double *A, double *R) { for (long i = 0; i < n; i++) orc_eig_inv4d(A + 16 * i, R + 16 * i); }

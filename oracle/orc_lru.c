/*
 * oracle/orc_lru.c -- sliding submap of the CPU oracle (TEST INFRASTRUCTURE ONLY):
 *   IVox::AddPoints with the LRU voxel cache ..... /root/reference/src/jueying_lio/include/ivox3d/ivox3d.h:256-281
 *   LaserMapping::MapIncremental (add-filter) ..... /root/reference/src/jueying_lio/src/laser_mapping.cc:525-583
 *   LaserMapping::PointBodyToWorld ................ /root/reference/src/jueying_lio/src/laser_mapping.cc:855-864
 * Exact sequential semantics of the reference's std::list + unordered_map: a new voxel goes to the
 * front, an existing one is spliced to the front on every insertion, and the back voxel is erased
 * whenever the map size reaches `capacity` after a creation.
 */
#include "orc_internal.h"

#include <stdlib.h>

typedef struct lru_vox {
  int alive;
  long prev, next;      /* recency list, -1 = none */
  long npts;            /* points currently in the voxel */
} lru_vox;

typedef struct orc_lru_state {
  orc_vhash h;          /* key -> voxel id (ids are never recycled; `alive` says whether the voxel exists) */
  lru_vox *vox;
  long nvox_ids, vox_cap;
  long head, tail, count;
  float *xyz;           /* master point array, insertion order */
  int *pvox;            /* voxel id of every point */
  unsigned char *dead;
  long np, pcap;
  long hash_cap_pts;    /* size hint the hash was created for */
} orc_lru_state;

static void list_unlink(orc_lru_state *L, long v) {
  lru_vox *x = &L->vox[v];
  if (x->prev >= 0) L->vox[x->prev].next = x->next; else L->head = x->next;
  if (x->next >= 0) L->vox[x->next].prev = x->prev; else L->tail = x->prev;
  x->prev = x->next = -1;
}

static void list_push_front(orc_lru_state *L, long v) {
  lru_vox *x = &L->vox[v];
  x->prev = -1;
  x->next = L->head;
  if (L->head >= 0) L->vox[L->head].prev = v; else L->tail = v;
  L->head = v;
}

void orc_lru_free(oracle *o) {
  orc_lru_state *L = o->lru;
  if (!L) return;
  orc_vhash_free(&L->h);
  free(L->vox); free(L->xyz); free(L->pvox); free(L->dead);
  free(L);
  o->lru = NULL;
}

void orc_lru_reset(oracle *o) { orc_lru_free(o); }

static void lru_insert_one(oracle *o, orc_lru_state *L, const float p[3]) {
  int k[3];
  orc_ivox_key(o, p, k);
  if (L->h.count * 2 + 2 > L->h.cap) {   /* grow + rehash (ids are stable: re-insert in id order) */
    orc_vhash nh;
    orc_vhash_init(&nh, L->h.cap);
    int *keys = (int *)malloc(sizeof(int) * 3 * (size_t)(L->nvox_ids + 1));
    for (long s = 0; s < L->h.cap; s++) if (L->h.val[s] >= 0) memcpy(keys + 3 * L->h.val[s], L->h.keys + 3 * s, 3 * sizeof(int));
    for (long id = 0; id < L->nvox_ids; id++) orc_vhash_insert(&nh, keys[3 * id], keys[3 * id + 1], keys[3 * id + 2]);
    free(keys);
    orc_vhash_free(&L->h);
    L->h = nh;
  }
  const long v = orc_vhash_insert(&L->h, k[0], k[1], k[2]);
  if (v >= L->vox_cap) {
    L->vox_cap = L->vox_cap ? L->vox_cap * 2 : 1024;
    L->vox = (lru_vox *)realloc(L->vox, sizeof(lru_vox) * (size_t)L->vox_cap);
  }
  if (v >= L->nvox_ids) { L->vox[v].alive = 0; L->vox[v].prev = L->vox[v].next = -1; L->vox[v].npts = 0; L->nvox_ids = v + 1; }
  if (L->np >= L->pcap) {
    L->pcap = L->pcap ? L->pcap * 2 : 4096;
    L->xyz = (float *)realloc(L->xyz, sizeof(float) * 3 * (size_t)L->pcap);
    L->pvox = (int *)realloc(L->pvox, sizeof(int) * (size_t)L->pcap);
    L->dead = (unsigned char *)realloc(L->dead, (size_t)L->pcap);
  }
  memcpy(L->xyz + 3 * L->np, p, 3 * sizeof(float));
  L->pvox[L->np] = (int)v;
  L->dead[L->np] = 0;
  L->np++;
  if (!L->vox[v].alive) {                    /* iter == grids_map_.end(): create at the front   ivox3d.h:261-268 */
    L->vox[v].alive = 1;
    L->vox[v].npts = 1;
    list_push_front(L, v);
    L->count++;
    const long cap = o->cfg.map_capacity;
    if (cap > 0 && L->count >= cap) {        /* grids_map_.size() >= capacity_: erase the back   :270-273 */
      const long ev = L->tail;
      list_unlink(L, ev);
      L->vox[ev].alive = 0;
      L->vox[ev].npts = 0;
      L->count--;
      for (long i = 0; i < L->np; i++) if (L->pvox[i] == ev) L->dead[i] = 1;   /* O(N) per eviction: fine for a checker */
    }
  } else {                                   /* existing voxel: insert, splice to the front   :274-278 */
    L->vox[v].npts++;
    list_unlink(L, v);
    list_push_front(L, v);
  }
}

static orc_lru_state *lru_get(oracle *o) {
  if (o->lru) return o->lru;
  orc_lru_state *L = (orc_lru_state *)calloc(1, sizeof(orc_lru_state));
  L->head = L->tail = -1;
  orc_vhash_init(&L->h, 1024);
  o->lru = L;
  /* the map that exists so far was itself built by AddPoints (laser_mapping.cc:314-319): replay it */
  for (long i = 0; i < o->tgt.n; i++) lru_insert_one(o, L, o->tgt.xyz + 3 * i);
  return L;
}

/* after a batch: the target cloud = surviving points in insertion order */
static void lru_commit(oracle *o, orc_lru_state *L) {
  long w = 0;
  for (long i = 0; i < L->np; i++) {
    if (L->dead[i]) continue;
    if (w != i) { memcpy(L->xyz + 3 * w, L->xyz + 3 * i, 3 * sizeof(float)); L->pvox[w] = L->pvox[i]; L->dead[w] = 0; }
    w++;
  }
  L->np = w;
  free(o->tgt.xyz);
  o->tgt.n = w;
  o->tgt.xyz = (float *)malloc(sizeof(float) * 3 * (size_t)(w > 0 ? w : 1));
  memcpy(o->tgt.xyz, L->xyz, sizeof(float) * 3 * (size_t)w);
  o->tgt_ivox.valid = 0;
  orc_gauss_invalidate(o, 1);
}

int orc_target_insert(void *h, const float *xyz, long n, long stride) {
  oracle *o = (oracle *)h;
  orc_lru_state *L = lru_get(o);
  for (long i = 0; i < n; i++) lru_insert_one(o, L, xyz + i * stride);
  lru_commit(o, L);
  return 0;
}

long orc_target_size(void *h) { return ((oracle *)h)->tgt.n; }

long orc_target_voxels(void *h) {
  oracle *o = (oracle *)h;
  return lru_get(o)->count;
}

void orc_get_target(void *h, float *out_xyz) {
  oracle *o = (oracle *)h;
  memcpy(out_xyz, o->tgt.xyz, sizeof(float) * 3 * (size_t)o->tgt.n);
}

/* Eigen _transformVector in double */
static void qrot(const double q[4], const double v[3], double r[3]) {
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int a = 0; a < 3; a++) r[a] = v[a] + q[3] * uv[a] + c[a];
}

int orc_map_incremental(void *h, const orc_lio_state *s, double filter_size_map, int ekf_inited, long *n_added) {
  oracle *o = (oracle *)h;
  const long n = o->src.n;
  const float fs = (float)filter_size_map;
  float *add = (float *)malloc(sizeof(float) * 3 * (size_t)(n + 1)), *noneed = (float *)malloc(sizeof(float) * 3 * (size_t)(n + 1));
  long na = 0, nn = 0;
  const int have_nn = o->nn != NULL && o->nn_cap >= n && ekf_inited;
  for (long i = 0; i < n; i++) {
    const float *pb = o->src.xyz + 3 * i;
    const double vb[3] = {pb[0], pb[1], pb[2]};
    double v1[3], v2[3];
    qrot(s->off_R, vb, v1);                                  /* PointBodyToWorld  laser_mapping.cc:855-864 */
    for (int a = 0; a < 3; a++) v1[a] += s->off_T[a];
    qrot(s->rot, v1, v2);
    const float pw[3] = {(float)(v2[0] + s->pos[0]), (float)(v2[1] + s->pos[1]), (float)(v2[2] + s->pos[2])};
    if (have_nn && o->nn[i * 5] >= 0) {                      /* !nearest_points_[i].empty() && flg_EKF_inited_ */
      float center[3];
      for (int a = 0; a < 3; a++) center[a] = (floorf(pw[a] / fs) + 0.5f) * fs;                      /* :547-548 */
      const float *n0 = o->tgt.xyz + 3 * (long)o->nn[i * 5];
      const double half = 0.5 * (double)fs;
      if ((double)fabsf(n0[0] - center[0]) > half && (double)fabsf(n0[1] - center[1]) > half && (double)fabsf(n0[2] - center[2]) > half) {
        memcpy(noneed + 3 * nn, pw, sizeof(pw)); nn++;      /* :552-557 */
        continue;
      }
      const float dx = pw[0] - center[0], dy = pw[1] - center[1], dz = pw[2] - center[2];
      const float dist = dx * dx + dy * dy + dz * dz;
      int need_add = 1;
      if (o->nn[i * 5 + 4] >= 0) {                           /* points_near.size() >= NUM_MATCH_POINTS */
        for (int k = 0; k < 5; k++) {
          const float *q = o->tgt.xyz + 3 * (long)o->nn[i * 5 + k];
          const float ex = q[0] - center[0], ey = q[1] - center[1], ez = q[2] - center[2];
          if ((double)(ex * ex + ey * ey + ez * ez) < (double)dist + 1e-6) { need_add = 0; break; }   /* :563-566 */
        }
      }
      if (need_add) { memcpy(add + 3 * na, pw, sizeof(pw)); na++; }
    } else {
      memcpy(add + 3 * na, pw, sizeof(pw)); na++;
    }
  }
  orc_lru_state *L = lru_get(o);
  for (long i = 0; i < na; i++) lru_insert_one(o, L, add + 3 * i);          /* ivox_->AddPoints(points_to_add) */
  for (long i = 0; i < nn; i++) lru_insert_one(o, L, noneed + 3 * i);       /* ivox_->AddPoints(point_no_need_downsample) */
  lru_commit(o, L);
  free(add); free(noneed);
  if (n_added) *n_added = na + nn;
  if (o->nn) for (long i = 0; i < n; i++) o->nn[i * 5] = -1;   /* stale after the map changed */
  return 0;
}

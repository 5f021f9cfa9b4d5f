/*
 * oracle/orc_internal.h -- shared state of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "pcm_oracle.h"
#include "orc_linalg.h"
#include <stddef.h>

typedef struct orc_cloud { float *xyz; long n; } orc_cloud;

typedef struct orc_vhash { int *keys; int *val; long cap; long count; } orc_vhash;

typedef struct orc_ivox {
  orc_vhash h;
  long nvox;
  int *vox_start; /* nvox+1 */
  int *vox_pts;   /* target point indices grouped by voxel, insertion order */
  int valid;
} orc_ivox;

typedef struct orc_distpt_buf { void *data; size_t bytes; } orc_distpt_buf;

struct orc_gauss_state;
struct orc_lru_state;

typedef struct oracle {
  orc_config cfg;
  orc_cloud src, tgt;
  orc_ivox tgt_ivox;
  float *plane;            /* 4 per source point (last linearize) */
  unsigned char *selected; /* per source point */
  long plane_cap;
  int num_inliers;
  /* LsqRegistration state */
  double lm_lambda;
  double final_hessian[36];
  double last_cost;
  int num_linearize, num_compute_error;
  /* trace */
  double *trace;
  int trace_max, trace_n;
  struct orc_gauss_state *gauss;
  void *gicp;                  /* GICP / VGICP state (orc_gicp.c) */
  void *pclndt;                /* pclomp NDT state (orc_pclndt.c) */
  struct orc_lru_state *lru;   /* sliding-map state (orc_lru.c) */
  int *nn;                     /* [n_src][5] target indices of the last matching call, -1 = none */
  long nn_cap;
  /* LaserMapping members that outlive one ObsModel call and one frame (laser_mapping.cc:335-339): kept only
   * in the reference-semantics mode of orc_obs_model */
  double nb_radius;            /* > 0: NeighborSearchMethod::DIRECT_RADIUS (orc_set_neighbor_radius) */
  int lio_ref;                 /* orc_set_lio_reference_semantics */
  int knn_order;               /* orc_set_knn_order: ORC_KNN_ORDER_* */
  float *ref_plane;            /* plane_coef_           [ref_n][4] */
  float *ref_resid;            /* residuals_            [ref_n]    */
  unsigned char *ref_sel;      /* point_selected_surf_  [ref_n]    */
  long ref_n, ref_cap;
} oracle;

void orc_vhash_init(orc_vhash *h, long expected);
void orc_vhash_free(orc_vhash *h);
int orc_vhash_find(const orc_vhash *h, int x, int y, int z);
int orc_vhash_insert(orc_vhash *h, int x, int y, int z);
void orc_ivox_key(const oracle *o, const float p[3], int key[3]);
int orc_ivox_knn(const oracle *o, const float q[3], int *idx_out, float *d2_out, orc_distpt_buf *buf);
int orc_esti_plane(const float *pts, int n, int K, int min_pts, float threshold, float plane[4]);
void orc_prepare_model(oracle *o);
void orc_lru_free(oracle *o);
void orc_lru_reset(oracle *o);

/* orc_models_gauss.c: GICP / VGICP / NDT residual models */
double orc_gauss_linearize(oracle *o, const double T[16], double *H, double *b);
double orc_gauss_compute_error(oracle *o, const double T[16]);
void orc_gauss_prepare(oracle *o);
void orc_gauss_invalidate(oracle *o, int target);
void orc_gauss_swap(oracle *o);
void orc_gauss_free(oracle *o);

/* orc_pclndt.c */
int orc_pclndt_align(oracle *o, const float guess[16], orc_result *out);
void orc_pclndt_invalidate(oracle *o);
void orc_pclndt_free(oracle *o);

/* orc_gicp.c */
double orc_gicp_linearize(oracle *o, const double T[16], double *H, double *b);
double orc_gicp_compute_error(oracle *o, const double T[16]);
void orc_gicp_prepare(oracle *o);
void orc_gicp_invalidate(oracle *o, int target);
void orc_gicp_swap(oracle *o);
void orc_gicp_free(oracle *o);
void orc_calc_covariances(const oracle *o, const orc_cloud *c, double *covs);
void orc_calc_covariances_f(const oracle *o, const orc_cloud *c, float *covs);   /* CUDA-core float semantics */

#endif

/*
 * oracle/pcm_oracle.h -- C API of the CPU oracle (ctypes-loaded by tests/ and bench.py).
 *
 * TEST INFRASTRUCTURE ONLY: the oracle is the parity checker and the reported
 * CPU baseline.  The product path (pointcloud-slam_amd/) never includes, links
 * or loads anything from oracle/.
 *
 * PARITY PIN STATUS ("parity unpinned" by the reference's own fixtures): the
 * reference's hot path cannot be compiled here (needs PCL, FLANN, Boost, a
 * complete Eigen) and its only known-answer fixture
 * (fast_gicp/data/relative.txt) refers to two .pcd files that are not in the
 * tree.  The oracle is therefore a line-by-line restatement self-pinned by
 * analytic known-answer tests (tests/test_oracle_*.py) and by golden vectors
 * it generated itself (tests/golden/, script committed).
 *
 * All 4x4 transforms crossing this API are ROW-MAJOR.
 */
#ifndef PCM_ORACLE_H
#define PCM_ORACLE_H

#ifdef __cplusplus
extern "C" {
/* pcl::Registration::getFitnessScore(max_range) of the source cloud under T (orc_gicp.c) */
double orc_fitness_score(void *h, const float T[16], double max_range);

#endif

enum { ORC_MODEL_P2PLANE = 0, ORC_MODEL_GICP = 1, ORC_MODEL_VGICP = 2, ORC_MODEL_NDT_P2D = 3, ORC_MODEL_NDT_D2D = 4, ORC_MODEL_NDT_OMP = 5, ORC_MODEL_VGICP_CUDA = 6 };
enum { ORC_OPT_GN = 0, ORC_OPT_LM = 1 };
enum { ORC_REG_NONE = 0, ORC_REG_MIN_EIG = 1, ORC_REG_NORMALIZED_MIN_EIG = 2, ORC_REG_PLANE = 3, ORC_REG_FROBENIUS = 4,
       ORC_REG_PCLOMP = 5 /* pclomp::GeneralizedIterativeClosestPoint::computeCovariances (gicp_omp_impl.hpp:48-122) */ };

typedef struct orc_config {
  int model;
  int optimizer;              /* lsq_registration_impl.hpp:15 default LM */
  int max_iterations;         /* :11  64 */
  double rotation_eps;        /* :12  2e-3 */
  double translation_eps;     /* :13  5e-4 */
  int lm_max_iterations;      /* :17  10 */
  double lm_init_lambda_factor; /* :18 1e-9 */
  double voxel_resolution;    /* ivox3d.h:54 0.2 | fast_vgicp_impl.hpp:22 1.0 */
  int num_neighbors;          /* 1, 7, 19 or 27 */
  int knn;                    /* options.h:14 NUM_MATCH_POINTS 5 */
  int min_knn;                /* options.h:15 MIN_NUM_MATCH_POINTS 3 */
  double max_range;           /* ivox3d.h:80 max_range 5.0 */
  double plane_threshold;     /* options.cc:10 0.1 */
  double max_corr_dist;       /* fast_gicp_impl.hpp:18 FLT_MAX (GICP) */
  int k_correspondences;      /* fast_gicp_impl.hpp:16 20 */
  int regularization;         /* fast_gicp_impl.hpp:20 PLANE */
  int num_threads;
  long map_capacity;          /* ivox3d.h:57 capacity_ 1000000 (0 = unlimited) */
  double ndt_step_size;       /* ndt_omp_impl.hpp:48 step_size_ 0.1 (More-Thuente maximum step) */
  double ndt_outlier_ratio;   /* ndt_omp_impl.hpp:48 outlier_ratio_ 0.55 */
  int voxel_mode;             /* VGICP VoxelAccumulationMode: 0 ADDITIVE (fast_vgicp_impl.hpp:25), 1 ADDITIVE_WEIGHTED, 2 MULTIPLICATIVE */
  double rbf_kernel_width;    /* VGICP_CUDA: > 0 selects NearestNeighborMethod::GPU_RBF_KERNEL (cuda/covariance_estimation_rbf.cu); fast_vgicp_cuda.cu:25 0.25 */
  double rbf_max_dist;        /* fast_vgicp_cuda.cu:26 3.0 */
} orc_config;

typedef struct orc_result {
  float T[16];               /* final_transformation_ = x0.cast<float>() (row-major) */
  double T64[16];            /* x0 before the float cast */
  double H[36];              /* final_hessian_ */
  double cost;               /* last linearize() cost */
  int iterations;            /* nr_iterations_ (index of last outer iteration) */
  int converged;
  int num_linearize;         /* passes actually executed (for byte accounting) */
  int num_compute_error;
  int num_inliers;           /* correspondences used by the last linearize */
} orc_result;

void orc_default_config(orc_config *c);
void *orc_create(const orc_config *c);
void orc_destroy(void *h);
int orc_set_target(void *h, const float *xyz, long n, long stride_floats);
int orc_set_source(void *h, const float *xyz, long n, long stride_floats);
void orc_swap_source_and_target(void *h);
double orc_linearize(void *h, const double T[16], double H[36], double b[6]);
double orc_compute_error(void *h, const double T[16]);
int orc_num_inliers(void *h);
/* planes (4 floats) + selected flag of every source point from the last linearize */
int orc_get_planes(void *h, float *planes, unsigned char *selected, long n);
int orc_align(void *h, const float guess[16], orc_result *out);
/* per-iteration trace: each linearize() appends 1+36+6 doubles (cost,H,b) */
void orc_set_trace(void *h, double *buf, int max_records);
int orc_trace_count(void *h);

/* jueying_lio measurement model (IEKF h_dyn_share callback): state as in state_ikfom,
 * quaternions in Eigen coefficient order (x, y, z, w). */
typedef struct orc_lio_state {
  double rot[4];     /* s.rot            world <- imu */
  double pos[3];     /* s.pos */
  double off_R[4];   /* s.offset_R_L_I   imu <- lidar */
  double off_T[3];   /* s.offset_T_L_I */
} orc_lio_state;
/* common::Pose6D (jueying_lio/msg/Pose6D.msg): one IMU pose of the frame */
typedef struct orc_imu_pose { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; } orc_imu_pose;
/* ImuProcess::UndistortPcl backward propagation (imu_processing.hpp:245-285), in place; the time of a point (ms, float) is
 * the float at index time_index of its record (PointXYZINormal::curvature) */
void orc_undistort(float *pts, long n, long stride_floats, long time_index, const orc_imu_pose *poses, int npose, const orc_lio_state *st);
/* pcl::VoxelGrid down-sampling of a scan (laser_mapping.cc:323-328): out holds up to n records; returns the count, -1 on index overflow */
long orc_voxel_downsample(const float *pts, long n, long stride_floats, float leaf, float *out);

/* PointCloudPreprocess::AviaHandler (jueying_lio/src/pointcloud_preprocess.cc:44-88): n livox CustomPoint records of 20 bytes -> kept points, 12 floats each; returns their count */
long orc_livox_filter(const unsigned char *msg, long n, int num_scans, int point_filter_num, double blind, float *out);

/* pclomp GICP-BFGS functor (ndt_omp/include/pclomp/gicp_omp_impl.hpp:246-365, :519-529, :125-176); orc_gicp_bfgs.c */
void orc_gicp_bfgs_apply_state(const float base[16], const double x[6], float T[16]);
void orc_gicp_bfgs_r_derivative(const double x[6], const double R[9], double g[6]);
int orc_gicp_bfgs_fdf(const float *src, const float *tgt, long stride_f, const int *idx_src, const int *idx_tgt, long m, const float *maha,
                      const float base[16], const double x[6], int mode, double *f_out, double g[6]);
int orc_set_covariances(void *h, int target, const double *cov9, long n);   /* setSource/TargetCovariances  fast_gicp_impl.hpp:93-100 */
/* the correspondence step of pclomp GICP's computeTransformation (gicp_omp_impl.hpp:405-472) on a GICP oracle; orc_gicp.c */
int orc_gicp_bfgs_correspondences(void *h, const float transformation[16], const float guess[16], int *idx_src, int *idx_tgt, float *maha9, long *m_out);
/* LaserMapping::ObsModel (jueying_lio/src/laser_mapping.cc:592-701) + the reduction the IEKF
 * applies to it, HTH = h_x^T h_x and h_x^T h (esekfom.hpp:1687,1706).  converge != 0: re-match
 * (5-NN + plane fit); converge == 0: re-use the planes of the previous call.  Returns 0, or -1
 * when there is no effective point (ekfom_data.valid = false, :657-661). */
int orc_obs_model(void *h, const orc_lio_state *s, int extrinsic_est_en, int converge, double HTH[144], double HTh[12], int *n_eff, double *sum_h2);
/* on != 0: orc_obs_model keeps residuals_, point_selected_surf_ and plane_coef_ across calls AND frames exactly as the
 * members of LaserMapping do (laser_mapping.cc:335-339 resize-with-default per frame; :616-636 a selected point that
 * fails the 81 pd2^2 test stays selected and contributes the residual stored for its index by an earlier call).
 * on == 0 (default): the clean semantics of SURVEY a14 -- such a point is dropped for that call. */
void orc_set_lio_reference_semantics(void *h, int on);
/* order of the <= 5 neighbours handed to esti_plane: ascending distance (default; the HIP kernels' order) or the order libstdc++'s
 * std::nth_element leaves in IVox::GetClosestPoint (ivox3d.h:173-178, ivox3d_node.hpp:176-181) */
#define ORC_KNN_ORDER_ASCENDING 0
#define ORC_KNN_ORDER_LIBSTDCXX 1
void orc_set_knn_order(void *h, int order);
/* NDT_P2D / NDT_D2D / VGICP_CUDA: radius > 0 selects NeighborSearchMethod::DIRECT_RADIUS (voxel offsets with |offset| <= radius + 1e-3,
 * src/fast_gicp/cuda/ndt_cuda.cu:70-83, fast_vgicp_cuda.cu:77-90); 0 returns to the DIRECT1 / 7 / 27 tables */
void orc_set_neighbor_radius(void *h, double radius);
/* the three per-point members after the last orc_obs_model in reference-semantics mode (n = current scan size) */
int orc_get_lio_members(void *h, float *plane4, float *resid, unsigned char *selected, long n);

/* IVox::AddPoints with the LRU voxel cache (jueying_lio/include/ivox3d/ivox3d.h:256-281) */
int orc_target_insert(void *h, const float *xyz, long n, long stride_floats);
/* LaserMapping::MapIncremental (jueying_lio/src/laser_mapping.cc:525-583) using the neighbours of the last
 * orc_obs_model(converge != 0) call; returns the number of points inserted through n_added. */
int orc_map_incremental(void *h, const orc_lio_state *s, double filter_size_map, int ekf_inited, long *n_added);
long orc_target_size(void *h);
long orc_target_voxels(void *h);
void orc_get_target(void *h, float *out_xyz);

/* building blocks exposed for unit pinning */
void orc_test_so3_exp(const double omega[3], double R[9]);
void orc_test_ldlt6_solve(const double A[36], const double b[6], double x[6]);
int orc_test_esti_plane(const float *pts_xyz, int n, float threshold, float plane[4]);
int orc_test_knn(void *h, const float q[3], int *idx_out, float *d2_out);
long orc_test_voxel_key(void *h, const float p[3], int key[3]);
int orc_test_gauss_voxel(void *h, const float p[3], float mean[3], float cov[9], int *n);
/* pclomp NDT (orc_pclndt.c): score, gradient and Hessian at the pose vector p = (t, euler xyz) as the
 * line search evaluates them; the double-precision Hessian pass; a voxel leaf; small pieces */
double orc_pclndt_derivatives(void *h, const double p[6], int compute_hessian, double g[6], double H[36]);
void orc_pclndt_hessian(void *h, const double p[6], double H[36]);
double orc_pclndt_score(void *h, const float T[16]);   /* calculateScore of the source transformed by T (row-major float) */
int orc_pclndt_leaf(void *h, const float pt[3], double mean[3], double icov[9], int *n);
void orc_pclndt_pose(const double p[6], float T[16]);
void orc_pclndt_euler(const float R[9], float e[3]);
void orc_pclndt_svd_solve(const double H[36], const double b[6], double x[6]);
/* exact k nearest neighbours of q in the target (grid search of orc_gicp.c); returns the count */
int orc_test_knn_exact(void *h, const float q[3], int k, int *idx, float *d2);
/* regularised per-point covariances (9 doubles each) of the target (1) or source (0) cloud */
void orc_test_covariances(void *h, int target, double *covs);
void orc_test_covariances_f(void *h, int target, float *covs);   /* CUDA-core float semantics (9 floats per point) */

#ifdef __cplusplus
}
#endif
#endif

/*
 * include/pcm_amd.h -- C ABI of the MI355X-native scan-to-submap registration path.
 *
 * This is the drop-in boundary: a plain C interface (pointers + sizes, no
 * Eigen / PCL / torch types) that the reference's PCL-style operator surface
 * binds to.  Each entry point cites the reference interface it replaces
 * (paths relative to /root/reference/src/pointcloud_match/fast_gicp unless
 * they start with jueying_lio/ or ndt_omp/).  The header-only C++ adapter
 * include/pcm_amd/registration.hpp re-creates the pcl::Registration subclass
 * on top of these calls; INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - every 4x4 transform is ROW-MAJOR (Eigen::Matrix4f is column-major: the
 *     adapter transposes);
 *   - point clouds are arrays of records whose first three floats are x,y,z
 *     (pcl::PointXYZ stride 16, PointXYZI 32, PointXYZINormal 48 bytes);
 *   - all functions return 0 on success, a negative pcm_status otherwise, and
 *     never throw or abort (the reference abort()s on bad enums:
 *     include/fast_gicp/gicp/fast_vgicp_voxel.hpp:13-15);
 *   - a context is single-threaded like a pcl::Registration object; different
 *     contexts may live on different threads / GPUs.
 */
#ifndef PCM_AMD_H
#define PCM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCM_ABI_VERSION 3

typedef enum pcm_status {
  PCM_OK = 0,
  PCM_ERR_INVALID_ARGUMENT = -1,
  PCM_ERR_NO_INPUT = -2,       /* align() before setInputSource/Target */
  PCM_ERR_HIP = -3,            /* a HIP runtime call failed; see pcm_last_error */
  PCM_ERR_UNSUPPORTED = -4,
  PCM_ERR_OUT_OF_RANGE = -5,   /* voxel coordinate outside +-2^20 cells */
  PCM_ERR_NOT_CONVERGED = -6,  /* "lm not converged!!"  impl/lsq_registration_impl.hpp:69-72 (result still written) */
  PCM_ERR_INTERNAL = -7        /* a pair of a batch was not driven to the end of its loop (library bug); its pose is not a result */
} pcm_status;

/* residual models */
typedef enum pcm_model {
  PCM_MODEL_P2PLANE = 0, /* jueying_lio/src/laser_mapping.cc:592-701  5-NN plane fit, n.p+d */
  PCM_MODEL_GICP = 1,    /* impl/fast_gicp_impl.hpp:114-237 */
  PCM_MODEL_VGICP = 2,   /* impl/fast_vgicp_impl.hpp:72-204, src/fast_gicp/cuda/compute_derivatives.cu */
  PCM_MODEL_NDT_P2D = 3, /* src/fast_gicp/cuda/ndt_compute_derivatives.cu:33-102 */
  PCM_MODEL_NDT_D2D = 4, /* src/fast_gicp/cuda/ndt_compute_derivatives.cu:104-175 */
  PCM_MODEL_VGICP_CUDA = 6, /* FastVGICPCuda's float core: src/fast_gicp/cuda/{covariance_estimation,covariance_regularization,gaussian_voxelmap,
                             * find_voxel_correspondences,compute_derivatives}.cu (resolution 1.0, DIRECT1, PLANE: impl/fast_vgicp_cuda_impl.hpp:24-27) */
  PCM_MODEL_NDT_OMP = 5  /* pclomp::NormalDistributionsTransform: pointcloud_match/ndt_omp/include/pclomp/ndt_omp_impl.hpp:69-880
                          * (Newton step + More-Thuente line search; max_iterations 35, translation_eps 0.1 = transformation_epsilon_,
                          *  voxel_resolution 1.0, num_neighbors 7 = DIRECT7 are that class's defaults) */
} pcm_model;

/* LSQ_OPTIMIZER_TYPE  include/fast_gicp/gicp/lsq_registration.hpp:13 */
typedef enum pcm_optimizer { PCM_OPT_GAUSS_NEWTON = 0, PCM_OPT_LEVENBERG_MARQUARDT = 1 } pcm_optimizer;

/* RegularizationMethod  include/fast_gicp/gicp/gicp_settings.hpp:6 */
typedef enum pcm_regularization {
  PCM_REG_NONE = 0, PCM_REG_MIN_EIG = 1, PCM_REG_NORMALIZED_MIN_EIG = 2, PCM_REG_PLANE = 3, PCM_REG_FROBENIUS = 4,
  PCM_REG_PCLOMP = 5   /* pclomp::GeneralizedIterativeClosestPoint::computeCovariances (ndt_omp/include/pclomp/gicp_omp_impl.hpp:48-122):
                        * raw second moments with float products, singular values -> (1, 1, gicp_epsilon_ = 0.001) */
} pcm_regularization;

/* where a point buffer lives */
typedef enum pcm_memory { PCM_MEM_HOST = 0, PCM_MEM_DEVICE = 1 } pcm_memory;

/*
 * Registration knobs = the reference's plain setters
 * (setMaximumIterations, setRotationEpsilon, setTransformationEpsilon,
 *  setInitialLambdaFactor: impl/lsq_registration_impl.hpp:8-38;
 *  setResolution / setNeighborSearchMethod: impl/fast_vgicp_impl.hpp:28-40;
 *  ivox_grid_resolution / ivox_nearby_type / esti_plane_threshold:
 *  jueying_lio/config/livox.yaml:44-46).
 */
typedef struct pcm_config {
  int32_t model;                 /* pcm_model */
  int32_t optimizer;             /* pcm_optimizer; default LM (lsq_registration_impl.hpp:15) */
  int32_t max_iterations;        /* 64   (:11) */
  int32_t lm_max_iterations;     /* 10   (:17) */
  double rotation_eps;           /* 2e-3 (:12) */
  double translation_eps;        /* 5e-4 (:13) */
  double lm_init_lambda_factor;  /* 1e-9 (:18) */
  float voxel_resolution;        /* iVox / voxel-map cell size [m] */
  int32_t num_neighbors;         /* 1, 7, 19 or 27 cells searched around the query; pclomp NDT: 0 = KDTREE radius search */
  int32_t knn;                   /* NUM_MATCH_POINTS 5      jueying_lio/include/options.h:14 */
  int32_t min_knn;               /* MIN_NUM_MATCH_POINTS 3  jueying_lio/include/options.h:15 */
  float max_range;               /* GetClosestPoint max_range 5.0  jueying_lio/include/ivox3d/ivox3d.h:80 */
  float plane_threshold;         /* ESTI_PLANE_THRESHOLD 0.1  jueying_lio/src/options.cc:10 */
  float max_corr_dist;           /* corr_dist_threshold_ FLT_MAX  impl/fast_gicp_impl.hpp:18 */
  int32_t k_correspondences;     /* 20  impl/fast_gicp_impl.hpp:16 */
  int32_t regularization;        /* pcm_regularization; PLANE  impl/fast_gicp_impl.hpp:20 */
  int32_t sort_source;           /* 1: order the scan along a Morton curve on device (speed only; default 1) */
  int32_t flags;                 /* PCM_FLAG_*: bits 0-1, 3, 4 and 6 speed / debugging only (never change a result), bit 2 selects the ObsModel semantics, bit 5 the neighbour row order */
  int32_t map_capacity;          /* sliding map: max voxels kept, LRU beyond (IVox capacity_ 1000000, ivox3d.h:57); 0 = unlimited */
  float ndt_step_size;           /* pclomp NDT: step_size_ 0.1 (maximum More-Thuente step)  ndt_omp_impl.hpp:48 */
  float ndt_outlier_ratio;       /* pclomp NDT: outlier_ratio_ 0.55  ndt_omp_impl.hpp:48 */
  int32_t batch_window;          /* pcm_align_batch: pairs iterating at a time; finished pairs hand their slot to queued ones (0 = all at once; speed only) */
  int32_t voxel_mode;            /* VGICP VoxelAccumulationMode (gicp_settings.hpp:10): 0 ADDITIVE (default), 1 ADDITIVE_WEIGHTED, 2 MULTIPLICATIVE */
  float neighbor_search_radius;  /* NDT_P2D / NDT_D2D / VGICP_CUDA: > 0 selects NeighborSearchMethod::DIRECT_RADIUS -- every voxel offset with
                                  * |offset| <= radius + 1e-3, radius in voxels (cuda/ndt_cuda.cu:70-83, cuda/fast_vgicp_cuda.cu:77-90;
                                  * setNeighborSearchMethod(method, radius)); num_neighbors is not read then.  0 (default): off */
  int32_t covariance_method;     /* VGICP_CUDA: PCM_COV_KNN (k nearest neighbours, default) or PCM_COV_RBF_KERNEL -- NearestNeighborMethod::GPU_RBF_KERNEL
                                  * of FastVGICPCuda (fast_vgicp_cuda_impl.hpp:107,136; cuda/covariance_estimation_rbf.cu:59-151) */
  float rbf_kernel_width;        /* 0.25: the weight of a point at squared distance d2 is expf(-rbf_kernel_width * d2)  (fast_vgicp_cuda.cu:25, :81) */
  float rbf_max_dist;            /* 3.0: points farther than this do not take part  (fast_vgicp_cuda.cu:26; setKernelWidth: 5 x width when not given) */
} pcm_config;

#define PCM_COV_KNN 0
#define PCM_COV_RBF_KERNEL 1

#define PCM_FLAG_NO_LDS_STAGING 1   /* probe the global table per lane instead of the per-tile LDS grid */
#define PCM_FLAG_LIO_REFERENCE_SEMANTICS 4
/* pcm_obs_model keeps residuals_ and point_selected_surf_ across calls AND scans exactly as the members of LaserMapping do
 * (jueying_lio/src/laser_mapping.cc:335-339 one resize-with-default per frame; :616-636 a selected point that fails the
 * `p_body.norm() > 81 pd2^2` test keeps its flag and contributes the residual an earlier call -- possibly of an older frame --
 * stored for its index; a point never stored contributes 0).  Set it before pcm_set_source of the first scan.  Off (default):
 * such a point is dropped for that call, the result depends on the current scan, map and state only. */
#define PCM_FLAG_COUNTED_SEARCH 8    /* P2PLANE: k_linearize_counted (linearize_counted.hip: voxel point counts in the LDS cell grid, four candidates per cell and
                                     * trip, rolled cell loop, DPP reductions) instead of k_linearize; same neighbour lists, planes and sums bit for bit.
                                     * 14 % fewer vector instructions and a third of the code, measured 12 % SLOWER (profiles/r03_bench_ab_*.json): A/B switch */
#define PCM_FLAG_NEIGHBOUR_LISTS 16   /* P2PLANE against a static target: the linearize pass runs on per-voxel candidate lists -- for every voxel a query can fall
                                       * into, the points of its neighbour voxels in the reference's visit order, contiguous (27 x 16 B per map point), built
                                       * on the device (neighbour_lists.hip) -- instead of re-deriving the candidates per pass through LDS.  Same candidates
                                       * in the same order: bit-identical results, 1.8x the kernel speed.  DEFAULT: the lists are built when a target is
                                       * registered against the second time; this flag builds them with the map (first registration already).  Never for a
                                       * target that has grown (pcm_target_insert / pcm_map_incremental), the LIO model, or next to another kernel flag.
                                       * pclomp NDT (PCM_MODEL_NDT_OMP): the same policy for the grid's neighbour-LEAF lists (the leaves a point's search
                                       * visits, in visiting order; 16 B per (voxel, neighbour leaf)). */
#define PCM_FLAG_NO_NEIGHBOUR_LISTS 64 /* never build them: the tile kernel (kernels.hip) serves every pass */
#define PCM_FLAG_REFERENCE_KNN_ORDER 32
/* P2PLANE align / linearize: hand esti_plane its neighbours in the row order the reference's IVox::GetClosestPoint leaves -- the
 * order of libstdc++'s std::nth_element (jueying_lio/include/ivox3d/ivox3d.h:173-178, ivox3d_node.hpp:176-181) -- instead of
 * ascending distance.  Same neighbour set; the float plane fit of a row-permuted system rounds differently (measured on the
 * bench pairs: ~17 % of the planes differ in the last bits, poses by up to 8e-5 m when an LM iteration count flips;
 * profiles/r03_knn_order_sensitivity.json).  Compatibility mode: a slower kernel (private candidate array per scan point, no LDS
 * staging), maps with at most 121 points per voxel (PCM_ERR_UNSUPPORTED beyond); pcm_obs_model does not take it. */
#define PCM_FLAG_FUSED_STEP 2       /* GN: take the step in the search kernel's last workgroup (write-through hand-off of the partial rows) instead of
                                     * a second launch; same sums in the same order; measured slower at every round size, off by default */

/* out-parameters of align(): getFinalTransformation / hasConverged /
 * getFinalHessian / nr_iterations_  (lsq_registration_impl.hpp:40-79) */
typedef struct pcm_result {
  float T[16];          /* final_transformation_ = x0.cast<float>()  (:77) */
  double T64[16];       /* x0 before the float cast */
  double H[36];         /* final_hessian_ (:119,166) */
  double cost;          /* cost of the last linearize */
  int32_t iterations;   /* nr_iterations_ */
  int32_t converged;    /* converged_ */
  int32_t num_linearize;      /* linearize passes executed */
  int32_t num_compute_error;  /* compute_error passes executed (LM) */
  int32_t num_inliers;        /* correspondences used by the last linearize */
  int32_t status;             /* pcm_status of this pair */
} pcm_result;

/* counters for byte accounting / roofline (bench.py) */
typedef struct pcm_stats {
  uint64_t linearize_launches;   /* residual-kernel launches since reset */
  uint64_t point_passes;         /* scan points evaluated (sum over launches) */
  uint64_t candidates;           /* map points scanned by the kNN search */
  uint64_t slots_probed;         /* hash slots read */
  double linearize_ms;           /* HIP-event time of those launches on the context stream */
  uint64_t target_voxels;        /* occupied voxels of the current target */
  uint64_t target_slots;         /* hash-table capacity */
  uint64_t tiles;                /* 256-point tiles searched (counter passes only) */
  uint64_t tiles_lds_grid;       /* ... whose voxel box fitted the LDS grid */
  uint64_t tiles_lds_points;     /* ... whose map points were staged through LDS as well */
  double residual_ms;            /* HIP-event time of the residual/reduction launches (every-launch mode only) */
  uint64_t timed_launches;       /* launches bracketed by HIP events (= linearize_launches unless profiling bit3 samples them) */
  uint64_t timed_pair_slots;     /* sum of the pair-list lengths of the timed launches ... */
  uint64_t launched_pair_slots;  /* ... and of all launches: the share of point_passes that falls to the timed ones */
  uint64_t lru_batch_hazards;    /* sliding map: voxels a batch touched whose previous touch was older than the batch's eviction cut-off -- the
                                  * reference's sequential LRU list (ivox3d.h:256-281) may have dropped such a voxel before the batch reached it
                                  * and re-created it with the batch's points only; the batch rule here keeps it whole.  0 = the map equals the
                                  * sequential result.  Conservative (every voxel that MIGHT differ is counted): 0-24 per frame out of 10^6
                                  * voxels in the synthetic config-5 loops */
} pcm_stats;

typedef struct pcm_ctx pcm_ctx;

/* fills the reference defaults listed above */
void pcm_default_config(pcm_config *cfg);

/* construct / destroy one registration object bound to a HIP device.
 * Replaces: FastGICP()/FastVGICPCuda()/NDTCuda() constructors
 * (impl/fast_gicp_impl.hpp:8-24, impl/fast_vgicp_cuda_impl.hpp:21-38) and the
 * LaserMapping iVox construction (jueying_lio/src/laser_mapping.cc:16). */
pcm_ctx *pcm_create(int device, const pcm_config *cfg);
void pcm_destroy(pcm_ctx *ctx);
const char *pcm_last_error(const pcm_ctx *ctx);
int pcm_get_config(const pcm_ctx *ctx, pcm_config *out);
int pcm_set_config(pcm_ctx *ctx, const pcm_config *cfg);   /* setters; target structures rebuilt lazily if needed */
int pcm_set_stream(pcm_ctx *ctx, void *hip_stream);        /* run on a caller stream (hipStream_t) */

/* setInputTarget / setInputSource  (impl/fast_gicp_impl.hpp:71-90): `tag` is
 * the caller's pointer identity; an equal non-zero tag makes the call a no-op,
 * like the reference's `if (target_ == cloud) return;`.  The library copies
 * the xyz fields into device memory it owns (voxel hash built lazily at the next
 * align).  One exception, for speed: a SOURCE given as a device buffer with a
 * 16-byte stride (pcl::PointXYZ layout) is used in place, not copied -- like the
 * reference's shared_ptr input it must stay alive and unchanged until the
 * align() that uses it has returned. */
int pcm_set_target(pcm_ctx *ctx, const void *points, size_t n, size_t stride_bytes, int memory, uint64_t tag);
int pcm_set_source(pcm_ctx *ctx, const void *points, size_t n, size_t stride_bytes, int memory, uint64_t tag);
int pcm_swap_source_and_target(pcm_ctx *ctx);              /* impl/fast_gicp_impl.hpp:50-58 */
int pcm_clear_source(pcm_ctx *ctx);                        /* :60-64 */
int pcm_clear_target(pcm_ctx *ctx);                        /* :66-69 */

/* pcl::Registration::align(out, guess) -> computeTransformation
 * (impl/lsq_registration_impl.hpp:52-79).  Transforming the output cloud is
 * left to the adapter (pcl::transformPointCloud, :78). */
int pcm_align(pcm_ctx *ctx, const float guess[16], pcm_result *out);

/* LsqRegistration::evaluateCost -> linearize (lsq_registration_impl.hpp:46-49)
 * and compute_error (impl/fast_gicp_impl.hpp:213-237). H,b may be NULL. */
int pcm_linearize(pcm_ctx *ctx, const double T[16], double H[36], double b[6], double *cost, int32_t *num_inliers);
int pcm_compute_error(pcm_ctx *ctx, const double T[16], double *cost);

/* parity hook: the plane (nx,ny,nz,d) fitted to every scan point by the last
 * pcm_linearize (plane_coef_ of jueying_lio/src/laser_mapping.cc:621-622), in the
 * scan's device order; nx = NaN marks a point that was not selected.  `out`
 * holds 4*n floats. */
int pcm_get_planes(pcm_ctx *ctx, float *out, size_t n);

/* parity hook for PCM_FLAG_LIO_REFERENCE_SEMANTICS: residuals_[i] and point_selected_surf_[i]
 * (jueying_lio/src/laser_mapping.cc:337-338, 619-635) as the last pcm_obs_model left them, in the order of the
 * caller's scan; n must equal the source size. */
int pcm_get_lio_members(pcm_ctx *ctx, float *residuals, uint8_t *selected, size_t n);

/* jueying_lio measurement model: the h_dyn_share callback LaserMapping::ObsModel
 * (jueying_lio/src/laser_mapping.cc:592-701) together with the reduction the IEKF applies
 * to its output, HTH = h_x^T h_x (12x12) and h_x^T h (esekfom.hpp:1687,1706).
 * State = the pose part of state_ikfom; quaternions in Eigen coefficient order (x,y,z,w).
 * rematch = ekfom_data.converge: non-zero -> 5-NN + plane fit for every scan point;
 * zero -> the planes of the previous call are re-used (laser_mapping.cc:616).
 * Target = the map (pcm_set_target), source = the down-sampled scan in the LiDAR frame.
 * Two semantics for a selected point that fails the ||p|| > 81 pd2^2 test (laser_mapping.cc:631-635):
 *   default                           -- the point is dropped for this call (a function of scan, map and state only);
 *   PCM_FLAG_LIO_REFERENCE_SEMANTICS  -- the reference's: point_selected_surf_[i] stays set and the row carries the residual an
 *                                        earlier call (of this frame or of an older one, by index) stored in residuals_[i];
 *                                        both members persist across calls and scans with std::vector::resize semantics
 *                                        (laser_mapping.cc:335-339).  Parity hook: pcm_get_lio_members. */
typedef struct pcm_lio_state {
  double rot[4];     /* s.rot            world <- imu */
  double pos[3];     /* s.pos */
  double off_R[4];   /* s.offset_R_L_I   imu <- lidar */
  double off_T[3];   /* s.offset_T_L_I */
} pcm_lio_state;

typedef struct pcm_obs_result {
  double HTH[144];   /* row-major 12x12 */
  double HTh[12];
  double sum_h2;     /* sum of squared residuals */
  int32_t n_eff;     /* effect_feat_num_ */
  int32_t valid;     /* ekfom_data.valid (0 when n_eff < 1, laser_mapping.cc:657-661) */
} pcm_obs_result;

int pcm_obs_model(pcm_ctx *ctx, const pcm_lio_state *state, int extrinsic_est_en, int rematch, pcm_obs_result *out);

/* Sliding submap (jueying_lio): IVox::AddPoints (jueying_lio/include/ivox3d/ivox3d.h:256-281) --
 * append points to the target; voxels beyond cfg.map_capacity are dropped least-recently-
 * touched first.  The voxel hash is rebuilt on the device at the next matching call. */
int pcm_target_insert(pcm_ctx *ctx, const void *points, size_t n, size_t stride_bytes, int memory);

/* LaserMapping::MapIncremental (jueying_lio/src/laser_mapping.cc:525-583): transform the current
 * scan with the updated state (PointBodyToWorld, :855-864), apply the map add-filter against the
 * neighbours found by the last pcm_obs_model(rematch != 0), and insert the survivors
 * (points_to_add first, then point_no_need_downsample).  ekf_inited = flg_EKF_inited_. */
int pcm_map_incremental(pcm_ctx *ctx, const pcm_lio_state *state, float filter_size_map, int ekf_inited, size_t *num_added);

/* current target points in insertion order (x,y,z per point); *n receives the count (query with out = NULL) */
int pcm_get_target(pcm_ctx *ctx, float *out_xyz, size_t capacity_points, size_t *n);

/* pclomp NDT (PCM_MODEL_NDT_OMP): one derivatives evaluation at the pose vector p = (x, y, z, roll, pitch, yaw)
 * exactly as the line search runs it.  pass 0: score + gradient + Hessian (float inner products), pass 1: score +
 * gradient, pass 2: the double-precision Hessian of computeHessian (uses the angle tables of p as well).
 * Replaces NormalDistributionsTransform::computeDerivatives / computeHessian
 * (pointcloud_match/ndt_omp/include/pclomp/ndt_omp_impl.hpp:168-267, 498-559). */
int pcm_ndt_derivatives(pcm_ctx *ctx, const double p[6], int pass, double *score, double g[6], double H[36]);

/* pcl::Registration::getFitnessScore(max_range) on the device: the source cloud transformed by the float pose T (row-major 4x4),
 * exact nearest target point of every source point, mean of the squared distances that are <= max_range (PCL compares the SQUARED
 * distance with max_range; its default is the largest double).  No point in range: the largest double, as PCL returns.
 * Replaces the CPU kd-tree pass every call site runs right after align(): jueying_slam/src/localization.cpp:325-326,
 * jueying_slam/src/mapOptmization.cpp:693,719, fast_gicp/src/align.cpp:63, fast_gicp/src/python/main.cpp get_fitness_score. */
int pcm_fitness_score(pcm_ctx *ctx, const float T[16], double max_range, double *score);

/* pclomp NDT: calculateScore (ndt_omp_impl.hpp:835-880) of the source cloud transformed by T (row-major float 4x4) */
int pcm_ndt_score(pcm_ctx *ctx, const float T[16], double *score);

/* GICP / VGICP: regularised per-point covariances (row-major 3x3 doubles, INPUT order) of the source
 * (target = 0) or target (target = 1) cloud; computes them if needed.  Query the count with out = NULL.
 * Replaces FastGICP::getSourceCovariances / getTargetCovariances
 * (fast_gicp/include/fast_gicp/gicp/fast_gicp.hpp:64-70; computed at impl/fast_gicp_impl.hpp:239-298). */
int pcm_get_covariances(pcm_ctx *ctx, int target, double *out, size_t capacity_points, size_t *n);
/* setSourceCovariances / setTargetCovariances (impl/fast_gicp_impl.hpp:93-100; pclomp gicp_omp.h:165,186): hand in the
 * per-point covariances instead of having them computed -- n matrices of `elems` doubles (9 = 3x3, 16 = Matrix4d, its 3x3
 * block is read), input order.  As in the reference they are used while their count equals the cloud's (:104-109), dropped by the
 * next pcm_set_source / pcm_set_target (:78,89) and swapped by pcm_swap_source_and_target (:55).  GICP and VGICP models. */
int pcm_set_covariances(pcm_ctx *ctx, int target, const double *covs, size_t n, int elems);

/* common::Pose6D (jueying_lio/msg/Pose6D.msg): one propagated IMU pose of the frame (rot row-major) */
typedef struct pcm_imu_pose {
  double offset_time;   /* seconds after the first lidar point */
  double acc[3], gyr[3], vel[3], pos[3], rot[9];
} pcm_imu_pose;

/* Motion compensation of a scan into its frame-end pose, in place (x, y, z of every record are rewritten).
 * Replaces the backward-propagation loop of ImuProcess::UndistortPcl (jueying_lio/include/imu_processing.hpp:245-285).
 * `time_offset_bytes`: where the float time stamp of a point [ms] sits in its record (PointXYZINormal::curvature = 36: x y z pad | normal_x normal_y normal_z pad | intensity curvature);
 * points sorted by time (imu_processing.hpp:177-178); `poses`: IMUpose_ (host memory), `end_state`: the propagated state. */
int pcm_undistort(pcm_ctx *ctx, void *points, size_t n, size_t stride_bytes, size_t time_offset_bytes, int memory, const pcm_imu_pose *poses, int num_poses,
                  const pcm_lio_state *end_state);

/* pcl::VoxelGrid down-sampling of a scan: one centroid per occupied leaf, in increasing leaf-index order, every float
 * field of the record averaged (records of 3..16 floats, x y z first).  `out` must hold n records; *n_out = cells.
 * Replaces voxel_scan_.filter() of LaserMapping::Run (jueying_lio/src/laser_mapping.cc:323-328). */
int pcm_voxel_downsample(pcm_ctx *ctx, const void *points, size_t n, size_t stride_bytes, int memory, float leaf_size, void *out, size_t capacity_points, size_t *n_out);

/* PointCloudPreprocess::AviaHandler (jueying_lio/src/pointcloud_preprocess.cc:44-88): the n points of a livox_ros_driver::CustomMsg
 * (20-byte records {uint32 offset_time; float x, y, z; uint8 reflectivity, tag, line; pad} -- msg->points.data()) filtered by line, tag,
 * point_filter_num, the duplicate test against the previous copied point and the blind radius, with the reference's own operator
 * precedence; the kept points leave in input order as pcl::PointXYZINormal records (48 bytes: x y z 1, 0 0 0 0, intensity, curvature
 * = offset_time / 1e6 [ms], 0 0).  `out` must hold n records; *n_out = kept points. */
int pcm_livox_filter(pcm_ctx *ctx, const void *custom_points, size_t n, int memory, int num_scans, int point_filter_num, double blind, void *out, size_t capacity_points,
                     size_t *n_out);

/* pclomp GICP-BFGS (jueying_slam's GICP_OMP option): the functor its BFGS minimises, evaluated on the device.
 * set_correspondences packs the outer iteration's correspondence set once -- tmp_src_/tmp_tgt_ (records of
 * stride_bytes, x y z first), tmp_idx_src_/tmp_idx_tgt_ (m indices, in range: checked for host memory only) and
 * mahalanobis_ (n_src Matrix4f, column-major, indexed by the source index) -- as estimateRigidTransformationBFGS sets
 * them (ndt_omp/include/pclomp/gicp_omp_impl.hpp:199-203, filled at :430-480).
 * fdf evaluates OptimizationFunctorWithIndices at x = (tx ty tz roll pitch yaw) on top of base_transformation_
 * (`base_T`, row-major 4x4): mode 0 = operator() (:246-274, f only), 1 = df (:278-327, g only), 2 = fdf (:331-365). */
int pcm_gicp_bfgs_set_correspondences(pcm_ctx *ctx, const void *src, size_t n_src, const void *tgt, size_t n_tgt, size_t stride_bytes, const int32_t *idx_src,
                                      const int32_t *idx_tgt, size_t m, const float *mahalanobis, int memory);
int pcm_gicp_bfgs_fdf(pcm_ctx *ctx, const float *base_T, const double *x, int mode, double *f, double *g);
/* The correspondence step of pclomp GICP's computeTransformation on the device (gicp_omp_impl.hpp:405-472), for a GICP context whose
 * source / target are *input_ / *target_ (regularization PCM_REG_PCLOMP = pclomp's computeCovariances): output = guess * input,
 * query = transformation * output, exact nearest target point within max_corr_dist, mahalanobis_ = (R C1 R^T + C2)^-1 cast to
 * float; the pairs, in source order, become the record set pcm_gicp_bfgs_fdf evaluates (cloud_src = output, as at :479) without
 * leaving the device.  `transformation`, `guess`: row-major 4x4 (transformation_, guess).  *m = number of pairs. */
int pcm_gicp_bfgs_update_correspondences(pcm_ctx *ctx, const float *transformation, const float *guess, size_t *m);
/* parity hook: the pairs of the last update (source_indices / target_indices, :466-472) and their 3x3 float matrices (row-major);
 * any pointer may be NULL */
int pcm_gicp_bfgs_get_correspondences(pcm_ctx *ctx, int32_t *idx_src, int32_t *idx_tgt, float *mahalanobis9, size_t capacity);

/* One LiDAR frame of LaserMapping::Run with the scan resident on the device from the driver message to the map update
 * (jueying_lio/src/laser_mapping.cc:323-347, 525-583): the hand-offs between the operators above never pass through host memory.
 *   pcm_lio_frame_begin   raw livox_ros_driver::CustomMsg points (20-byte records, see pcm_livox_filter) -- the ONLY host -> device
 *                         copy of the frame -- -> AviaHandler filter (pointcloud_preprocess.cc:44-88) -> motion compensation into
 *                         the frame-end pose (imu_processing.hpp:245-285; skipped when num_poses < 2) -> voxel-grid down-sampling
 *                         (laser_mapping.cc:323-328; leaf_size 0 = none) -> the result (scan_down_body_) becomes the SOURCE of
 *                         this object; *n_scan = its size.  The reference sorts the scan by time before compensating it
 *                         (imu_processing.hpp:177-178); the message order is kept here (a point's compensation depends on its own
 *                         stamp only and a Livox message is time-ordered).
 *   pcm_obs_model x k     the IEKF of the caller between the calls (esekfom.hpp:1685-1735), as before
 *   pcm_lio_frame_end     = pcm_map_incremental with the updated state: add-filter + AddPoints, the map stays on the device
 * Target = the map (pcm_set_target once, then it slides by itself). */
typedef struct pcm_lio_frame_params {
  int32_t num_scans;         /* 6   config/livox.yaml:8  scan_line */
  int32_t point_filter_num;  /* 2   config/livox.yaml:40 */
  double blind;              /* 0.1 config/livox.yaml:9 (compared squared, pointcloud_preprocess.cc:70-72) */
  float leaf_size;           /* filter_size_surf 0.5  config/livox.yaml:38; 0 = no down-sampling */
  int32_t reserved;
} pcm_lio_frame_params;
int pcm_lio_frame_begin(pcm_ctx *ctx, const void *custom_points, size_t n, int memory, const pcm_lio_frame_params *params, const pcm_imu_pose *poses, int num_poses,
                        const pcm_lio_state *end_state, size_t *n_scan);
int pcm_lio_frame_end(pcm_ctx *ctx, const pcm_lio_state *state, float filter_size_map, int ekf_inited, size_t *num_added);
/* the current source scan, x y z per point in its stored order (tests: the scan pcm_lio_frame_begin produced); out may be NULL */
int pcm_get_source(pcm_ctx *ctx, float *out_xyz, size_t capacity_points, size_t *n);

/* Batch of independent registration objects on one device (BASELINE config 3:
 * independent scan/submap pairs): all GN/LM loops advance in lock-step kernel
 * launches, no host round trip per iteration.  `guesses` = n x 16 floats.
 * `host_out` (n results) and/or `device_out` (device pointer to n packed
 * pcm_result records, e.g. the buffer handed to an RCCL all_gather) may be NULL. */
int pcm_align_batch(pcm_ctx *const *ctxs, int n, const float *guesses, pcm_result *host_out, void *device_out);

/* profiling flags: bit0 = bracket every residual launch with HIP events on the
 * launch stream (pcm_stats.linearize_ms); bit1 = collect the kNN candidate /
 * probe counters (slower kernel variant; use in an untimed pass); bit2 = in-kernel
 * phase stamps of the correspondence-search kernel (diagnostic build); bit3 (with
 * bit0) = bracket only every 4th launch, the phase moving from batch to batch: an
 * event costs a few microseconds on the critical path of every round, which at one
 * pair per round is a tenth of the round (pcm_stats.timed_launches counts them) */
int pcm_set_profiling(pcm_ctx *ctx, int flags);
/* diagnostic (profiling bit2): per-phase s_memtime sums of k_linearize, [7] = tiles; resets on read */
int pcm_debug_phase_cycles(pcm_ctx *ctx, uint64_t out[8]);
int pcm_get_stats(pcm_ctx *ctx, pcm_stats *out);
int pcm_reset_stats(pcm_ctx *ctx);
int pcm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PCM_AMD_H */

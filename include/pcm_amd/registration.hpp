// include/pcm_amd/registration.hpp -- header-only pcl::Registration adapter over the
// C ABI of include/pcm_amd.h.
//
// Drop-in for the reference's registration operators: same base class
// (pcl::Registration<PointSource, PointTarget, float>), same setters and call order
// as fast_gicp::LsqRegistration / FastGICP / FastVGICP
// (/root/reference/src/pointcloud_match/fast_gicp/include/fast_gicp/gicp/lsq_registration.hpp:15-85,
//  fast_gicp.hpp:19-100, fast_vgicp.hpp) so that call sites such as
// jueying_slam/src/localization.cpp:162-189,277,317-340 compile unchanged after
//     using Registration = pcm_amd::P2PlaneRegistration<pcl::PointXYZ, pcl::PointXYZ>;
// This header needs PCL + Eigen and therefore only compiles inside a ROS/PCL
// workspace (neither exists in the build container of this repository); it contains
// no algorithm, only the translation between PCL/Eigen types and plain pointers.
#pragma once

#if __has_include(<pcl/registration/registration.h>)

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>

#include <Eigen/Core>
#include <Eigen/Eigenvalues>
#include <Eigen/Geometry>
#include <cfloat>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../pcm_amd.h"

namespace pcm_amd {

enum class LSQ_OPTIMIZER_TYPE { GaussNewton, LevenbergMarquardt };   // lsq_registration.hpp:13

template <typename PointSource, typename PointTarget>
class LsqRegistration : public pcl::Registration<PointSource, PointTarget, float> {
public:
  using Scalar = float;
  using Base = pcl::Registration<PointSource, PointTarget, Scalar>;
  using Matrix4 = typename Base::Matrix4;
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = typename Base::PointCloudTarget;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;

protected:
  using Base::converged_;
  using Base::final_transformation_;
  using Base::input_;
  using Base::max_iterations_;
  using Base::nr_iterations_;
  using Base::reg_name_;
  using Base::target_;
  using Base::transformation_epsilon_;

public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

  explicit LsqRegistration(int model, int device = 0) {
    reg_name_ = "pcm_amd::LsqRegistration";
    pcm_default_config(&cfg_);
    cfg_.model = model;
    max_iterations_ = cfg_.max_iterations;                 // 64    lsq_registration_impl.hpp:11
    transformation_epsilon_ = cfg_.translation_eps;        // 5e-4  :13
    ctx_ = pcm_create(device, &cfg_);
    if (!ctx_) throw std::runtime_error("pcm_create failed");
    final_hessian_.setIdentity();                          // :21
  }
  ~LsqRegistration() override { pcm_destroy(ctx_); }
  LsqRegistration(const LsqRegistration&) = delete;
  LsqRegistration& operator=(const LsqRegistration&) = delete;

  // ---- LsqRegistration surface (lsq_registration_impl.hpp:26-49) ----
  void setRotationEpsilon(double eps) { cfg_.rotation_eps = eps; }
  void setInitialLambdaFactor(double f) { cfg_.lm_init_lambda_factor = f; }
  void setDebugPrint(bool) {}
  void setOptimizer(LSQ_OPTIMIZER_TYPE t) { cfg_.optimizer = t == LSQ_OPTIMIZER_TYPE::GaussNewton ? PCM_OPT_GAUSS_NEWTON : PCM_OPT_LEVENBERG_MARQUARDT; }
  const Eigen::Matrix<double, 6, 6>& getFinalHessian() const { return final_hessian_; }
  int getFinalNumIteration() const { return nr_iterations_; }                                    // ndt_omp.h:228-232

  // pcl::Registration::getFitnessScore(max_range) on the device (pcm_fitness_score): every reference call site asks for it right
  // after align() (localization.cpp:325-326, mapOptmization.cpp:693,719, align.cpp:63).  PCL's member is NOT virtual: a call through
  // a pcl::Registration base pointer still runs PCL's CPU kd-tree pass -- keep the adapter's type at the call site.
  double getFitnessScore(double max_range = DBL_MAX) {
    float T[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[i * 4 + j] = final_transformation_(i, j);
    double score = 0.0;
    check(pcm_fitness_score(ctx_, T, max_range, &score), "pcm_fitness_score");
    return score;
  }

  double evaluateCost(const Eigen::Matrix4f& relative_pose, Eigen::Matrix<double, 6, 6>* H = nullptr, Eigen::Matrix<double, 6, 1>* b = nullptr) {
    push_config();
    double T[16], Hr[36], br[6], cost = 0.0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[i * 4 + j] = static_cast<double>(relative_pose(i, j));   // Eigen is column-major: transpose into the row-major ABI
    check(pcm_linearize(ctx_, T, Hr, br, &cost, nullptr), "pcm_linearize");
    if (H) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) (*H)(i, j) = Hr[i * 6 + j];
    if (b) for (int i = 0; i < 6; i++) (*b)(i) = br[i];
    return cost;
  }

  // ---- FastGICP surface (fast_gicp_impl.hpp:26-90) ----
  void setNumThreads(int) {}                                             // no meaning on the GPU
  // corr_dist_threshold_ of the GICP family (fast_gicp_impl.hpp:18,136).  The point-to-plane matcher does not read it: its
  // search radius is iVox's own max_range (ivox3d.h:132, 5.0 m), which stays what the configuration says.
  void setMaxCorrespondenceDistance(double d) { Base::setMaxCorrespondenceDistance(d); cfg_.max_corr_dist = static_cast<float>(d); }
  void setMaxRange(double r) { cfg_.max_range = static_cast<float>(r); }   // IVox::GetClosestPoint max_range
  void setResolution(double r) { cfg_.voxel_resolution = static_cast<float>(r); }      // fast_vgicp_impl.hpp:28-30
  void setNumNeighborCells(int n) { cfg_.num_neighbors = n; }            // ivox_nearby_type 0/6/18/26 -> 1/7/19/27

  virtual void swapSourceAndTarget() {
    input_.swap(target_);
    check(pcm_swap_source_and_target(ctx_), "pcm_swap_source_and_target");
  }
  virtual void clearSource() { input_.reset(); check(pcm_clear_source(ctx_), "pcm_clear_source"); }
  virtual void clearTarget() { target_.reset(); check(pcm_clear_target(ctx_), "pcm_clear_target"); }

  void setInputSource(const PointCloudSourceConstPtr& cloud) override {
    if (input_ == cloud) return;                                         // fast_gicp_impl.hpp:72-74
    Base::setInputSource(cloud);
    check(pcm_set_source(ctx_, cloud->points.data(), cloud->size(), sizeof(PointSource), PCM_MEM_HOST, reinterpret_cast<uint64_t>(cloud.get())), "pcm_set_source");
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {
    if (target_ == cloud) return;                                        // :83-85
    Base::setInputTarget(cloud);
    check(pcm_set_target(ctx_, cloud->points.data(), cloud->size(), sizeof(PointTarget), PCM_MEM_HOST, reinterpret_cast<uint64_t>(cloud.get())), "pcm_set_target");
  }

protected:
  // pcl::Registration::align() calls this (lsq_registration_impl.hpp:52-79)
  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {
    push_config();
    float g[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) g[i * 4 + j] = guess(i, j);
    pcm_result r;
    const int rc = pcm_align(ctx_, g, &r);
    if (rc == PCM_ERR_NOT_CONVERGED) std::cerr << "lm not converged!!" << std::endl;      // :70
    else check(rc, "pcm_align");
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) final_transformation_(i, j) = r.T[i * 4 + j];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) final_hessian_(i, j) = r.H[i * 6 + j];
    nr_iterations_ = r.iterations;
    converged_ = r.converged != 0;
    last_cost_ = r.cost;
    pcl::transformPointCloud(*input_, output, final_transformation_);                    // :78
  }

  void push_config() {
    cfg_.max_iterations = max_iterations_;                   // setMaximumIterations
    cfg_.translation_eps = transformation_epsilon_;          // setTransformationEpsilon
    check(pcm_set_config(ctx_, &cfg_), "pcm_set_config");
  }
  void check(int rc, const char* what) const {
    if (rc != PCM_OK) throw std::runtime_error(std::string(what) + ": " + pcm_last_error(ctx_));
  }

  pcm_ctx* ctx_ = nullptr;
  pcm_config cfg_;
  Eigen::Matrix<double, 6, 6> final_hessian_;
  double last_cost_ = 0.0;   // cost (LSQ models) / score (pclomp NDT) of the last evaluation
};

// point-to-plane scan-to-submap ICP with jueying_lio's matcher semantics
template <typename PointSource, typename PointTarget>
class P2PlaneRegistration : public LsqRegistration<PointSource, PointTarget> {
public:
  explicit P2PlaneRegistration(int device = 0) : LsqRegistration<PointSource, PointTarget>(PCM_MODEL_P2PLANE, device) { this->reg_name_ = "pcm_amd::P2PlaneRegistration"; }
};

// fast_gicp::FastGICP (gicp/fast_gicp.hpp:24-95): setCorrespondenceRandomness, setRegularizationMethod, covariances
enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS };   // gicp_settings.hpp
enum class NeighborSearchMethod { DIRECT27, DIRECT7, DIRECT1, /* VGICP_CUDA / NDTCuda only */ DIRECT_RADIUS };   // gicp_settings.hpp:8
enum class VoxelAccumulationMode { ADDITIVE, ADDITIVE_WEIGHTED, MULTIPLICATIVE };              // gicp_settings.hpp

template <typename PointSource, typename PointTarget>
class GicpRegistration : public LsqRegistration<PointSource, PointTarget> {
public:
  explicit GicpRegistration(int device = 0, int model = PCM_MODEL_GICP) : LsqRegistration<PointSource, PointTarget>(model, device) { this->reg_name_ = "pcm_amd::GicpRegistration"; }
  void setCorrespondenceRandomness(int k) { this->cfg_.k_correspondences = k; }                         // fast_gicp_impl.hpp:61-63
  void setRegularizationMethod(RegularizationMethod m) {                                                // :66-68
    static const int map[5] = {PCM_REG_NONE, PCM_REG_MIN_EIG, PCM_REG_NORMALIZED_MIN_EIG, PCM_REG_PLANE, PCM_REG_FROBENIUS};
    this->cfg_.regularization = map[static_cast<int>(m)];
  }
  // getSourceCovariances / getTargetCovariances  fast_gicp.hpp:64-70  (Matrix4d with the 3x3 block set)
  std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>> getCovariances(bool target) {
    this->push_config();
    size_t n = 0;
    this->check(pcm_get_covariances(this->ctx_, target ? 1 : 0, nullptr, 0, &n), "pcm_get_covariances");
    std::vector<double> raw(n * 9);
    this->check(pcm_get_covariances(this->ctx_, target ? 1 : 0, raw.data(), n, &n), "pcm_get_covariances");
    std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>> out(n, Eigen::Matrix4d::Zero());
    for (size_t i = 0; i < n; i++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) out[i](a, b) = raw[i * 9 + a * 3 + b];
    return out;
  }
  void setCovariances(bool target, const std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>& covs) {
    this->push_config();
    std::vector<double> raw(covs.size() * 9);
    for (size_t i = 0; i < covs.size(); i++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) raw[i * 9 + a * 3 + b] = covs[i](a, b);
    this->check(pcm_set_covariances(this->ctx_, target ? 1 : 0, raw.data(), covs.size(), 9), "pcm_set_covariances");
  }
  // setSourceCovariances / setTargetCovariances  fast_gicp.hpp:60-62 (Matrix4d, the 3x3 block is the covariance)
  void setSourceCovariances(const std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>& covs) { setCovariances(false, covs); }
  void setTargetCovariances(const std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>& covs) { setCovariances(true, covs); }
  std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>> getSourceCovariances() { return getCovariances(false); }
  std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>> getTargetCovariances() { return getCovariances(true); }
};

// fast_gicp::FastVGICP (gicp/fast_vgicp.hpp): resolution 1.0, DIRECT1, ADDITIVE  (impl/fast_vgicp_impl.hpp:22-25)
template <typename PointSource, typename PointTarget>
class VgicpRegistration : public GicpRegistration<PointSource, PointTarget> {
public:
  explicit VgicpRegistration(int device = 0) : GicpRegistration<PointSource, PointTarget>(device, PCM_MODEL_VGICP) {
    this->reg_name_ = "pcm_amd::VgicpRegistration";
    this->cfg_.voxel_resolution = 1.0f;
    this->cfg_.num_neighbors = 1;
  }
  void setNeighborSearchMethod(NeighborSearchMethod m) { this->cfg_.num_neighbors = m == NeighborSearchMethod::DIRECT27 ? 27 : (m == NeighborSearchMethod::DIRECT7 ? 7 : 1); }
  void setVoxelAccumulationMode(VoxelAccumulationMode m) { this->cfg_.voxel_mode = static_cast<int>(m); }   // fast_vgicp_impl.hpp:38-40
};

namespace detail {
// setNeighborSearchMethod(method, radius) of the CUDA-core classes (ndt_cuda_impl.hpp:30-32, fast_vgicp_cuda_impl.hpp:59-61):
// DIRECT_RADIUS takes the radius in voxels (ndt_cuda.cu:70-83); the other methods ignore it.  Against the reference: the radius
// is stored as a float and must lie in (0, 3] voxels (align() returns PCM_ERR_INVALID_ARGUMENT beyond; the reference accepts any double)
inline void set_search_method(pcm_config& cfg, NeighborSearchMethod m, double radius) {
  cfg.neighbor_search_radius = 0.f;
  if (m == NeighborSearchMethod::DIRECT_RADIUS) cfg.neighbor_search_radius = static_cast<float>(radius);
  else cfg.num_neighbors = m == NeighborSearchMethod::DIRECT27 ? 27 : (m == NeighborSearchMethod::DIRECT7 ? 7 : 1);
}
}  // namespace detail

// fast_gicp::FastVGICPCuda (gicp/fast_vgicp_cuda.hpp; impl/fast_vgicp_cuda_impl.hpp:21-31): the float CUDA core -- 20-NN covariances,
// PLANE, resolution 1.0, DIRECT1 (cuda/fast_vgicp_cuda.cu:26-34)
template <typename PointSource, typename PointTarget>
class VgicpCudaRegistration : public GicpRegistration<PointSource, PointTarget> {
public:
  explicit VgicpCudaRegistration(int device = 0) : GicpRegistration<PointSource, PointTarget>(device, PCM_MODEL_VGICP_CUDA) {
    this->reg_name_ = "pcm_amd::VgicpCudaRegistration";
    this->cfg_.voxel_resolution = 1.0f;
    this->cfg_.num_neighbors = 1;
  }
  void setCorrespondenceRandomness(int) {}                                                       // a no-op there too (fast_vgicp_cuda_impl.hpp:37-38)
  // setNearestNeighborSearchMethod (fast_vgicp_cuda_impl.hpp:64-66): CPU_PARALLEL_KDTREE and GPU_BRUTEFORCE both give the exact
  // k nearest neighbours the covariances are built from -- one device implementation serves both; GPU_RBF_KERNEL selects the
  // RBF-kernel covariance estimator (cuda/covariance_estimation_rbf.cu:59-151 -> pcm_config.covariance_method = PCM_COV_RBF_KERNEL)
  enum class NearestNeighborMethod { CPU_PARALLEL_KDTREE, GPU_BRUTEFORCE, GPU_RBF_KERNEL };      // fast_vgicp_cuda.hpp:21
  void setNearestNeighborSearchMethod(NearestNeighborMethod m) { this->cfg_.covariance_method = m == NearestNeighborMethod::GPU_RBF_KERNEL ? PCM_COV_RBF_KERNEL : PCM_COV_KNN; }
  void setKernelWidth(double kernel_width, double max_dist = -1.0) {                             // fast_vgicp_cuda_impl.hpp:46-52 (read in RBF mode only)
    this->cfg_.rbf_kernel_width = static_cast<float>(kernel_width);
    this->cfg_.rbf_max_dist = static_cast<float>(max_dist <= 0.0 ? kernel_width * 5.0 : max_dist);
  }

public:
  void setNeighborSearchMethod(NeighborSearchMethod m, double radius = -1.0) { detail::set_search_method(this->cfg_, m, radius); }
};

// fast_gicp::NDTCuda (ndt/ndt_cuda.hpp:21-71): D2D, DIRECT7, resolution 1.0  (cuda/ndt_cuda.cu:15-22)
enum class NDTDistanceMode { P2D, D2D };
template <typename PointSource, typename PointTarget>
class NdtRegistration : public LsqRegistration<PointSource, PointTarget> {
public:
  explicit NdtRegistration(int device = 0) : LsqRegistration<PointSource, PointTarget>(PCM_MODEL_NDT_D2D, device) {
    this->reg_name_ = "pcm_amd::NdtRegistration";
    this->cfg_.voxel_resolution = 1.0f;
    this->cfg_.num_neighbors = 7;
  }
  void setDistanceMode(NDTDistanceMode m) { this->cfg_.model = m == NDTDistanceMode::P2D ? PCM_MODEL_NDT_P2D : PCM_MODEL_NDT_D2D; }
  void setNeighborSearchMethod(NeighborSearchMethod m, double radius = -1.0) { detail::set_search_method(this->cfg_, m, radius); }
};

// pclomp::NormalDistributionsTransform (ndt_omp/include/pclomp/ndt_omp.h:77-310): the operator jueying_slam's
// localization constructs (jueying_slam/src/localization.cpp:162-189).  Defaults of that class (ndt_omp_impl.hpp:48,60-63).
enum NeighborSearchMethodOmp { KDTREE, DIRECT26, DIRECT7, DIRECT1 };   // ndt_omp.h:60
template <typename PointSource, typename PointTarget>
class PclNdtRegistration : public LsqRegistration<PointSource, PointTarget> {
public:
  explicit PclNdtRegistration(int device = 0) : LsqRegistration<PointSource, PointTarget>(PCM_MODEL_NDT_OMP, device) {
    this->reg_name_ = "pcm_amd::PclNdtRegistration";
    this->cfg_.voxel_resolution = 1.0f;
    this->cfg_.num_neighbors = 7;
    this->max_iterations_ = 35;
    this->transformation_epsilon_ = 0.1;
  }
  void setStepSize(double s) { this->cfg_.ndt_step_size = static_cast<float>(s); }               // ndt_omp.h:166
  double getStepSize() const { return this->cfg_.ndt_step_size; }                                 // ndt_omp.h:157
  void setOutlierRatio(double r) { this->cfg_.ndt_outlier_ratio = static_cast<float>(r); }       // ndt_omp.h:188
  double getOutlierRatio() const { return this->cfg_.ndt_outlier_ratio; }                         // ndt_omp.h:179
  float getResolution() const { return this->cfg_.voxel_resolution; }                             // ndt_omp.h:141
  void setNeighborhoodSearchMethod(NeighborSearchMethodOmp m) {                                  // ndt_omp.h:198
    this->cfg_.num_neighbors = m == KDTREE ? 0 : (m == DIRECT26 ? 27 : (m == DIRECT7 ? 7 : 1));
  }
  double getTransformationProbability() const { return trans_probability_; }                     // ndt_omp.h:207
  // getMaxEigen (ndt_omp.h:209-223): largest pseudo-eigenvalue of the final Hessian / 1e5 -- the degeneracy metric of the localisation node
  double getMaxEigen() const {
    Eigen::EigenSolver<Eigen::Matrix<double, 6, 6>> eigen_solver(this->final_hessian_);
    const Eigen::Matrix<double, 6, 6> mat_E = eigen_solver.pseudoEigenvalueMatrix();
    double max_eigen = mat_E(0, 0);
    for (int i = 0; i < 6; i++) if (mat_E(i, i) > max_eigen) max_eigen = mat_E(i, i);
    return max_eigen / 100000.0;
  }
  // calculateScore(cloud) (ndt_omp_impl.hpp:835-880): negative log likelihood of the ALREADY TRANSFORMED cloud the caller passes;
  // here the source cloud under the given pose (identity = the cloud as it is)
  double calculateScore(const Eigen::Matrix4f& pose = Eigen::Matrix4f::Identity()) {
    this->push_config();
    float T[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[i * 4 + j] = pose(i, j);
    double score = 0.0;
    this->check(pcm_ndt_score(this->ctx_, T, &score), "pcm_ndt_score");
    return score;
  }
protected:
  void computeTransformation(typename LsqRegistration<PointSource, PointTarget>::PointCloudSource& output,
                             const typename LsqRegistration<PointSource, PointTarget>::Matrix4& guess) override {
    LsqRegistration<PointSource, PointTarget>::computeTransformation(output, guess);
    trans_probability_ = this->last_cost_ / static_cast<double>(this->input_->points.size());   // ndt_omp_impl.hpp:145
  }
  double trans_probability_ = 0.0;
};

}  // namespace pcm_amd

// The call sites spell the pclomp enumerators unqualified inside namespace pclomp (jueying_slam/src/localization.cpp:169-186:
// `ndt->setNeighborhoodSearchMethod(pclomp::DIRECT7)`).  Define PCM_AMD_PCLOMP_ALIASES before including this header, in a
// translation unit that no longer includes <pclomp/ndt_omp.h>, to keep those lines unchanged.
#ifdef PCM_AMD_PCLOMP_ALIASES
namespace pclomp {
using NeighborSearchMethod = pcm_amd::NeighborSearchMethodOmp;
using pcm_amd::KDTREE;
using pcm_amd::DIRECT26;
using pcm_amd::DIRECT7;
using pcm_amd::DIRECT1;
template <typename PointSource, typename PointTarget>
using NormalDistributionsTransform = pcm_amd::PclNdtRegistration<PointSource, PointTarget>;
}  // namespace pclomp
#endif

#endif  // __has_include(<pcl/registration/registration.h>)

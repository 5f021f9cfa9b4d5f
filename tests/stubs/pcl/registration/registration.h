// declaration-only stand-in of pcl::Registration (see tests/stubs/README.md): the members the adapter reads or overrides
#pragma once
#include <Eigen/Core>
#include <pcl/point_cloud.h>
#include <string>
namespace pcl {
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  virtual ~Registration() {}
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  Matrix4 getFinalTransformation() { return final_transformation_; }
  bool hasConverged() const { return converged_; }
  double getFitnessScore(double max_range = 1e300) { return max_range; }   // NOT virtual in PCL either
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) { computeTransformation(output, guess); }
protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Matrix4 final_transformation_;
  int nr_iterations_ = 0, max_iterations_ = 10;
  double transformation_epsilon_ = 0.0, corr_dist_threshold_ = 0.0;
  bool converged_ = false;
};
template <typename PointT, typename Scalar>
void transformPointCloud(const PointCloud<PointT>&, PointCloud<PointT>&, const Eigen::Matrix<Scalar, 4, 4>&) {}
}  // namespace pcl

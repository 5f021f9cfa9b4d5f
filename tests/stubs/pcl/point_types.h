#pragma once
namespace pcl {
struct PointXYZ { float x, y, z, pad; };
struct PointXYZI { float x, y, z, pad, intensity, p1, p2, p3; };
}  // namespace pcl

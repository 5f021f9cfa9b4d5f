// orc_knn_libstdcxx.cpp -- TEST INFRASTRUCTURE (CPU oracle), never linked into the product.
//
// The order in which IVox::GetClosestPoint hands the <= 5 neighbours to common::esti_plane is whatever std::nth_element of the
// reference's standard library leaves behind:
//   IVoxNode::KNNPointByCondition   /root/reference/src/jueying_lio/include/ivox3d/ivox3d_node.hpp:158-183
//       per voxel:   if more than K in-range points were appended: nth_element(begin + old, begin + old + K - 1, end); resize(old + K)
//   IVox::GetClosestPoint           /root/reference/src/jueying_lio/include/ivox3d/ivox3d.h:173-178
//       all voxels:  if more than max_num candidates: nth_element(begin, begin + max_num - 1, end); resize(max_num)
//                    nth_element(begin, begin, end)            -> the minimum in front, the rest in introselect's order
// with DistPoint::operator< = `dist < rhs.dist` (ivox3d_node.hpp:118).  That order is a property of the library, not of the
// reference's own code, so it is not restated: this file calls the std::nth_element of the container's libstdc++ (g++ 11.4,
// <bits/stl_algo.h> __introselect) -- the toolchain family the reference's x86 builds use -- on a struct with the same
// comparison.  (libc++ or MSVC would give another order; the reference pins none.)
#include <algorithm>

extern "C" {
typedef struct { double dist; int idx; } orc_distpt;   // = orc_distpt of pcm_oracle.c (DistPoint without the node pointer: only `dist` is compared)
void orc_std_nth_element(orc_distpt* first, int nth, int n);
}

namespace {
struct DistLess {
  bool operator()(const orc_distpt& a, const orc_distpt& b) const { return a.dist < b.dist; }   // DistPoint::operator<  ivox3d_node.hpp:118
};
}  // namespace

extern "C" void orc_std_nth_element(orc_distpt* first, int nth, int n) { std::nth_element(first, first + nth, first + n, DistLess()); }

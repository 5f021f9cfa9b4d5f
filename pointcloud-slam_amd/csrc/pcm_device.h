// pcm_device.h -- device-side data layout shared by the build and residual kernels.
//
// HBM layout of one target (submap), built once in pcm_set_target():
//   pts      float4[M]   map points grouped by voxel (voxels in ascending key
//                        order, points of a voxel in input order); .w carries
//                        the original input index as raw int bits
//   slots    Slot[cap]   linear-probed voxel hash, cap = pow2 >= 4 * #voxels
//                        (load <= 0.25 so a miss ends after ~1.3 probes);
//                        16-byte slot like the reference's CUDA bucket
//                        (pair<Vector3i,int>, gaussian_voxelmap.cuh:33)
// One source (scan): float4[N] (optionally re-ordered along the voxel grid).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcm {

constexpr int kCoordBias = 1 << 20;        // voxel coords must lie in [-2^20, 2^20)
constexpr uint64_t kEmptyKey = ~0ull;

struct Slot {
  uint64_t key;    // packed voxel coordinate, kEmptyKey when free
  uint32_t start;  // first point of the voxel in `pts`
  uint32_t count;  // number of points
};
static_assert(sizeof(Slot) == 16, "Slot must be 16 bytes");

__host__ __device__ inline uint64_t pack_key(int x, int y, int z) {
  return ((uint64_t)(uint32_t)(x + kCoordBias) << 42) | ((uint64_t)(uint32_t)(y + kCoordBias) << 21) |
         (uint64_t)(uint32_t)(z + kCoordBias);
}

// 32-bit mix of the three coordinates; additive in each coordinate before the
// finaliser so the 27 neighbour hashes share their partial products.
__host__ __device__ inline uint32_t hash_part_x(int x) { return (uint32_t)x * 0x9E3779B1u; }
__host__ __device__ inline uint32_t hash_part_y(int y) { return (uint32_t)y * 0x85EBCA77u; }
__host__ __device__ inline uint32_t hash_part_z(int z) { return (uint32_t)z * 0xC2B2AE3Du; }
__host__ __device__ inline uint32_t hash_finish(uint32_t h) {
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}
__host__ __device__ inline uint32_t hash_coord(int x, int y, int z) {
  return hash_finish(hash_part_x(x) + hash_part_y(y) + hash_part_z(z));
}

// voxel-coordinate conventions of the reference
enum CoordMode : int32_t {
  COORD_ROUND = 0,       // iVox Pos2Grid: round(p * inv_res)          jueying_lio/include/ivox3d/ivox3d.h:283-286
  COORD_FLOOR_HALF = 1   // fast_gicp: floor(p / res - 0.5)            include/fast_gicp/cuda/vector3_hash.cuh:35-38
};

// Explicit global-address-space accessors.  Pointers that reach a kernel through a
// descriptor struct are generic to the compiler; a generic (flat) load counts on
// both vmcnt and lgkmcnt and may alias LDS, which serialises every LDS write
// behind it.  Going through address_space(1) gives global_load/global_store.
#define PCM_GLOBAL __attribute__((address_space(1)))
typedef float pcm_v4f __attribute__((ext_vector_type(4)));
typedef unsigned int pcm_v4u __attribute__((ext_vector_type(4)));
#if defined(__HIPCC__)
__device__ inline float4 gload4(const float4* p) {
  const pcm_v4f v = *(const PCM_GLOBAL pcm_v4f*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline uint4 gload4u(const void* p) {
  const pcm_v4u v = *(const PCM_GLOBAL pcm_v4u*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline void gstore4(float4* p, const float4& v) { *(PCM_GLOBAL pcm_v4f*)p = (pcm_v4f){v.x, v.y, v.z, v.w}; }
__device__ inline void gstore_f(float* p, float v) { *(PCM_GLOBAL float*)p = v; }
__device__ inline void gstore_d(double* p, double v) { *(PCM_GLOBAL double*)p = v; }
__device__ inline double gload_d(const double* p) { return *(const PCM_GLOBAL double*)p; }
#endif

struct TargetView {
  const float4* pts;
  const Slot* slots;
  uint32_t mask;       // cap - 1
  uint32_t num_points;
  float inv_res;       // float(1.0 / res)
  float res;
};

struct SourceView {
  const float4* pts;
  uint32_t num_points;
};

// per-pair descriptor read by the residual kernels
struct PairDesc {
  TargetView tgt;
  SourceView src;
  float4* planes;       // N: fitted plane of each scan point from the last linearize (w = d); x = NaN -> not selected
  double* partials;     // [blocks_per_pair][kPartialStride]
};

constexpr int kNumSums = 29;        // 21 (H upper) + 6 (b) + cost + inlier count
constexpr int kPartialStride = 32;  // doubles per block partial (padded)

struct KernelParams {
  int32_t num_neighbors;
  int32_t knn;
  int32_t min_knn;
  float max_range_sq;         // smallest float >= max_range^2 (see best_offer)
  float plane_threshold;
  int32_t blocks_per_pair;    // k_residual_reduce grid.x
  int32_t points_per_block;   // k_residual_reduce points per workgroup
  int32_t tiles_per_pair;     // k_corr_search grid.x (256-point tiles)
  int32_t use_lds;            // 0: always probe the global table per lane (A/B and parity checks)
};

}  // namespace pcm

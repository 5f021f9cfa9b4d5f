// pcm_device.h -- device-side data layout shared by the build and search kernels.
//
// HBM layout of one target (submap), built once in pcm_set_target():
//   pts        float4[M]      map points grouped by voxel; voxels grouped by 8x8x8
//                             BRICK (brick-major, then z-fastest inside the brick), so
//                             all points of a brick are ONE contiguous run; .w carries
//                             the voxel-in-brick index (bits 0..8) as raw int bits, and on
//                             the first point of a voxel bit 31 + the voxel's point count
//   vox_start  uint32[V+1]    first point of every occupied voxel, same order
//   bricks     BrickSlot[cap] linear-probed hash of the occupied bricks (cap = pow2 >=
//                             4 x #bricks); the reference's CUDA bucket is one
//                             pair<Vector3i,int> per VOXEL (gaussian_voxelmap.cuh:33)
//   bmask      uint32[cap*16] 512-bit voxel occupancy of the brick in that slot
//   bpref      uint16[cap*16] occupied voxels before each mask word (rank prefix)
// A voxel lookup = one brick probe (shared by up to 512 cells), one mask bit, one
// popcount and two adjacent vox_start words.  For a LiDAR surface map the whole index
// (bricks + masks + vox_start) is a few MB per 1 M points and stays in L2 / Infinity
// Cache; only the map points themselves stream from HBM.
// One source (scan): float4[N] (re-ordered along the voxel grid for locality).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcm {

constexpr int kCoordBias = 1 << 20;        // voxel coords must lie in [-2^20, 2^20)
constexpr int kBrickShift = 3;             // 8 x 8 x 8 voxels per brick
constexpr int kBrickBias = 1 << 17;        // brick coords lie in [-2^17, 2^17)
constexpr uint64_t kEmptyKey = ~0ull;

struct BrickSlot {
  uint64_t key;       // packed brick coordinate, kEmptyKey when free
  uint32_t vox_base;  // index of the brick's first occupied voxel in vox_start
  uint32_t nvox;      // occupied voxels in the brick
  uint32_t pt_start;  // first map point of the brick in pts
  uint32_t npts;      // map points in the brick (contiguous)
  uint32_t pad[2];
};
static_assert(sizeof(BrickSlot) == 32, "BrickSlot must be 32 bytes");
constexpr uint32_t kMaxBrickSlots = 1u << 22;
constexpr int kRefMaxVoxelPoints = 121;   // PCM_FLAG_REFERENCE_KNN_ORDER: most points one voxel may hold (the kernel's private candidate array is 27 x 5 + this)
constexpr uint32_t kMaxTagCount = (1u << 22) - 1u;   // points of a voxel as its head point's tag carries them (bits 9..30 of pts.w)

__host__ __device__ inline uint64_t pack_brick(int bx, int by, int bz) {
  return ((uint64_t)(uint32_t)(bx + kBrickBias) << 36) | ((uint64_t)(uint32_t)(by + kBrickBias) << 18) | (uint64_t)(uint32_t)(bz + kBrickBias);
}
// voxel index inside its brick, z fastest
__host__ __device__ inline uint32_t local_index(int x, int y, int z) { return (uint32_t)(((x & 7) << 6) | ((y & 7) << 3) | (z & 7)); }
// sort key of a point: brick-major, then local voxel index
__host__ __device__ inline uint64_t point_key(int x, int y, int z) {
  return (pack_brick(x >> kBrickShift, y >> kBrickShift, z >> kBrickShift) << 9) | local_index(x, y, z);
}

__host__ __device__ inline uint32_t hash_finish(uint32_t h) {
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}
__host__ __device__ inline uint32_t hash_coord(int x, int y, int z) {
  return hash_finish((uint32_t)x * 0x9E3779B1u + (uint32_t)y * 0x85EBCA77u + (uint32_t)z * 0xC2B2AE3Du);
}

// Gaussian voxel payload of the NDT / VGICP models: one 48-byte record per occupied voxel,
// same order as vox_start (reference layout: int + Vector3f + Matrix3f = 52 B,
// fast_gicp/include/fast_gicp/cuda/gaussian_voxelmap.cuh:36-38)
struct GaussVoxel {
  float mx, my, mz;
  int32_t n;
  float c00, c01, c02, c11, c12, c22;   // regularised covariance (symmetric)
  float pad0, pad1;
};
static_assert(sizeof(GaussVoxel) == 48, "GaussVoxel must be 48 bytes");

// voxel-coordinate conventions of the reference
enum CoordMode : int32_t {
  COORD_ROUND = 0,       // iVox Pos2Grid: round(p * inv_res)          jueying_lio/include/ivox3d/ivox3d.h:283-286
  COORD_FLOOR_HALF = 1,  // fast_gicp CUDA: floor(p / res - 0.5), float  include/fast_gicp/cuda/vector3_hash.cuh:35-38
  COORD_FLOOR_HALF_D = 2,// fast_gicp CPU VGICP: the same in double     include/fast_gicp/gicp/fast_vgicp_voxel.hpp:158-160
  COORD_FLOOR_MUL = 3    // pclomp VoxelGridCovariance build: floor(p * inverse_leaf_size)   ndt_omp/include/pclomp/voxel_grid_covariance_omp_impl.hpp:220-222
};

__host__ __device__ inline int voxel_coord(float v, float res, float inv_res, int mode) {
  if (mode == COORD_ROUND) return (int)roundf(v * inv_res);
  if (mode == COORD_FLOOR_HALF) return (int)floorf(v / res - 0.5f);
  if (mode == COORD_FLOOR_MUL) return (int)floorf(v * inv_res);
  return (int)floor((double)v / (double)res - 0.5);
}

// voxel payload of the CPU-semantics VGICP model (AdditiveGaussianVoxel, fast_vgicp_voxel.hpp:104-122): doubles
struct VgVoxel {
  double mean[3];
  double cov[6];     // xx xy xz yy yz zz
  int32_t n;
  int32_t pad;
};
static_assert(sizeof(VgVoxel) == 80, "VgVoxel must be 80 bytes");

// voxel payload of the CUDA-core VGICP model (PCM_MODEL_VGICP_CUDA): float mean, full 3x3 float covariance (the mean of
// the point covariances, which are V diag V^-1 products and not exactly symmetric), point count
// (fast_gicp/include/fast_gicp/cuda/gaussian_voxelmap.cuh:36-38)
struct VgcVoxel {
  float mean[3];
  int32_t n;
  float cov[9];
  float pad[3];
};
static_assert(sizeof(VgcVoxel) == 64, "VgcVoxel must be 64 bytes");

// pclomp VoxelGridCovariance::Leaf as the NDT derivatives read it (mean_, icov_, nr_points; doubles)
// ndt_omp/include/pclomp/voxel_grid_covariance_omp.h:90-190
struct PclLeaf {
  double mean[3];
  double icov[9];
  int32_t n;             // nr_points; -1 when the covariance was rejected (impl :331-335, 353-357)
  int32_t in_centroids;  // 1: the leaf entered the centroid cloud the KDTREE search runs on (>= 6 points; before the rejections)
  float centroid[3];     // leaf.centroid: float sum of the points / float(n)  (impl :241-242, 275)
  float pad;
};
static_assert(sizeof(PclLeaf) == 120, "PclLeaf must be 120 bytes");
// What the float passes of computeDerivatives read of a leaf, in ONE 64-byte line: the mean (double: x_trans = x - mean is taken in
// double and cast) and (float)icov -- the cast updateDerivatives applies at every use (ndt_omp_impl.hpp:469) made once, same bits.
struct PclLeafF {
  double mean[3];
  float ci[9];
  int32_t n;   // = PclLeaf::n
};
static_assert(sizeof(PclLeafF) == 64, "PclLeafF must be one 64-byte line");

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
__device__ inline uint2 gload2u(const void* p) {
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  const v2u v = *(const PCM_GLOBAL v2u*)p;
  return make_uint2(v.x, v.y);
}
__device__ inline uint32_t gload_u(const uint32_t* p) { return *(const PCM_GLOBAL uint32_t*)p; }
__device__ inline uint32_t gload_u16(const uint16_t* p) { return *(const PCM_GLOBAL uint16_t*)p; }
__device__ inline void gstore4(float4* p, const float4& v) { *(PCM_GLOBAL pcm_v4f*)p = (pcm_v4f){v.x, v.y, v.z, v.w}; }
__device__ inline void gstore_f(float* p, float v) { *(PCM_GLOBAL float*)p = v; }
__device__ inline void gstore_d(double* p, double v) { *(PCM_GLOBAL double*)p = v; }
__device__ inline double gload_d(const double* p) { return *(const PCM_GLOBAL double*)p; }
#endif

// pose-dependent constants of one pclomp NDT derivatives pass (kernel argument, filled on the host by pclndt_host.h)
struct NdtOmpParams {
  float T[16];            // final_transformation_ (row-major)
  float j_ang[8][4];      // computeAngleDerivatives, float matrix   ndt_omp_impl.hpp:308-317
  float h_ang[16][4];     // :339-364
  double j_ang_d[8][3];   // double vectors j_ang_a_ .. j_ang_h_      :298-306
  double h_ang_d[15][3];  // h_ang_a2_ .. h_ang_f3_                   :319-337
  double gauss_d1, gauss_d2;
  int32_t num_neighbors;  // 1, 7 or 27
  int32_t pad;
};

struct TargetView {
  const float4* pts;
  const uint32_t* vox_start;
  const BrickSlot* bricks;
  const uint32_t* bmask;
  const uint16_t* bpref;
  const GaussVoxel* gvox;   // NDT / VGICP models only
  uint32_t mask;       // brick-table capacity - 1
  uint32_t num_points;
  float inv_res;       // float(1.0 / res)
  float res;
};

struct SourceView {
  const float4* pts;         // scan points (P2PLANE, NDT P2D)
  const GaussVoxel* gvox;    // source voxel distributions (NDT D2D: the elements are these)
  uint32_t num_points;       // number of source elements
};

// jueying_lio state for the measurement model (floats, prepared on the host exactly as
// laser_mapping.cc:602-603,669-671 cast them)
struct LioPose {
  float q_wl[4];     // (s.rot * s.offset_R_L_I).cast<float>(), Eigen coefficient order x,y,z,w
  float t_wl[3];     // (s.rot * s.offset_T_L_I + s.pos).cast<float>()
  float off_t[3];    // s.offset_T_L_I
  float off_R[9];    // s.offset_R_L_I.toRotationMatrix()      (row-major)
  float Rt[9];       // s.rot.toRotationMatrix().transpose()
  float pad[2];
};

// per-pair descriptor read by the residual kernels
struct PairDesc {
  TargetView tgt;
  TargetView nl;         // P2PLANE with PCM_FLAG_NEIGHBOUR_LISTS: the candidate lists of the map (neighbour_lists.hip); pts == nullptr otherwise
  SourceView src;
  float4* planes;       // N: fitted plane of each scan point from the last linearize (w = d); x = NaN -> not selected
  LioPose lio;          // LIO measurement model only
  uint32_t* nn;         // LIO: [N][5] neighbour indices into tgt.pts (~0u: none) from the last matching call
  float2* lio_aux;      // LIO reference semantics: residuals_[i] (.x) and point_selected_surf_[i] (.y != 0) in the caller's scan order
  int32_t* corr;        // NDT / VGICP: [elements][offsets] matched target voxel (or -1) of the last linearize; GICP: [N] matched target point
  const double* src_cov;   // GICP / VGICP: [N][6] regularised covariance of every source point (xx xy xz yy yz zz)
  const double* tgt_cov;   // GICP: [M][6] of every map point (map order)
  const VgVoxel* vvox;     // VGICP: voxel distributions, same order as vox_start
  const VgcVoxel* cvox;    // VGICP_CUDA: float voxel distributions
  double* maha;            // GICP / VGICP: [correspondence][6] (cov_B + R cov_A R^T)^-1 of the last linearize
  double* partials;     // [workgroups of the round][kPartialStride]
  unsigned int* counter;  // arrival tickets of the round's workgroups (0 between rounds)
};

// one object of a batched pclomp NDT registration (pclndt.hip)
struct NdtObject {
  TargetView tg;
  const PclLeaf* leaves;
  const PclLeafF* leaves_f;
  TargetView nl;          // neighbour-leaf lists of the grid (neighbour_lists.hip); pts == nullptr: none, the cells are looked up one by one
  const float4* src;
  uint32_t n, per;        // scan points, points per workgroup
  int32_t nblocks, pad;   // workgroup rows of a pass
  double* partials;       // [nblocks][kNdtStride]
};

// pair handled by grid entry b
#define PCM_PAIR_OF(kp, b) ((kp).use_list ? (int)(kp).active[(b)] : (int)(b))

constexpr int kNumSums = 29;        // 21 (H upper) + 6 (b) + cost + inlier count
constexpr int kPartialStride = 32;  // doubles per block partial (padded)
constexpr int kLioSums = 92;        // 78 (HTH upper) + 12 (H^T h) + sum h^2 + count
constexpr int kLioStride = 96;

constexpr int kMaxListedPairs = 128;

struct KernelParams {
  int32_t num_neighbors;
  int32_t knn;
  int32_t min_knn;
  float max_range_sq;         // smallest float >= max_range^2 (see best_offer)
  float plane_threshold;
  int32_t blocks_per_pair;    // k_trial grid.x
  int32_t points_per_block;   // k_trial points per workgroup
  int32_t tiles_per_pair;     // k_linearize grid.x (256-point tiles)
  int32_t use_lds;            // 0: always probe the global table per lane (A/B and parity checks)
  int32_t do_step;            // 1: k_finish_round runs the GN/LM step; 0: it exports the sums (parity hooks)
  int32_t lio_rematch;        // LIO: ekfom_data.converge (1: search + plane fit, 0: re-use the stored planes)
  int32_t lio_extrinsic;      // LIO: extrinsic_est_en (columns 6..11 of h_x)
  int32_t nb_range;           // NDT / VGICP_CUDA DIRECT_RADIUS: ceil(radius) (0: the DIRECT1 / 7 / 27 tables); the offsets are the cube
  double nb_radius;           //   [-range, range]^3 in i, j, k order, those with |offset| > radius + 1e-3 skipped (ndt_cuda.cu:70-83)
  int32_t lio_ref;            // LIO: 0 clean semantics; reference semantics with the per-point members indexed by the tile position (1) or by
                              //      the scan position kept in the point's w (2: the scan was re-ordered on device)
  int32_t lin_points_per_block;  // source elements per workgroup of the linearize kernel (256 for k_linearize tiles)
  int32_t coord_mode;         // CoordMode of the target map (GICP / VGICP kernels)
  int32_t use_list;           // 1: the grid's pair axis indexes `active` (only pairs the host still believes active are launched)
  uint8_t active[kMaxListedPairs];   // pair index of each grid entry (batches of <= kMaxListedPairs pairs; indices < 256)
  double max_corr_sq;         // GICP: corr_dist_threshold_^2 (double, as pcl::Registration holds it)
};

}  // namespace pcm

// pcm_host.h -- host-side objects behind the C ABI (include/pcm_amd.h).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/pcm_amd.h"
#include "lsq_step.h"
#include "pcm_device.h"

namespace pcm {

struct Cloud {
  float4* d_pts = nullptr;  // compact float4 points, input order (device copy, or the caller's own device buffer)
  size_t n = 0;
  size_t cap = 0;           // capacity of an owned buffer
  uint64_t tag = 0;
  bool borrowed = false;    // d_pts aliases a caller-owned 16-byte-stride device buffer (zero copy)
  void drop_buffer() {
    if (d_pts && !borrowed) hipFree(d_pts);
    d_pts = nullptr; cap = 0; borrowed = false;
  }
  void release() {
    drop_buffer();
    n = 0; tag = 0;
  }
};

struct TargetMap {   // layout: pcm_device.h
  BrickSlot* bricks = nullptr;
  uint32_t* bmask = nullptr;
  uint16_t* bpref = nullptr;
  uint32_t* vox_start = nullptr;
  float4* pts = nullptr;
  GaussVoxel* gvox = nullptr;   // NDT models
  uint32_t* order = nullptr;    // input index of every map point (kept on request: GICP covariances are reported in input order)
  size_t order_cap = 0;
  // the sorted index of the point log the tables were built from (key, log position), kept for the next batch of a sliding map
  // (voxel_hash.hip: merged, not re-sorted), and its double buffer
  uint64_t *keys_s = nullptr, *keys_t = nullptr;
  uint32_t *idx_s = nullptr, *idx_t = nullptr;
  size_t keys_cap = 0, keys_t_cap = 0, idx_cap = 0, idx_t_cap = 0, vox_cap = 0, pts_cap = 0;
  uint32_t bricks_cap = 0;      // allocated slots of bricks / bmask / bpref (cap <= bricks_cap is the table in use)
  uint32_t index_n = 0;         // log points keys_s / idx_s cover (0: no usable index)
  char* arena = nullptr;        // scratch of the incremental updates (grow-only; voxel_hash.hip BuildScratch)
  size_t arena_cap = 0;
  int* h_ctr = nullptr;         // pinned host copy of the build's counter record (one read-back per synchronisation point)
  uint32_t cap = 0, num_voxels = 0, num_bricks = 0, num_points = 0;
  uint32_t max_voxel_points = 0;   // most points in one voxel
  float res = 0.f, inv_res = 0.f;
  int coord_mode = 0;
  bool valid = false;
  void release() {
    if (bricks) hipFree(bricks);
    if (bmask) hipFree(bmask);
    if (bpref) hipFree(bpref);
    if (vox_start) hipFree(vox_start);
    if (pts) hipFree(pts);
    if (gvox) hipFree(gvox);
    if (order) hipFree(order);
    if (keys_s) hipFree(keys_s);
    if (keys_t) hipFree(keys_t);
    if (idx_s) hipFree(idx_s);
    if (idx_t) hipFree(idx_t);
    if (h_ctr) hipHostFree(h_ctr);
    if (arena) hipFree(arena);
    arena = nullptr; arena_cap = 0;
    h_ctr = nullptr;
    gvox = nullptr; order = nullptr; keys_s = keys_t = nullptr; idx_s = idx_t = nullptr;
    bricks = nullptr; bmask = nullptr; bpref = nullptr; vox_start = nullptr; pts = nullptr;
    keys_cap = keys_t_cap = idx_cap = idx_t_cap = vox_cap = pts_cap = order_cap = 0; bricks_cap = 0; index_n = 0;
    cap = num_voxels = num_bricks = num_points = 0; max_voxel_points = 0; valid = false;
  }
};

// n_indexed > 0: the first n_indexed points of the log are what map->keys_s / idx_s index; only the points behind them are new
int build_target_map(hipStream_t stream, float4* d_pts, uint32_t* n_inout, float res, int coord_mode, bool want_gauss, uint32_t capacity_voxels, TargetMap* map,
                     std::string* err, bool keep_order = false, uint32_t n_indexed = 0, uint32_t* lru_hazards = nullptr, bool subsort = false);
int load_points_to_device(hipStream_t stream, const void* points, size_t n, size_t stride, int memory, uint32_t seq0, float4* d_out, std::string* err);
// batched scan re-ordering (voxel_hash.hip)
struct SortJob {
  const float4* src;   // scan in input order
  float4* dst;         // scan along the world-grid Morton curve
  uint32_t n;
  uint32_t offset;     // first element of this scan in the concatenated key array
  uint32_t guess_index;
  uint32_t pad;
};
struct SortScratch {
  uint64_t* keys = nullptr;   // 2 x cap
  uint32_t* vals = nullptr;   // 2 x cap
  void* tmp = nullptr;
  size_t cap = 0, tmp_bytes = 0;
};
// MapIncremental on the device (voxel_hash.hip)
struct LioStateD { double rot[4], pos[3], off_R[4], off_T[3]; };
int map_incremental_device(hipStream_t stream, const float4* scan, bool scan_reordered, uint32_t n, const LioStateD& s, float filter_size_map, const uint32_t* nn,
                           const float4* map_pts, uint32_t seq0, float4* out_append, uint32_t* num_added, std::string* err);
// preprocess.hip
int undistort_device(hipStream_t stream, void* d_points, size_t n, size_t stride, size_t time_off, const pcm_imu_pose* d_poses, int npose, const LioStateD& s, std::string* err);
size_t voxel_downsample_scratch_bytes(size_t n);
size_t livox_filter_scratch_bytes(size_t n);
int livox_filter_device(hipStream_t stream, const void* d_msg, size_t n, int num_scans, int point_filter_num, double blind, void* d_out, size_t* n_out, void* scratch, std::string* err);
int voxel_downsample_device(hipStream_t stream, const void* d_in, size_t n, size_t stride, float leaf, float* d_out, size_t* n_out, void* scratch, std::string* err);
// gicp_bfgs.hip
constexpr int kGicpBfgsMaxBlocks = 512;   // rows of the partial-sum table
size_t gicp_bfgs_scratch_bytes(size_t m);
int gicp_bfgs_pack_device(hipStream_t stream, const void* d_src, const void* d_tgt, size_t stride, const int* d_idx_src, const int* d_idx_tgt, const float* d_maha, size_t m,
                          void* d_records, std::string* err);
int upload_covariances(hipStream_t stream, const TargetMap& map, const double* h_cov6, double* d_cov, std::string* err);   // input-order 6-double rows -> map order
// correspondence step of pclomp GICP-BFGS on the device (gicp.hip): packs the functor's records in source order, returns their number
int gicp_bfgs_correspond_device(hipStream_t stream, const TargetMap& tmap, int coord_mode, const TargetMap& smap, const double* src_cov, const double* tgt_cov,
                                const float* guess, const float* transformation, double max_corr_dist, float4* d_records, int32_t* d_idx_src, int32_t* d_idx_tgt, uint32_t* m_out,
                                std::string* err);
int gicp_bfgs_fdf_device(hipStream_t stream, const void* d_records, size_t m, const float T[16], const float base[16], double* d_partials, double* d_sums, std::string* err);
int sort_sources_batched(hipStream_t stream, const SortJob* d_jobs, int njobs, uint32_t max_n, uint32_t total, const float* d_guesses, float res,
                         SortScratch* ws, std::string* err);

// residual-kernel launchers (kernels.hip, ndt.hip, gicp.hip)
struct LaunchGeom {
  int npairs;
  int blocks_per_pair;
  int points_per_block;
};
void launch_linearize(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes,
                      unsigned long long* d_stats, bool timing);
void launch_linearize_counted(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes,
                           unsigned long long* d_stats, bool timing = false);
// per-voxel candidate lists of a static point-to-plane map (neighbour_lists.hip)
struct NeighbourLists {
  TargetMap index;            // brick hash over one stand-in point per voxel of the dilated occupied set: voxel -> list rank
  uint32_t* start = nullptr;  // [num_lists + 1] first candidate of every list
  float4* pts = nullptr;      // candidates, a list after the other, in the reference's visit order; w = index in the map's point array
  size_t start_cap = 0, pts_cap = 0, num_candidates = 0;
  uint32_t num_lists = 0;
  int num_neighbors = 0;      // the neighbourhood the lists were built for
  int kind = 0;               // 0: candidate points (P2PLANE); 1: neighbour leaves of a pclomp NDT grid (centroid, leaf index); 2: rows of neighbour voxel indices (k_ndt)
  bool valid = false;
  void release();
};
int build_neighbour_lists(hipStream_t stream, const TargetMap& map, int num_neighbors, NeighbourLists* out, std::string* err, const PclLeaf* ndt_leaves = nullptr, bool voxel_slots = false);
TargetView view_of_lists(const NeighbourLists& l);
void launch_linearize_lists(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes);
void launch_linearize_reforder(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes);
void launch_linearize_fused(hipStream_t stream, const PairDesc* d_descs, PairState* d_states, const KernelParams& kp, const LsqParams& lp, int npairs, unsigned char* d_flags_row);
void launch_lio_obs(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp);
void launch_lio_finish(hipStream_t stream, const double* d_partials, int nblocks, double* d_out);
void launch_lio_members_init(hipStream_t stream, float2* aux, uint32_t first, uint32_t last);   // entries [first, last) = (residual 0, selected)
void launch_trial(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs);
void launch_finish_round(hipStream_t stream, const PairDesc* d_descs, PairState* d_states, const KernelParams& kp, const LsqParams& lp, int npairs, bool trial_round,
                         bool write_flags, unsigned char* d_flags_row, double* d_sums, unsigned int* d_queue = nullptr, int total_pairs = 0);
void launch_ndt(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, int kind, bool trial);   // kind 0 P2D, 1 D2D, 2 VGICP_CUDA
void launch_fitness(hipStream_t stream, const TargetView& tg, int coord_mode, const float4* src, uint32_t n, const float* T, double max_range, double* d_out);
void launch_gicp(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool vgicp, bool trial);
// gicp.hip: kNN covariances of every point of a built map (map order, 6 doubles each); VGICP voxel distributions
int compute_covariances(hipStream_t stream, const TargetMap& map, int k, int regularization, double* d_out, std::string* err, const TargetMap* fine = nullptr);
// RBF-kernel covariances of the CUDA core (GPU_RBF_KERNEL): map order out, sums over the input order
int compute_covariances_rbf(hipStream_t stream, const TargetMap& map, const float4* d_input_order, uint32_t n, double kernel_width, double max_dist, int regularization, double* d_out,
                            std::string* err);
int build_vgicp_voxels(hipStream_t stream, const TargetMap& map, const double* d_cov, int mode, VgVoxel* d_out, std::string* err);
int build_vgc_voxels(hipStream_t stream, const TargetMap& map, const double* d_cov, VgcVoxel* d_out, std::string* err);
// pclndt.hip: pclomp NDT leaves and derivative passes (pass 0: score+gradient+Hessian, 1: score+gradient, 2: double Hessian only)
int build_pclndt_leaves(hipStream_t stream, const TargetMap& map, PclLeaf* d_out, PclLeafF* d_out_f, std::string* err);
int pclndt_workgroups(uint32_t n, uint32_t* per_out);
namespace ndtomp { struct NdtMachine; }
NdtObject make_ndt_object(const TargetMap& map, const PclLeaf* leaves, const PclLeafF* leaves_f, const TargetView& nl, const float4* src, uint32_t n, double* d_partials);
// one round of a batched pclomp NDT registration: the pass every live object waits for, then the sums + solver step per object
void launch_pclndt_batch_round(hipStream_t stream, const NdtObject* d_objs, ndtomp::NdtMachine* d_ms, int nobj, int max_blocks, unsigned char* d_flags_row);
void launch_pclndt_pass(hipStream_t stream, const TargetMap& map, const PclLeaf* leaves, const PclLeafF* leaves_f, const TargetView& nl, const float4* src, uint32_t n, const NdtOmpParams& P, int pass, double* d_partials, double* d_out,
                        double gauss_d3 = 0.0);   // pass 3: calculateScore (needs gauss_d3)
void launch_init_states(hipStream_t stream, PairState* d_states, const float* d_guesses, int npairs, int max_iterations, int window, unsigned int* d_queue);
void launch_pack_results(hipStream_t stream, const PairState* d_states, pcm_result* d_results, int npairs);

}  // namespace pcm

struct pcm_ctx {
  int device = 0;
  pcm_config cfg{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  pcm::Cloud src, tgt;
  pcm::TargetMap map;
  pcm::TargetMap srcmap;          // NDT D2D: the source's own voxel distributions
  bool tgt_dynamic = false;       // the target grew since pcm_set_target (pcm_target_insert / pcm_map_incremental)
  bool nlists_failed = false;     // the lists of this map could not be built (memory): not tried again
  int map_uses = 0;               // prepare() calls since the map was (re)built
  pcm::NeighbourLists nlists;     // P2PLANE, PCM_FLAG_NEIGHBOUR_LISTS: candidate lists of `map` (static targets)
  pcm::TargetMap covfine;         // GICP: the cloud whose covariances are being computed, on a grid 8x finer (kNN index of dense neighbourhoods only)
  int32_t* corr = nullptr;        // NDT: matched voxel per (element, offset) of the last linearize
  size_t corr_cap = 0;
  // GICP / VGICP: per-point covariances in MAP order (the source elements are srcmap.pts), voxel distributions, Mahalanobis cache
  double* src_cov = nullptr;
  double* tgt_cov = nullptr;
  size_t src_cov_cap = 0, tgt_cov_cap = 0;
  bool src_cov_valid = false, tgt_cov_valid = false;
  int cov_k = 0, cov_reg = -1, cov_vmode = -1;
  float cov_rbf_w = -1.f, cov_rbf_d = -1.f;   // RBF parameters the cached covariances were computed with (-1: kNN covariances)
  pcm::VgVoxel* vvox = nullptr;
  size_t vvox_cap = 0;
  pcm::VgcVoxel* cvox = nullptr;   // VGICP_CUDA
  size_t cvox_cap = 0;
  double* maha = nullptr;
  size_t maha_cap = 0;
  // pclomp NDT: leaf payload of the map, partial rows, result row (device + pinned host)
  pcm::PclLeaf* pleaf = nullptr;
  pcm::PclLeafF* pleaf_f = nullptr;   // the float passes' 64-byte view of the same leaves
  size_t pleaf_cap = 0;
  bool pleaf_valid = false;
  double* ndt_partials = nullptr;
  size_t ndt_partials_cap = 0;
  double* ndt_out = nullptr;
  double* ndt_out_host = nullptr;
  float4* src_order = nullptr;   // the scan re-ordered along the world-grid Morton curve (speed only)
  size_t src_order_cap = 0;
  bool src_sorted = false;       // src_order holds the current source
  float4* planes = nullptr;
  unsigned int* counter = nullptr;   // round tickets (device, one word)
  size_t planes_cap = 0;
  std::string err;
  pcm_stats stats{};
  uint64_t phase_cycles[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // diagnostic (profiling bit2)
  uint32_t* nn = nullptr;          // LIO: the <= 5 neighbours (indices into map.pts) of every scan point from the last matching call
  size_t nn_cap = 0;
  uint32_t next_seq = 0;           // next insertion sequence number of the target point log
  bool lio_planes_valid = false;   // planes of the last pcm_obs_model(rematch=1) belong to the current scan
  float2* lio_aux = nullptr;       // PCM_FLAG_LIO_REFERENCE_SEMANTICS: residuals_ / point_selected_surf_ of LaserMapping, in the caller's scan order;
  size_t lio_aux_n = 0, lio_aux_cap = 0;   // they outlive the scan (std::vector::resize semantics, laser_mapping.cc:337-338)
  void* ws = nullptr;   // batch workspace owned by this context (pcm_api.hip)
  void* ndt_ws = nullptr;   // pclomp NDT: objects + solver machines of a batch (pcm_api.hip)
  char* pre_arena = nullptr;   // grow-only device scratch of the pre-processing operators
  size_t pre_arena_cap = 0;
  char* bfgs = nullptr;        // GICP-BFGS functor: packed correspondence records + partial sums (gicp_bfgs.hip)
  size_t bfgs_cap = 0, bfgs_m = 0;
  std::vector<double> user_cov[2];   // [0] source, [1] target: covariances handed in by the caller (6 per point, input order); empty = compute
  int32_t* bfgs_idx = nullptr; // [2][bfgs_idx_cap] source / target index of every packed pair (device-side correspondence step)
  size_t bfgs_idx_cap = 0;
  double* bfgs_host = nullptr; // pinned, device-visible: the 14 sums land here without a copy command
  int profiling = 0;  // bit0: HIP-event timing of residual launches, bit1: kNN counters
};

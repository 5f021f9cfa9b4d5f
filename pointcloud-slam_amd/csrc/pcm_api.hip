// pcm_api.hip -- the C ABI of include/pcm_amd.h: registration objects, the batched
// device-resident GN/LM loop and the parity hooks (linearize / compute_error).
//
// Host-side counterpart of the reference's wrappers
//   FastVGICPCuda / NDTCuda host classes   /root/reference/src/pointcloud_match/fast_gicp/include/fast_gicp/gicp/impl/fast_vgicp_cuda_impl.hpp:21-180
//   LsqRegistration::computeTransformation  .../impl/lsq_registration_impl.hpp:52-79
// The reference crosses host<->device >= 4 times per Gauss-Newton iteration
// (SURVEY.md §2.3); here the loop state lives on the device and the host only
// polls a per-round "pairs still active" counter one round behind the GPU.
#include "pcm_host.h"
#include "pclndt_host.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>

using namespace pcm;

namespace {

#define HIPCK(ctx, x)                                                                \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->err = std::string(#x) + ": " + hipGetErrorString(e_);                   \
      return PCM_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

// grow-only device workspace shared by the batch launches of one device
struct Workspace {
  int device = -1;
  PairDesc* d_descs = nullptr;
  PairState* d_states = nullptr;
  float* d_guesses = nullptr;
  pcm_result* d_results = nullptr;
  double* d_partials = nullptr;
  double* d_sums = nullptr;
  unsigned char* d_flags = nullptr;  // device view of h_flags
  unsigned char* h_flags = nullptr;  // mapped pinned host memory: [round][pair] status bytes written by k_finish_round
  size_t cap_flags = 0;
  unsigned long long* d_stats = nullptr;
  unsigned int* d_queue = nullptr;   // batch window: index of the next queued pair
  SortJob* d_jobs = nullptr;
  SortScratch sort;
  int cap_pairs = 0;
  size_t cap_partials = 0;
  int cap_rounds = 0;
  std::vector<hipEvent_t> ev_round;
  std::vector<hipEvent_t> ev_prof;
};

// Spin until a round's status byte (mapped pinned host memory, stored by the step kernel) is non-zero.  On both error returns
// the queued kernels may still be writing into the workspace and the flags: the stream is drained first, so that the caller can
// destroy the context safely.  The runtime is asked (hipStreamQuery: has the stream died?) only after a wait far beyond any
// round -- a query takes the locks the other slots' launches need; polled every few microseconds by several waiting threads it
// throttled every launch of the process.
int wait_status_byte(pcm_ctx* c0, hipStream_t st, volatile unsigned char* p, std::chrono::steady_clock::time_point t_start) {
  unsigned spins = 0;
  auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);
  while (*p == 0) {
    if ((++spins & 0xfff) != 0) continue;
    const auto now = std::chrono::steady_clock::now();
    if (now - t_start > std::chrono::seconds(20)) { (void)hipStreamSynchronize(st); c0->err = "timeout waiting for the GPU round status"; return PCM_ERR_HIP; }
    if (now >= next_query) {
      next_query = now + std::chrono::milliseconds(5);
      if (hipStreamQuery(st) == hipSuccess && *p == 0) { c0->err = "stream drained without a round status (kernel fault?)"; return PCM_ERR_HIP; }
    }
  }
  return PCM_OK;
}

// The workspace belongs to the first context of a batch (contexts are single-threaded
// objects), so independent batches may run concurrently from different host threads
// on their own streams -- e.g. the stragglers of one batch under the bulk of the next.
int ensure_ws(pcm_ctx* c, Workspace** out, int npairs, size_t partial_doubles, int rounds) {
  if (!c->ws) c->ws = new (std::nothrow) Workspace();
  if (!c->ws) { c->err = "out of host memory"; return PCM_ERR_HIP; }
  Workspace& w = *static_cast<Workspace*>(c->ws);
  w.device = c->device;
  if (npairs > w.cap_pairs) {
    if (w.d_descs) { hipFree(w.d_descs); hipFree(w.d_states); hipFree(w.d_guesses); hipFree(w.d_results); hipFree(w.d_sums); hipFree(w.d_jobs); }
    // a failed hipMalloc below returns at once (HIPCK): nothing freed here may stay reachable, or a retry / pcm_destroy frees it twice
    w.d_descs = nullptr; w.d_states = nullptr; w.d_guesses = nullptr; w.d_results = nullptr; w.d_sums = nullptr; w.d_jobs = nullptr;
    w.cap_pairs = 0;
    const int cap = std::max(npairs, 64);
    HIPCK(c, hipMalloc(&w.d_descs, sizeof(PairDesc) * cap));
    HIPCK(c, hipMalloc(&w.d_states, sizeof(PairState) * cap));
    HIPCK(c, hipMalloc(&w.d_guesses, sizeof(float) * 16 * cap));
    HIPCK(c, hipMalloc(&w.d_results, sizeof(pcm_result) * cap));
    HIPCK(c, hipMalloc(&w.d_sums, sizeof(double) * kPartialStride * cap));
    HIPCK(c, hipMalloc(&w.d_jobs, sizeof(SortJob) * cap));
    w.cap_pairs = cap;
  }
  if (partial_doubles > w.cap_partials) {
    if (w.d_partials) hipFree(w.d_partials);
    w.d_partials = nullptr; w.cap_partials = 0;
    HIPCK(c, hipMalloc(&w.d_partials, sizeof(double) * partial_doubles));
    w.cap_partials = partial_doubles;
  }
  if ((size_t)rounds * (size_t)std::max(npairs, 64) > w.cap_flags) {
    if (w.h_flags) hipHostFree(w.h_flags);
    w.h_flags = nullptr; w.cap_flags = 0;
    const size_t bytes = (size_t)rounds * (size_t)std::max(npairs, 64);
    // per-round status bytes of every pair live in mapped pinned host memory: k_finish_round
    // stores them directly (posted writes); the host polls them, no event / copy per round
    HIPCK(c, hipHostMalloc(reinterpret_cast<void**>(&w.h_flags), bytes, hipHostMallocMapped));
    HIPCK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&w.d_flags), w.h_flags, 0));
    w.cap_flags = bytes;
  }
  if (!w.d_stats) HIPCK(c, hipMalloc(&w.d_stats, sizeof(unsigned long long) * 16));
  if (!w.d_queue) HIPCK(c, hipMalloc(&w.d_queue, sizeof(unsigned int)));
  while ((int)w.ev_round.size() < 2) {
    hipEvent_t e;
    HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    w.ev_round.push_back(e);
  }
  *out = &w;
  return PCM_OK;
}

void free_ws(pcm_ctx* c) {
  Workspace* w = static_cast<Workspace*>(c->ws);
  if (!w) return;
  hipFree(w->d_descs); hipFree(w->d_states); hipFree(w->d_guesses); hipFree(w->d_results); hipFree(w->d_partials); hipFree(w->d_sums);
  hipFree(w->d_stats); hipFree(w->d_queue); hipFree(w->d_jobs); hipFree(w->sort.keys); hipFree(w->sort.vals); hipFree(w->sort.tmp);
  if (w->h_flags) hipHostFree(w->h_flags);
  for (hipEvent_t e : w->ev_round) hipEventDestroy(e);
  for (hipEvent_t e : w->ev_prof) hipEventDestroy(e);
  delete w;
  c->ws = nullptr;
}

int coord_mode_for(int model) {
  if (model == PCM_MODEL_P2PLANE || model == PCM_MODEL_GICP) return COORD_ROUND;   // GICP: the grid is only the kNN index, any convention serves
  if (model == PCM_MODEL_NDT_OMP) return COORD_FLOOR_MUL;
  return model == PCM_MODEL_VGICP ? COORD_FLOOR_HALF_D : COORD_FLOOR_HALF;
}

// PCM_COV_FINE_INDEX=1 switches the fine kNN index of the covariance pass on (measured: fewer candidates, but the second index
// build and its query order cost more than they save on the bench scans -- DESIGN section 3)
bool cov_fine_index_enabled() {
  static const bool on = [] { const char* e = getenv("PCM_COV_FINE_INDEX"); return e && e[0] == '1'; }();
  return on;
}

// PCM_COV_SUBSORT=0: the scan's kNN index keeps input order inside its voxels (A/B measurements)
bool cov_subsort_enabled() {
  static const bool on = [] { const char* e = getenv("PCM_COV_SUBSORT"); return !(e && e[0] == '0'); }();
  return on;
}

size_t num_elements(const pcm_ctx* c) { return c->cfg.model == PCM_MODEL_NDT_D2D ? (size_t)c->srcmap.num_voxels : c->src.n; }

bool is_ndt(int model) { return model == PCM_MODEL_NDT_P2D || model == PCM_MODEL_NDT_D2D; }
bool is_gicp(int model) { return model == PCM_MODEL_GICP || model == PCM_MODEL_VGICP || model == PCM_MODEL_VGICP_CUDA; }   // models with per-point covariances
bool radius_model(int model) { return is_ndt(model) || model == PCM_MODEL_VGICP_CUDA; }
// offsets examined per element: DIRECT_RADIUS walks the cube around the voxel (the list is its subset), else the table
size_t neighbor_slots(const pcm_config& g) {
  if (radius_model(g.model) && g.neighbor_search_radius > 0.f) {
    const size_t D = 2 * (size_t)std::ceil((double)g.neighbor_search_radius) + 1;
    return D * D * D;
  }
  return (size_t)g.num_neighbors;
}
int ndt_kind(int model) { return model == PCM_MODEL_NDT_D2D ? 1 : (model == PCM_MODEL_VGICP_CUDA ? 2 : 0); }

int validate_config(pcm_ctx* c, const pcm_config& g) {
  if (g.model != PCM_MODEL_P2PLANE && !is_ndt(g.model) && !is_gicp(g.model) && g.model != PCM_MODEL_NDT_OMP) { c->err = "unknown registration model"; return PCM_ERR_UNSUPPORTED; }
  if (g.model == PCM_MODEL_NDT_OMP) {
    if (g.num_neighbors == 19) { c->err = "pclomp NDT neighbourhoods are KDTREE / DIRECT1 / DIRECT7 / DIRECT26 (num_neighbors 0, 1, 7, 27)"; return PCM_ERR_INVALID_ARGUMENT; }
    if (!(g.ndt_step_size > 0.f) || !(g.ndt_outlier_ratio > 0.f) || !(g.ndt_outlier_ratio < 1.f)) { c->err = "bad ndt_step_size / ndt_outlier_ratio"; return PCM_ERR_INVALID_ARGUMENT; }
  }
  if ((is_ndt(g.model) || g.model == PCM_MODEL_VGICP || g.model == PCM_MODEL_VGICP_CUDA) && g.num_neighbors == 19) {
    c->err = "NDT / VGICP neighbourhoods are DIRECT1 / DIRECT7 / DIRECT27 (num_neighbors 1, 7, 27)"; return PCM_ERR_INVALID_ARGUMENT;
  }
  if (is_gicp(g.model)) {
    if (g.k_correspondences < 1 || g.k_correspondences > 64) { c->err = "k_correspondences must be in [1, 64]"; return PCM_ERR_INVALID_ARGUMENT; }
    if (g.regularization < PCM_REG_NONE || g.regularization > PCM_REG_PCLOMP || (g.regularization == PCM_REG_PCLOMP && g.model == PCM_MODEL_VGICP_CUDA)) { c->err = "bad regularization method"; return PCM_ERR_INVALID_ARGUMENT; }
    if (!(g.max_corr_dist > 0.f)) { c->err = "max_corr_dist must be > 0"; return PCM_ERR_INVALID_ARGUMENT; }
    if (g.voxel_mode < 0 || g.voxel_mode > 2) { c->err = "voxel_mode must be 0 (ADDITIVE), 1 (ADDITIVE_WEIGHTED) or 2 (MULTIPLICATIVE)"; return PCM_ERR_INVALID_ARGUMENT; }
  }
  if (g.covariance_method != PCM_COV_KNN) {
    if (g.covariance_method != PCM_COV_RBF_KERNEL || g.model != PCM_MODEL_VGICP_CUDA) { c->err = "covariance_method: PCM_COV_RBF_KERNEL is a mode of VGICP_CUDA (NearestNeighborMethod::GPU_RBF_KERNEL)"; return PCM_ERR_INVALID_ARGUMENT; }
    if (!(g.rbf_kernel_width > 0.f) || !(g.rbf_max_dist > 0.f)) { c->err = "rbf_kernel_width and rbf_max_dist must be > 0"; return PCM_ERR_INVALID_ARGUMENT; }
  }
  if (g.neighbor_search_radius != 0.f) {   // NeighborSearchMethod::DIRECT_RADIUS: "supported on only VGICP_CUDA" (gicp_settings.hpp:8) and NDTCuda
    if (!radius_model(g.model)) { c->err = "neighbor_search_radius (DIRECT_RADIUS) is a mode of NDT_P2D / NDT_D2D / VGICP_CUDA"; return PCM_ERR_INVALID_ARGUMENT; }
    if (!(g.neighbor_search_radius > 0.f) || g.neighbor_search_radius > 3.f) { c->err = "neighbor_search_radius must be in (0, 3] voxels"; return PCM_ERR_INVALID_ARGUMENT; }
  }
  if (g.optimizer != PCM_OPT_GAUSS_NEWTON && g.optimizer != PCM_OPT_LEVENBERG_MARQUARDT) { c->err = "bad optimizer"; return PCM_ERR_INVALID_ARGUMENT; }
  if (!(g.voxel_resolution > 0.f)) { c->err = "voxel_resolution must be > 0"; return PCM_ERR_INVALID_ARGUMENT; }
  if (g.num_neighbors != 1 && g.num_neighbors != 7 && g.num_neighbors != 19 && g.num_neighbors != 27 && !(g.model == PCM_MODEL_NDT_OMP && g.num_neighbors == 0)) {
    c->err = "num_neighbors must be 1, 7, 19 or 27"; return PCM_ERR_INVALID_ARGUMENT;
  }
  if (g.knn != 5 || g.min_knn != 3) { c->err = "knn/min_knn are the reference constants 5/3 (options.h:14-15)"; return PCM_ERR_UNSUPPORTED; }
  if (!(g.rotation_eps > 0) || !(g.translation_eps > 0)) { c->err = "epsilons must be > 0"; return PCM_ERR_INVALID_ARGUMENT; }
  return PCM_OK;
}

int set_cloud(pcm_ctx* c, Cloud* cl, const void* points, size_t n, size_t stride, int memory, uint64_t tag, bool allow_borrow) {
  if (!points && n) { c->err = "null point buffer"; return PCM_ERR_INVALID_ARGUMENT; }
  if (stride < 3 * sizeof(float) || (stride % sizeof(float)) != 0) { c->err = "stride must be a multiple of 4 and >= 12 bytes"; return PCM_ERR_INVALID_ARGUMENT; }
  if (n > 0x7fffffffull) { c->err = "cloud too large"; return PCM_ERR_INVALID_ARGUMENT; }
  HIPCK(c, hipSetDevice(c->device));
  if (allow_borrow && memory == PCM_MEM_DEVICE && stride == sizeof(float4) && (reinterpret_cast<uintptr_t>(points) & 15u) == 0) {
    // a device-resident PointXYZ-layout scan is used in place (the kernels only read x,y,z):
    // like the reference's shared_ptr input, the caller keeps it alive and unchanged until align() returns
    cl->drop_buffer();
    cl->d_pts = const_cast<float4*>(static_cast<const float4*>(points));
    cl->borrowed = true;
    cl->n = n;
    cl->tag = tag;
    return PCM_OK;
  }
  if (cl->borrowed) cl->drop_buffer();
  if (n > cl->cap) {
    if (cl->d_pts) hipFree(cl->d_pts);
    cl->d_pts = nullptr; cl->cap = 0;
    HIPCK(c, hipMalloc(&cl->d_pts, sizeof(float4) * n));
    cl->cap = n;
  }
  cl->n = n;
  cl->tag = tag;
  int rc = load_points_to_device(c->stream, points, n, stride, memory, 0u, cl->d_pts, &c->err);
  if (rc != PCM_OK) return rc;
  HIPCK(c, hipStreamSynchronize(c->stream));  // the caller may free/reuse its buffer on return
  return PCM_OK;
}

// P2PLANE against a static target: the linearize pass runs on per-voxel candidate lists (neighbour_lists.hip).  The lists cost a
// build (several ms and 27 x 16 B per map point), so by default they are made when a target is registered against the SECOND time
// (the reference's own protocols re-use a target: fast_gicp/src/align.cpp:51-104, jueying_slam's localization against one global map);
// PCM_FLAG_NEIGHBOUR_LISTS builds them with the map, PCM_FLAG_NO_NEIGHBOUR_LISTS never.  A target that grows through
// pcm_target_insert / pcm_map_incremental would rebuild them with every batch: it keeps the tile kernel.
bool uses_neighbour_lists(const pcm_ctx* c) {
  if (c->cfg.model != PCM_MODEL_P2PLANE || c->tgt_dynamic || c->nlists_failed || (c->cfg.flags & PCM_FLAG_NO_NEIGHBOUR_LISTS)) return false;
  if (c->cfg.flags & (PCM_FLAG_REFERENCE_KNN_ORDER | PCM_FLAG_COUNTED_SEARCH | PCM_FLAG_NO_LDS_STAGING | PCM_FLAG_FUSED_STEP)) return false;   // another kernel was asked for
  return (c->cfg.flags & PCM_FLAG_NEIGHBOUR_LISTS) != 0 || c->map_uses >= 2;
}

// fast_gicp NDTCuda (P2D / D2D) and VGICP of the CUDA core with a DIRECT neighbourhood: k_ndt reads rows of neighbour voxel indices
bool uses_voxel_slot_lists(const pcm_ctx* c) {
  const int m = c->cfg.model;
  if (!(m == PCM_MODEL_NDT_P2D || m == PCM_MODEL_NDT_D2D || m == PCM_MODEL_VGICP_CUDA) || c->cfg.neighbor_search_radius > 0.f) return false;
  if (c->nlists_failed || (c->cfg.flags & PCM_FLAG_NO_NEIGHBOUR_LISTS) || c->map.coord_mode != COORD_FLOOR_HALF) return false;
  return (c->cfg.flags & PCM_FLAG_NEIGHBOUR_LISTS) != 0 || c->map_uses >= 2;
}

// pclomp NDT: the neighbour-leaf lists of the context's grid, or an empty view (the cells are then looked up one by one)
TargetView ndt_lists_view(const pcm_ctx* c) {
  const bool on = c->cfg.model == PCM_MODEL_NDT_OMP && c->nlists.valid && c->nlists.kind == 1 && c->nlists.num_neighbors == c->cfg.num_neighbors &&
                  !(c->cfg.flags & PCM_FLAG_NO_NEIGHBOUR_LISTS);
  return on ? view_of_lists(c->nlists) : TargetView{};
}

// lazy (re)build of everything the residual kernel needs
int prepare(pcm_ctx* c) {
  if (c->src.n == 0 || c->tgt.n == 0) { c->err = "align before setInputSource/setInputTarget"; return PCM_ERR_NO_INPUT; }
  HIPCK(c, hipSetDevice(c->device));
  const int mode = coord_mode_for(c->cfg.model);
  const bool gauss = is_ndt(c->cfg.model);
  const bool gicp = is_gicp(c->cfg.model);
  if (!c->map.valid || c->map.res != c->cfg.voxel_resolution || c->map.coord_mode != mode || (gauss && !c->map.gvox) || (gicp && !c->map.order)) {
    uint32_t n_log = (uint32_t)c->tgt.n;
    // the sliding-map capacity belongs to the iVox of the P2PLANE / LIO path; fast_gicp keeps every target point
    const uint32_t capacity = c->cfg.model == PCM_MODEL_P2PLANE ? (uint32_t)std::max(0, c->cfg.map_capacity) : 0u;
    // a map whose log only grew since its last build (pcm_target_insert / pcm_map_incremental) is updated: the new points are merged
    // into the sorted index it kept (voxel_hash.hip); anything else is built from scratch
    uint32_t hazards = 0;
    int rc = build_target_map(c->stream, c->tgt.d_pts, &n_log, c->cfg.voxel_resolution, mode, gauss, capacity, &c->map, &c->err, gicp, c->map.index_n, &hazards);
    c->stats.lru_batch_hazards += hazards;
    c->tgt.n = n_log;   // LRU eviction compacts the point log
    if (rc != PCM_OK) return rc;
    c->stats.target_voxels = c->map.num_voxels;
    c->stats.target_slots = c->map.cap;
    c->tgt_cov_valid = false;
    c->pleaf_valid = false;
    c->nlists.valid = false;
    c->nlists_failed = false;
    c->map_uses = 0;
  }
  if (c->map_uses < 1000000) c->map_uses++;
  if (uses_neighbour_lists(c) && (!c->nlists.valid || c->nlists.kind != 0 || c->nlists.num_neighbors != c->cfg.num_neighbors)) {
    // the candidate list of every voxel a query can fall into, built once per (static) target
    int rc = build_neighbour_lists(c->stream, c->map, c->cfg.num_neighbors, &c->nlists, &c->err);
    if (rc != PCM_OK) {
      if (c->cfg.flags & PCM_FLAG_NEIGHBOUR_LISTS) return rc;   // asked for explicitly
      c->nlists_failed = true;                                   // e.g. no memory for them: the tile kernel serves this target
      c->nlists.release();
      c->err.clear();
      (void)hipGetLastError();
    }
  }
  if (uses_voxel_slot_lists(c) && (!c->nlists.valid || c->nlists.kind != 2 || c->nlists.num_neighbors != c->cfg.num_neighbors)) {
    // fast_gicp NDTCuda / VGICP_CUDA with DIRECT1 / 7 / 27: rows of neighbour voxel indices, same policy as the candidate lists
    int rc = build_neighbour_lists(c->stream, c->map, c->cfg.num_neighbors, &c->nlists, &c->err, nullptr, true);
    if (rc != PCM_OK) {
      if (c->cfg.flags & PCM_FLAG_NEIGHBOUR_LISTS) return rc;
      c->nlists_failed = true;
      c->nlists.release();
      c->err.clear();
      (void)hipGetLastError();
    }
  }
  if (c->cfg.model == PCM_MODEL_NDT_OMP) {
    // VoxelGridCovariance leaves (NormalDistributionsTransform::init, ndt_omp.h:300-306) + pass buffers
    if (!c->pleaf_valid) {
      if (c->pleaf_cap < c->map.num_voxels) {
        if (c->pleaf) hipFree(c->pleaf);
        if (c->pleaf_f) hipFree(c->pleaf_f);
        c->pleaf = nullptr; c->pleaf_f = nullptr; c->pleaf_cap = 0;
        HIPCK(c, hipMalloc(&c->pleaf, sizeof(PclLeaf) * (size_t)c->map.num_voxels));
        HIPCK(c, hipMalloc(&c->pleaf_f, sizeof(PclLeafF) * (size_t)c->map.num_voxels));
        c->pleaf_cap = c->map.num_voxels;
      }
      int rc = build_pclndt_leaves(c->stream, c->map, c->pleaf, c->pleaf_f, &c->err);
      if (rc != PCM_OK) return rc;
      c->pleaf_valid = true;
      c->nlists.valid = false;
    }
    // neighbour-leaf lists of the grid (neighbour_lists.hip): same policy as the point-to-plane candidate lists -- from the second
    // registration against the target on, or with the grid when PCM_FLAG_NEIGHBOUR_LISTS asks for it
    const bool want_lists = !c->nlists_failed && !(c->cfg.flags & PCM_FLAG_NO_NEIGHBOUR_LISTS) && ((c->cfg.flags & PCM_FLAG_NEIGHBOUR_LISTS) || c->map_uses >= 2);
    if (want_lists && (!c->nlists.valid || c->nlists.kind != 1 || c->nlists.num_neighbors != c->cfg.num_neighbors)) {
      int rc = build_neighbour_lists(c->stream, c->map, c->cfg.num_neighbors, &c->nlists, &c->err, c->pleaf);
      if (rc != PCM_OK) {
        if (c->cfg.flags & PCM_FLAG_NEIGHBOUR_LISTS) return rc;
        c->nlists_failed = true;
        c->nlists.release();
        c->err.clear();
        (void)hipGetLastError();
      }
    }
    uint32_t per = 0;
    const size_t need = (size_t)pclndt_workgroups((uint32_t)c->src.n, &per) * 48;
    if (c->ndt_partials_cap < need) {
      if (c->ndt_partials) hipFree(c->ndt_partials);
      c->ndt_partials = nullptr; c->ndt_partials_cap = 0;
      HIPCK(c, hipMalloc(&c->ndt_partials, sizeof(double) * need));
      HIPCK(c, hipMemsetAsync(c->ndt_partials, 0, sizeof(double) * need, c->stream));
      c->ndt_partials_cap = need;
    }
    if (!c->ndt_out) HIPCK(c, hipMalloc(&c->ndt_out, sizeof(double) * 48));
    if (!c->ndt_out_host) HIPCK(c, hipHostMalloc(&c->ndt_out_host, sizeof(double) * 48));
    return PCM_OK;
  }
  if (gicp) {
    // FastGICP::computeTransformation: covariances of both clouds, lazily   fast_gicp_impl.hpp:102-110
    const bool rbf = c->cfg.model == PCM_MODEL_VGICP_CUDA && c->cfg.covariance_method == PCM_COV_RBF_KERNEL;   // NearestNeighborMethod::GPU_RBF_KERNEL
    const float rbf_w = rbf ? c->cfg.rbf_kernel_width : -1.f, rbf_d = rbf ? c->cfg.rbf_max_dist : -1.f;
    if (c->cov_k != c->cfg.k_correspondences || c->cov_reg != c->cfg.regularization + 100 * c->cfg.model || c->cov_vmode != c->cfg.voxel_mode || c->cov_rbf_w != rbf_w ||
        c->cov_rbf_d != rbf_d) {
      c->src_cov_valid = false; c->tgt_cov_valid = false;
      c->cov_k = c->cfg.k_correspondences; c->cov_reg = c->cfg.regularization + 100 * c->cfg.model; c->cov_vmode = c->cfg.voxel_mode;
      c->cov_rbf_w = rbf_w; c->cov_rbf_d = rbf_d;
    }
    // the scan's own grid is only the index of its kNN search: a finer cell keeps the candidate lists short where a
    // LiDAR scan is dense (near the sensor one 0.5 m voxel holds thousands of points)
    // (measured on Livox-shaped 100 k-point scans: 0.5 m cells are the optimum; 1.0 m costs 25 %, 0.25 m 15-40 %)
    const float src_res = std::min(c->cfg.voxel_resolution, 0.5f);
    if (!c->srcmap.valid || c->srcmap.res != src_res || c->srcmap.coord_mode != mode) {
      uint32_t n_src = (uint32_t)c->src.n;
      // sub-voxel order: 64 consecutive points of the brick-major scan are one patch (k_covariances; k_gicp reads it point by point)
      int rc = build_target_map(c->stream, c->src.d_pts, &n_src, src_res, mode, false, 0u, &c->srcmap, &c->err, true, 0u, nullptr, cov_subsort_enabled());
      if (rc != PCM_OK) return rc;
      c->src_cov_valid = false;
    }
    if (!c->tgt_cov_valid) {
      if (c->tgt_cov_cap < c->map.num_points) {
        if (c->tgt_cov) hipFree(c->tgt_cov);
        c->tgt_cov = nullptr; c->tgt_cov_cap = 0;
        HIPCK(c, hipMalloc(&c->tgt_cov, sizeof(double) * 6 * (size_t)c->map.num_points));
        c->tgt_cov_cap = c->map.num_points;
      }
      const int reg_code = c->cfg.regularization + (c->cfg.model == PCM_MODEL_VGICP_CUDA ? 16 : 0);   // + 16: float CUDA-core semantics
      // `if (target_covs_.size() != target_->size()) calculate_covariances(...)`  fast_gicp_impl.hpp:107-109
      const bool given = c->cfg.model != PCM_MODEL_VGICP_CUDA && c->user_cov[1].size() == (size_t)c->map.num_points * 6 && c->map.num_points == c->tgt.n;
      int rc = PCM_OK;
      if (given) rc = upload_covariances(c->stream, c->map, c->user_cov[1].data(), c->tgt_cov, &c->err);
      else if (rbf) rc = compute_covariances_rbf(c->stream, c->map, c->tgt.d_pts, (uint32_t)c->tgt.n, c->cfg.rbf_kernel_width, c->cfg.rbf_max_dist, c->cfg.regularization, c->tgt_cov, &c->err);
      else {
        uint32_t n_fine = (uint32_t)c->tgt.n;   // the fine kNN index (see the source cloud below); only when the map holds the whole log
        const bool fine = cov_fine_index_enabled() && c->map.num_points == c->tgt.n &&
                          build_target_map(c->stream, c->tgt.d_pts, &n_fine, std::min(c->cfg.voxel_resolution, 0.5f) * 0.125f, mode, false, 0u, &c->covfine, &c->err, true) == PCM_OK;
        rc = compute_covariances(c->stream, c->map, c->cfg.k_correspondences, reg_code, c->tgt_cov, &c->err, fine ? &c->covfine : nullptr);
      }
      if (rc != PCM_OK) return rc;
      if (c->cfg.model == PCM_MODEL_VGICP_CUDA) {
        if (c->cvox_cap < c->map.num_voxels) {
          if (c->cvox) hipFree(c->cvox);
          c->cvox = nullptr; c->cvox_cap = 0;
          HIPCK(c, hipMalloc(&c->cvox, sizeof(VgcVoxel) * (size_t)c->map.num_voxels));
          c->cvox_cap = c->map.num_voxels;
        }
        rc = build_vgc_voxels(c->stream, c->map, c->tgt_cov, c->cvox, &c->err);
        if (rc != PCM_OK) return rc;
      }
      if (c->cfg.model == PCM_MODEL_VGICP) {
        if (c->vvox_cap < c->map.num_voxels) {
          if (c->vvox) hipFree(c->vvox);
          c->vvox = nullptr; c->vvox_cap = 0;
          HIPCK(c, hipMalloc(&c->vvox, sizeof(VgVoxel) * (size_t)c->map.num_voxels));
          c->vvox_cap = c->map.num_voxels;
        }
        rc = build_vgicp_voxels(c->stream, c->map, c->tgt_cov, c->cfg.voxel_mode, c->vvox, &c->err);
        if (rc != PCM_OK) return rc;
      }
      c->tgt_cov_valid = true;
    }
    if (!c->src_cov_valid) {
      if (c->src_cov_cap < c->srcmap.num_points) {
        if (c->src_cov) hipFree(c->src_cov);
        c->src_cov = nullptr; c->src_cov_cap = 0;
        HIPCK(c, hipMalloc(&c->src_cov, sizeof(double) * 6 * (size_t)c->srcmap.num_points));
        c->src_cov_cap = c->srcmap.num_points;
      }
      const bool given = c->cfg.model != PCM_MODEL_VGICP_CUDA && c->user_cov[0].size() == (size_t)c->srcmap.num_points * 6 && c->srcmap.num_points == c->src.n;   // :104-106
      int rc = PCM_OK;
      if (given) rc = upload_covariances(c->stream, c->srcmap, c->user_cov[0].data(), c->src_cov, &c->err);
      else if (rbf) rc = compute_covariances_rbf(c->stream, c->srcmap, c->src.d_pts, (uint32_t)c->src.n, c->cfg.rbf_kernel_width, c->cfg.rbf_max_dist, c->cfg.regularization, c->src_cov, &c->err);
      else {
        // a second index of the scan on a grid 8x finer: where one voxel of the search grid holds hundreds of points (a LiDAR's near
        // field) the 20 nearest lie within a few centimetres, and a candidate box made of 0.5 m voxels is thousands of points
        uint32_t n_fine = (uint32_t)c->src.n;
        const bool fine = cov_fine_index_enabled() && build_target_map(c->stream, c->src.d_pts, &n_fine, src_res * 0.125f, mode, false, 0u, &c->covfine, &c->err, true) == PCM_OK;
        rc = compute_covariances(c->stream, c->srcmap, c->cfg.k_correspondences, c->cfg.regularization + (c->cfg.model == PCM_MODEL_VGICP_CUDA ? 16 : 0), c->src_cov, &c->err,
                                 fine ? &c->covfine : nullptr);
      }
      if (rc != PCM_OK) return rc;
      c->src_cov_valid = true;
    }
    const size_t ncorr = c->cfg.model == PCM_MODEL_VGICP_CUDA ? 0 : c->src.n * (size_t)(c->cfg.model == PCM_MODEL_VGICP ? c->cfg.num_neighbors : 1);
    if (c->maha_cap < ncorr) {
      if (c->maha) hipFree(c->maha);
      c->maha = nullptr; c->maha_cap = 0;
      HIPCK(c, hipMalloc(&c->maha, sizeof(double) * 6 * ncorr));
      c->maha_cap = ncorr;
    }
  }
  if (c->cfg.model == PCM_MODEL_NDT_D2D && (!c->srcmap.valid || c->srcmap.res != c->cfg.voxel_resolution)) {
    // D2D: the source elements are the source-voxel distributions (ndt_cuda.cu:120-129,156-158)
    uint32_t n_src = (uint32_t)c->src.n;
    int rc = build_target_map(c->stream, c->src.d_pts, &n_src, c->cfg.voxel_resolution, mode, true, 0u, &c->srcmap, &c->err);
    if (rc != PCM_OK) return rc;
  }
  if (gauss || gicp) {
    const size_t need = num_elements(c) * (c->cfg.model == PCM_MODEL_GICP ? (size_t)1 : neighbor_slots(c->cfg));
    if (c->corr_cap < need) {
      if (c->corr) hipFree(c->corr);
      c->corr = nullptr; c->corr_cap = 0;
      HIPCK(c, hipMalloc(&c->corr, sizeof(int32_t) * need));
      c->corr_cap = need;
    }
  }
  if (c->cfg.sort_source && c->src_order_cap < c->src.n) {
    if (c->src_order) hipFree(c->src_order);
    c->src_order = nullptr; c->src_order_cap = 0; c->src_sorted = false;
    HIPCK(c, hipMalloc(&c->src_order, sizeof(float4) * c->src.n));
    c->src_order_cap = c->src.n;
  }
  if (!c->counter) {
    HIPCK(c, hipMalloc(&c->counter, sizeof(unsigned int)));
    HIPCK(c, hipMemsetAsync(c->counter, 0, sizeof(unsigned int), c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
  }
  if (c->nn_cap < c->src.n) {
    if (c->nn) hipFree(c->nn);
    c->nn = nullptr; c->nn_cap = 0;
    HIPCK(c, hipMalloc(&c->nn, sizeof(uint32_t) * 5 * c->src.n));
    c->nn_cap = c->src.n;
  }
  if (c->planes_cap < c->src.n) {
    if (c->planes) hipFree(c->planes);
    c->planes = nullptr; c->planes_cap = 0;
    HIPCK(c, hipMalloc(&c->planes, sizeof(float4) * c->src.n));
    c->planes_cap = c->src.n;
  }
  return PCM_OK;
}

struct Geom {
  int blocks_per_pair;   // residual/reduction kernel
  int points_per_block;
  int tiles_per_pair;    // correspondence-search kernel (256-point tiles)
};

Geom pick_geom(size_t max_n, int npairs, bool ndt = false) {
  // residual kernel: streaming 32 B/point; each lane amortises the 29-value wave
  // reduction over several points, but keep >= ~1024 workgroups in flight
  size_t total = max_n * (size_t)npairs;
  size_t ppb = (total / 1024 + 255) / 256 * 256;
  ppb = std::min<size_t>(std::max<size_t>(ppb, 256), 2048);
  Geom g;
  g.points_per_block = (int)ppb;
  g.blocks_per_pair = (int)((max_n + ppb - 1) / ppb);
  g.tiles_per_pair = ndt ? g.blocks_per_pair : (int)((max_n + 255) / 256);   // NDT linearize uses the streaming geometry
  return g;
}

void fill_desc(const pcm_ctx* c, PairDesc* d, double* partials) {
  d->tgt.pts = c->map.pts;
  d->tgt.vox_start = c->map.vox_start;
  d->tgt.bricks = c->map.bricks;
  d->tgt.bmask = c->map.bmask;
  d->tgt.bpref = c->map.bpref;
  d->tgt.mask = c->map.cap - 1;
  d->tgt.num_points = c->map.num_points;
  d->tgt.inv_res = c->map.inv_res;
  d->tgt.res = c->map.res;
  d->tgt.gvox = c->map.gvox;
  d->nl = TargetView{};
  if (c->nlists.valid && c->nlists.num_neighbors == c->cfg.num_neighbors &&
      ((uses_neighbour_lists(c) && c->nlists.kind == 0) || (uses_voxel_slot_lists(c) && c->nlists.kind == 2)))
    d->nl = view_of_lists(c->nlists);
  d->src.pts = (c->cfg.sort_source && c->src_sorted) ? c->src_order : c->src.d_pts;
  if (is_gicp(c->cfg.model)) d->src.pts = c->srcmap.pts;   // brick-major copy of the scan: its covariances are in that order
  d->src_cov = c->src_cov;
  d->tgt_cov = c->tgt_cov;
  d->vvox = c->vvox;
  d->cvox = c->cvox;
  d->maha = c->maha;
  d->src.gvox = c->srcmap.gvox;
  d->src.num_points = (uint32_t)num_elements(c);
  d->corr = c->corr;
  d->nn = c->nn;
  d->planes = c->planes;
  d->partials = partials;
  d->counter = c->counter;
}

KernelParams kernel_params(const pcm_config& g, const Geom& geom) {
  KernelParams kp{};
  kp.num_neighbors = g.num_neighbors;
  if (radius_model(g.model) && g.neighbor_search_radius > 0.f) {
    kp.nb_range = (int32_t)std::ceil((double)g.neighbor_search_radius);
    kp.nb_radius = (double)g.neighbor_search_radius;
  }
  kp.knn = g.knn;
  kp.min_knn = g.min_knn;
  {  // d2 < fl  <=>  double(d2) < max_range^2  when fl is the smallest float >= max_range^2
    const double m2 = (double)g.max_range * (double)g.max_range;
    float fl = (float)m2;
    if ((double)fl < m2) fl = nextafterf(fl, INFINITY);
    kp.max_range_sq = fl;
  }
  kp.plane_threshold = g.plane_threshold;
  kp.blocks_per_pair = geom.blocks_per_pair;
  kp.points_per_block = geom.points_per_block;
  kp.tiles_per_pair = geom.tiles_per_pair;
  kp.use_lds = (g.flags & PCM_FLAG_NO_LDS_STAGING) ? 0 : 1;
  kp.do_step = 1;
  kp.lin_points_per_block = (is_ndt(g.model) || g.model == PCM_MODEL_VGICP_CUDA) ? geom.points_per_block : 256;
  kp.coord_mode = coord_mode_for(g.model);
  kp.max_corr_sq = (double)g.max_corr_dist * (double)g.max_corr_dist;
  return kp;
}

LsqParams lsq_params(const pcm_config& g) {
  LsqParams lp{};
  lp.optimizer = g.optimizer;
  lp.max_iterations = g.max_iterations;
  lp.lm_max_iterations = g.lm_max_iterations;
  lp.rotation_eps = g.rotation_eps;
  lp.translation_eps = g.translation_eps;
  lp.lm_init_lambda_factor = g.lm_init_lambda_factor;
  return lp;
}

bool same_solver_config(const pcm_config& a, const pcm_config& b) {
  return a.model == b.model && a.optimizer == b.optimizer && a.max_iterations == b.max_iterations && a.lm_max_iterations == b.lm_max_iterations &&
         a.rotation_eps == b.rotation_eps && a.translation_eps == b.translation_eps && a.lm_init_lambda_factor == b.lm_init_lambda_factor &&
         a.num_neighbors == b.num_neighbors && a.neighbor_search_radius == b.neighbor_search_radius && a.max_range == b.max_range && a.plane_threshold == b.plane_threshold && a.flags == b.flags &&
         (a.model == PCM_MODEL_P2PLANE || a.voxel_resolution == b.voxel_resolution) && (!is_gicp(a.model) || a.max_corr_dist == b.max_corr_dist);
}

int align_batch_impl(pcm_ctx* const* ctxs, int n, const float* guesses, pcm_result* host_out, void* device_out) {
  if (!ctxs || n <= 0 || !guesses) return PCM_ERR_INVALID_ARGUMENT;
  pcm_ctx* c0 = ctxs[0];
  if (!c0) return PCM_ERR_INVALID_ARGUMENT;
  size_t max_n = 0;
  for (int i = 0; i < n; i++) {
    pcm_ctx* c = ctxs[i];
    if (!c) { c0->err = "null context in batch"; return PCM_ERR_INVALID_ARGUMENT; }
    if (c->device != c0->device) { c0->err = "all contexts of a batch must live on one device"; return PCM_ERR_INVALID_ARGUMENT; }
    if (!same_solver_config(c->cfg, c0->cfg)) { c0->err = "all contexts of a batch must share the solver configuration"; return PCM_ERR_INVALID_ARGUMENT; }
    if (c->stream != c0->stream) {
      // inputs of the other contexts were produced on their own streams, which are idle after set_*()
    }
  }
  {
    // per-object preparation (the scan's kNN index and covariances of the GICP family are the heavy part: a radix sort with
    // host syncs per object): the objects are independent and own their streams, so up to 8 host threads prepare them side
    // by side; nothing is shared but the device
    std::vector<int> rcs((size_t)n, PCM_OK);
    const int nthreads = (is_gicp(c0->cfg.model) && n > 1) ? std::min(n, 8) : 1;
    bool distinct = true;
    for (int i = 0; i < n && distinct; i++) for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) { distinct = false; break; }
    if (nthreads <= 1 || !distinct) {
      for (int i = 0; i < n; i++) rcs[(size_t)i] = prepare(ctxs[i]);
    } else {
      std::atomic<int> next{0};
      auto worker = [&]() {
        for (;;) {
          const int i = next.fetch_add(1);
          if (i >= n) break;
          rcs[(size_t)i] = prepare(ctxs[i]);
        }
      };
      std::vector<std::thread> th;
      for (int t = 0; t < nthreads; t++) th.emplace_back(worker);
      for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; i++) {
      if (rcs[(size_t)i] != PCM_OK) { if (ctxs[i] != c0) c0->err = ctxs[i]->err; return rcs[(size_t)i]; }
      max_n = std::max(max_n, num_elements(ctxs[i]));
    }
  }
  // GICP / VGICP: the covariance kernels of the contexts were queued on their own streams without a host
  // sync (they overlap on the device); the batch kernels below run on c0's stream and read their output
  if (is_gicp(c0->cfg.model)) {
    for (int i = 0; i < n; i++) HIPCK(c0, hipStreamSynchronize(ctxs[i]->stream));
  }
  const pcm_config& g = c0->cfg;
  const bool ndt = is_ndt(g.model) || g.model == PCM_MODEL_VGICP_CUDA;   // residual kernel of the Gaussian-voxel family
  const bool gicp = is_gicp(g.model);                                     // per-point covariances, source elements = brick-major copy
  const Geom geom = pick_geom(max_n, n, ndt);
  const LsqParams lp = lsq_params(g);
  const KernelParams kp = kernel_params(g, geom);
  // worst case: every outer iteration = 1 linearize + lm_max_iterations trials
  // batch window: at most `window` pairs iterate at a time, a finished pair's slot goes to the next queued one on the
  // device (k_finish_round) -- the late rounds of a slow pair then overlap the early rounds of its successors
  const int window = (g.batch_window > 0 && g.max_iterations > 0) ? std::min(n, g.batch_window) : n;
  const int per_pair_rounds = std::max(1, g.max_iterations) * (g.optimizer == PCM_OPT_LEVENBERG_MARQUARDT ? 1 + std::max(1, g.lm_max_iterations) : 1);
  // Up to 256 pairs with a window of at most 64: the window is kept by the HOST.  Only the pairs of the launch list run; when the
  // status byte of a pair says "done" the next queued pair takes its place in the list (it was initialised with the others and
  // simply never launched before).  Every round then carries about `window` live pairs, so the fixed cost of a round (two launches,
  // their boundaries) is shared by that many registrations for the whole batch, not only in its first rounds.  No device-side hand-off.
  const bool host_window = window < n && n <= 256 && window <= kMaxListedPairs;
  // Round budget.  Host window: a slot serves its pairs one after the other, and every hand-over costs one extra round because the
  // status bytes are read one round behind; with every pair running to max_iterations a slot needs ceil(n / window) * (rounds + 1)
  // rounds, one more pair's worth covers an uneven hand-out (round-2 advisor finding: the old bound ran out for 256 pairs at a
  // window of 8 and 10 GN iterations and returned unfinished pairs as PCM_OK).  A pair the loop leaves unfinished is reported
  // with PCM_ERR_INTERNAL by k_pack_results, never silently.
  const int max_rounds = host_window ? ((n + window - 1) / window + 1) * (per_pair_rounds + 1) + 2
                                     : per_pair_rounds * (n - window + 1) + 1 + 2 * (n - window);   // + the rounds a handed-over pair spends PENDING
  const size_t per_pair_partials = (size_t)std::max(geom.blocks_per_pair, geom.tiles_per_pair) * kPartialStride;
  Workspace* w = nullptr;
  int rc = ensure_ws(c0, &w, n, per_pair_partials * n, max_rounds);
  if (rc != PCM_OK) return rc;
  hipStream_t st = c0->stream;

  HIPCK(c0, hipMemcpyAsync(w->d_guesses, guesses, sizeof(float) * 16 * n, hipMemcpyHostToDevice, st));
  {  // new scans are re-ordered along the world grid (at their initial guess) in one batched pass
    std::vector<SortJob> jobs;
    uint32_t total = 0, jmax = 0;
    for (int i = 0; i < n; i++) {
      pcm_ctx* c = ctxs[i];
      if (!c->cfg.sort_source || c->src_sorted || c->cfg.model == PCM_MODEL_NDT_D2D || gicp) continue;
      SortJob j{c->src.d_pts, c->src_order, (uint32_t)c->src.n, total, (uint32_t)i, 0};
      jobs.push_back(j);
      total += j.n;
      jmax = std::max(jmax, j.n);
    }
    if (!jobs.empty()) {
      HIPCK(c0, hipMemcpyAsync(w->d_jobs, jobs.data(), sizeof(SortJob) * jobs.size(), hipMemcpyHostToDevice, st));
      rc = sort_sources_batched(st, w->d_jobs, (int)jobs.size(), jmax, total, w->d_guesses, g.voxel_resolution, &w->sort, &c0->err);
      if (rc != PCM_OK) return rc;
      for (const SortJob& j : jobs) ctxs[j.guess_index]->src_sorted = true;
    }
  }
  std::vector<PairDesc> descs(n);
  for (int i = 0; i < n; i++) fill_desc(ctxs[i], &descs[i], w->d_partials + per_pair_partials * i);
  HIPCK(c0, hipMemcpyAsync(w->d_descs, descs.data(), sizeof(PairDesc) * n, hipMemcpyHostToDevice, st));
  std::memset(w->h_flags, 0, (size_t)max_rounds * n);
  launch_init_states(st, w->d_states, w->d_guesses, n, g.max_iterations, host_window ? n : window, w->d_queue);
  const bool stats_on = (c0->profiling & 1) != 0;      // HIP events around the residual launches
  const bool stats_sampled = stats_on && (c0->profiling & 8) != 0;   // ... every 4th launch only; the phase moves from batch to batch
  const unsigned prof_phase = stats_sampled ? (unsigned)(c0->stats.linearize_launches & 3u) : 0u;
  uint64_t timed_launches = 0, timed_slots = 0, launched_slots = 0;
  const bool timing_on = (c0->profiling & 4) != 0;     // diagnostic: in-kernel phase stamps (stats.phase_cycles)
  const bool counters_on = (c0->profiling & 2) != 0 || timing_on;   // kNN candidate / probe counters (slower kernel variant)
  if (counters_on) HIPCK(c0, hipMemsetAsync(w->d_stats, 0, sizeof(unsigned long long) * 16, st));
  const bool is_lm = g.optimizer == PCM_OPT_LEVENBERG_MARQUARDT;
  const bool write_sel = is_lm;  // trial passes re-use the planes of the selected set
  const bool counted_search = (g.flags & PCM_FLAG_COUNTED_SEARCH) != 0;   // k_linearize_counted (A/B)
  const bool ref_order = g.model == PCM_MODEL_P2PLANE && (g.flags & PCM_FLAG_REFERENCE_KNN_ORDER) != 0;   // neighbours in libstdc++'s nth_element order
  bool lists = g.model == PCM_MODEL_P2PLANE && !ref_order;   // k_linearize_lists: every context of the batch holds its map's candidate lists
  for (int i = 0; i < n && lists; i++) lists = uses_neighbour_lists(ctxs[i]) && ctxs[i]->nlists.valid && ctxs[i]->nlists.kind == 0;
  if (ref_order) {
    for (int i = 0; i < n; i++) {
      if (ctxs[i]->map.max_voxel_points > (uint32_t)kRefMaxVoxelPoints) {
        c0->err = "PCM_FLAG_REFERENCE_KNN_ORDER supports at most " + std::to_string(kRefMaxVoxelPoints) + " points per voxel (this map: " + std::to_string(ctxs[i]->map.max_voxel_points) + ")";
        return PCM_ERR_UNSUPPORTED;
      }
    }
  }

  int rounds_done = 0;
  size_t prof_used = 0;
  // Only the pairs the host still believes active are launched (the list rides in the kernel arguments, batches of
  // <= kMaxListedPairs pairs): an early-exit workgroup is not free, and the late rounds of a batch have one or two live pairs.
  // The list lags one round (the status bytes are read one round behind); a stale entry exits at once.
  const bool use_list = (n <= kMaxListedPairs && window == n) || host_window;
  std::vector<uint8_t> act((size_t)(host_window ? window : n));
  for (size_t i = 0; i < act.size(); i++) act[i] = (uint8_t)i;
  int next_queued = host_window ? window : n;   // host window: first pair that has not been launched yet
  std::vector<uint8_t> prev_list;
  KernelParams kpr = kp;
  for (int r = 0; r < max_rounds; r++) {
    const int nl = use_list ? (int)act.size() : n;
    if (use_list) {
      kpr.use_list = 1;
      std::memcpy(kpr.active, act.data(), act.size());
    }
    const bool timed = stats_on && (!stats_sampled || (((unsigned)r + prof_phase) & 3u) == 0u);
    launched_slots += (uint64_t)nl;
    if (timed) {
      while (w->ev_prof.size() < prof_used + 3) { hipEvent_t e; HIPCK(c0, hipEventCreate(&e)); w->ev_prof.push_back(e); }
      HIPCK(c0, hipEventRecord(w->ev_prof[prof_used], st));
      timed_launches++;
      timed_slots += (uint64_t)nl;
    }
    // per round: correspondence search + residual/Jacobian + reduction in one launch, then the tiny
    // per-pair sum + GN/LM step launch.  LM adds the (cheap) trial-cost launch + its step.
    // PCM_FLAG_FUSED_STEP (off by default): the last workgroup of a pair's search launch takes the GN step (write-through hand-off of
    // the partial rows, kernels.hip).  Measured slower than the second launch at every round size, the single-pair rounds
    // included (profiles/r02_fused_step_threshold_sweep.txt): every workgroup pays a store drain and a returned atomic.
    const bool fuse = use_list && !is_lm && !ndt && !gicp && !counters_on && !timing_on && kp.do_step && (g.flags & PCM_FLAG_FUSED_STEP);
    if (ndt) launch_ndt(st, w->d_descs, w->d_states, kpr, nl, ndt_kind(g.model), false);
    else if (gicp) launch_gicp(st, w->d_descs, w->d_states, kpr, nl, g.model == PCM_MODEL_VGICP, false);
    else if (fuse) launch_linearize_fused(st, w->d_descs, w->d_states, kpr, lp, nl, w->d_flags + (size_t)r * n);
    else if (ref_order) launch_linearize_reforder(st, w->d_descs, w->d_states, kpr, nl, write_sel);
    else if (lists && !counters_on && !timing_on) launch_linearize_lists(st, w->d_descs, w->d_states, kpr, nl, write_sel);
    else if (counted_search) launch_linearize_counted(st, w->d_descs, w->d_states, kpr, nl, write_sel, counters_on ? w->d_stats : nullptr, timing_on);
    else launch_linearize(st, w->d_descs, w->d_states, kpr, nl, write_sel, counters_on ? w->d_stats : nullptr, timing_on);
    if (timed) HIPCK(c0, hipEventRecord(w->ev_prof[prof_used + 1], st));
    if (!fuse) launch_finish_round(st, w->d_descs, w->d_states, kpr, lp, nl, false, !is_lm, w->d_flags + (size_t)r * n, w->d_sums, use_list ? nullptr : w->d_queue, n);
    if (is_lm) {
      if (ndt) launch_ndt(st, w->d_descs, w->d_states, kpr, nl, ndt_kind(g.model), true);
      else if (gicp) launch_gicp(st, w->d_descs, w->d_states, kpr, nl, g.model == PCM_MODEL_VGICP, true);
      else launch_trial(st, w->d_descs, w->d_states, kpr, nl);
      launch_finish_round(st, w->d_descs, w->d_states, kpr, lp, nl, true, true, w->d_flags + (size_t)r * n, w->d_sums, use_list ? nullptr : w->d_queue, n);
    }
    if (timed) {
      if (!stats_sampled) HIPCK(c0, hipEventRecord(w->ev_prof[prof_used + 2], st));
      prof_used += 3;
    }
    rounds_done = r + 1;
    std::vector<uint8_t> this_list = act;   // pairs launched in round r
    if (r >= 1) {  // look one round behind so the GPU always has the next round queued
      volatile unsigned char* row = w->h_flags + (size_t)(r - 1) * n;
      bool any_active = false;
      const auto t_start = std::chrono::steady_clock::now();
      std::vector<uint8_t> alive;
      const int np = use_list ? (int)prev_list.size() : n;
      for (int k = 0; k < np; k++) {
        const int i = use_list ? (int)prev_list[(size_t)k] : k;
        if (int wrc = wait_status_byte(c0, st, row + i, t_start)) return wrc;
        any_active |= row[i] == 1;
        if (row[i] == 1 && use_list) alive.push_back((uint8_t)i);
      }
      if (host_window) {
        // pairs launched in round r (this_list) but not in round r - 1 have no status byte yet: they stay
        for (uint8_t i : this_list) {
          bool seen = false;
          for (uint8_t q : prev_list) if (q == i) { seen = true; break; }
          if (!seen) { alive.push_back(i); any_active = true; }
        }
        while ((int)alive.size() < window && next_queued < n) { alive.push_back((uint8_t)next_queued++); any_active = true; }
      }
      if (!any_active) break;
      if (use_list) act.swap(alive);
    }
    prev_list.swap(this_list);
  }
  HIPCK(c0, hipGetLastError());
  pcm_result* d_res = device_out ? static_cast<pcm_result*>(device_out) : w->d_results;
  launch_pack_results(st, w->d_states, d_res, n);
  std::vector<pcm_result> tmp;
  pcm_result* h_res = host_out;
  if (!h_res) { tmp.resize(n); h_res = tmp.data(); }
  HIPCK(c0, hipMemcpyAsync(h_res, d_res, sizeof(pcm_result) * n, hipMemcpyDeviceToHost, st));
  HIPCK(c0, hipStreamSynchronize(st));

  if (counters_on) {
    unsigned long long hs[16];
    HIPCK(c0, hipMemcpy(hs, w->d_stats, sizeof(hs), hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; k++) c0->phase_cycles[k] += hs[8 + k];
    c0->stats.candidates += hs[0];
    c0->stats.slots_probed += hs[1];
    c0->stats.tiles += hs[3];
    c0->stats.tiles_lds_grid += hs[4];
    c0->stats.tiles_lds_points += hs[2];
  }
  if (stats_on) {
    double ms = 0.0, ms2 = 0.0;
    for (size_t k = 0; k + 2 < prof_used + 1; k += 3) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, w->ev_prof[k], w->ev_prof[k + 1]) == hipSuccess) ms += t;
      if (!stats_sampled && hipEventElapsedTime(&t, w->ev_prof[k + 1], w->ev_prof[k + 2]) == hipSuccess) ms2 += t;
    }
    c0->stats.linearize_ms += ms;
    c0->stats.residual_ms += ms2;
    c0->stats.timed_launches += timed_launches;
    c0->stats.timed_pair_slots += timed_slots;
  }
  c0->stats.launched_pair_slots += launched_slots;
  c0->stats.linearize_launches += (uint64_t)rounds_done;
  uint64_t passes = 0;
  int worst = PCM_OK;
  for (int i = 0; i < n; i++) {
    passes += (uint64_t)(h_res[i].num_linearize + h_res[i].num_compute_error) * num_elements(ctxs[i]);
    if (h_res[i].status != PCM_OK && worst != PCM_ERR_INTERNAL) worst = h_res[i].status;
  }
  c0->stats.point_passes += passes;
  if (worst == PCM_ERR_INTERNAL) c0->err = "the round budget of the batch ran out before every pair finished (library bug): unfinished pairs carry PCM_ERR_INTERNAL";
  else if (worst != PCM_OK) c0->err = "lm not converged!!";
  return worst;
}

// one LINEARIZE or TRIAL pass at a caller-supplied pose (parity hook)
int single_pass(pcm_ctx* c, const double T[16], bool linearize, double sums[kPartialStride]) {
  if (c->cfg.model == PCM_MODEL_NDT_OMP) { c->err = "the pclomp NDT model is evaluated through pcm_ndt_derivatives"; return PCM_ERR_UNSUPPORTED; }
  int rc = prepare(c);
  if (rc != PCM_OK) return rc;
  const bool ndt = is_ndt(c->cfg.model) || c->cfg.model == PCM_MODEL_VGICP_CUDA;
  const Geom geom = pick_geom(num_elements(c), 1, ndt);
  const KernelParams kp = kernel_params(c->cfg, geom);
  Workspace* w = nullptr;
  rc = ensure_ws(c, &w, 1, (size_t)std::max(geom.blocks_per_pair, geom.tiles_per_pair) * kPartialStride, 2);
  if (rc != PCM_OK) return rc;
  PairDesc d;
  fill_desc(c, &d, w->d_partials);
  PairState s;
  float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  init_state(s, ident);
  for (int i = 0; i < 16; i++) { s.x0[i] = T[i]; s.xi[i] = T[i]; }
  s.mode = linearize ? MODE_LINEARIZE : MODE_TRIAL;
  HIPCK(c, hipMemcpyAsync(w->d_descs, &d, sizeof(d), hipMemcpyHostToDevice, c->stream));
  HIPCK(c, hipMemcpyAsync(w->d_states, &s, sizeof(s), hipMemcpyHostToDevice, c->stream));
  KernelParams kp1 = kp;
  kp1.do_step = 0;   // the last workgroup exports the sums instead of stepping
  if (ndt) launch_ndt(c->stream, w->d_descs, w->d_states, kp1, 1, ndt_kind(c->cfg.model), !linearize);
  else if (is_gicp(c->cfg.model)) launch_gicp(c->stream, w->d_descs, w->d_states, kp1, 1, c->cfg.model == PCM_MODEL_VGICP, !linearize);
  else if (linearize && (c->cfg.flags & PCM_FLAG_REFERENCE_KNN_ORDER)) {
    if (c->map.max_voxel_points > (uint32_t)kRefMaxVoxelPoints) { c->err = "PCM_FLAG_REFERENCE_KNN_ORDER supports at most " + std::to_string(kRefMaxVoxelPoints) + " points per voxel"; return PCM_ERR_UNSUPPORTED; }
    launch_linearize_reforder(c->stream, w->d_descs, w->d_states, kp1, 1, true);
  }
  else if (linearize && uses_neighbour_lists(c) && c->nlists.valid && c->nlists.kind == 0) launch_linearize_lists(c->stream, w->d_descs, w->d_states, kp1, 1, true);
  else if (linearize && (c->cfg.flags & PCM_FLAG_COUNTED_SEARCH)) launch_linearize_counted(c->stream, w->d_descs, w->d_states, kp1, 1, true, nullptr);
  else if (linearize) launch_linearize(c->stream, w->d_descs, w->d_states, kp1, 1, true, nullptr, false);
  else launch_trial(c->stream, w->d_descs, w->d_states, kp1, 1);
  launch_finish_round(c->stream, w->d_descs, w->d_states, kp1, lsq_params(c->cfg), 1, !linearize, false, w->d_flags, w->d_sums);
  HIPCK(c, hipGetLastError());
  HIPCK(c, hipMemcpyAsync(sums, w->d_sums, sizeof(double) * kPartialStride, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  return PCM_OK;
}

// one pclomp NDT pass on the device: launch, read the 48-double row back (pass 0/1: H, g, score; pass 2: H)
int pclndt_eval(pcm_ctx* c, int pass, const NdtOmpParams& P, ndtomp::Eval* e, double gauss_d3 = 0.0) {
  launch_pclndt_pass(c->stream, c->map, c->pleaf, c->pleaf_f, ndt_lists_view(c), c->src.d_pts, (uint32_t)c->src.n, P, pass, c->ndt_partials, c->ndt_out, gauss_d3);
  HIPCK(c, hipGetLastError());
  HIPCK(c, hipMemcpyAsync(c->ndt_out_host, c->ndt_out, sizeof(double) * 48, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  std::memcpy(e->H, c->ndt_out_host, sizeof(double) * 36);
  if (pass == 3) e->score = c->ndt_out_host[0];
  else if (pass != 2) {
    std::memcpy(e->g, c->ndt_out_host + 36, sizeof(double) * 6);
    e->score = c->ndt_out_host[42];
  }
  c->stats.linearize_launches += 1;
  c->stats.point_passes += c->src.n;
  return PCM_OK;
}

auto make_ndt_solver(pcm_ctx* c) {
  auto ev = [c](int pass, const NdtOmpParams& P, ndtomp::Eval* e) { return pclndt_eval(c, pass, P, e); };
  ndtomp::Solver<decltype(ev)> s{ev};
  s.step_size = (double)c->cfg.ndt_step_size;
  s.eps = c->cfg.translation_eps;            // transformation_epsilon_
  s.outlier_ratio = (double)c->cfg.ndt_outlier_ratio;
  s.resolution = c->cfg.voxel_resolution;
  s.max_iterations = c->cfg.max_iterations;
  s.num_neighbors = c->cfg.num_neighbors;
  return s;
}

// buffers of a batched pclomp NDT registration, owned by the first context of the batch
struct NdtBatchWs {
  NdtObject* d_objs = nullptr;
  ndtomp::NdtMachine* d_ms = nullptr;
  NdtObject* h_objs = nullptr;            // pinned
  ndtomp::NdtMachine* h_ms = nullptr;     // pinned
  unsigned char* h_flags = nullptr;       // mapped pinned: [round][object] status bytes of k_pclndt_batch_step
  unsigned char* d_flags = nullptr;
  int cap = 0;
  size_t cap_flags = 0;
  hipStream_t gst[4] = {nullptr, nullptr, nullptr, nullptr};   // streams of the lock-step groups, created back to back
};

void free_ndt_batch_ws(void* p) {
  NdtBatchWs* w = static_cast<NdtBatchWs*>(p);
  if (!w) return;
  for (hipStream_t st : w->gst) if (st) (void)hipStreamDestroy(st);
  if (w->d_objs) hipFree(w->d_objs);
  if (w->d_ms) hipFree(w->d_ms);
  if (w->h_objs) hipHostFree(w->h_objs);
  if (w->h_ms) hipHostFree(w->h_ms);
  if (w->h_flags) hipHostFree(w->h_flags);
  delete w;
}

// pclomp::NormalDistributionsTransform::computeTransformation (ndt_omp_impl.hpp:69-156) for n objects: their solvers run on the
// device (ndtomp::NdtMachine, pclndt_host.h), one derivatives launch + one step launch per round for all of them
int pclndt_align_batch(pcm_ctx* const* ctxs, int n, const float* guesses, pcm_result* res) {
  pcm_ctx* c0 = ctxs[0];
  for (int i = 0; i < n; i++) {
    int rc = prepare(ctxs[i]);
    if (rc != PCM_OK) { if (i) c0->err = ctxs[i]->err; return rc; }
    if (ctxs[i]->stream != c0->stream) HIPCK(c0, hipStreamSynchronize(ctxs[i]->stream));   // its map / leaves were built on its own stream
  }
  if (!c0->ndt_ws) c0->ndt_ws = new (std::nothrow) NdtBatchWs();
  if (!c0->ndt_ws) { c0->err = "out of host memory"; return PCM_ERR_HIP; }
  NdtBatchWs& w = *static_cast<NdtBatchWs*>(c0->ndt_ws);
  if (n > w.cap) {
    if (w.d_objs) hipFree(w.d_objs);
    if (w.d_ms) hipFree(w.d_ms);
    if (w.h_objs) hipHostFree(w.h_objs);
    if (w.h_ms) hipHostFree(w.h_ms);
    w.d_objs = nullptr; w.d_ms = nullptr; w.h_objs = nullptr; w.h_ms = nullptr; w.cap = 0;
    const int cap = std::max(n, 16);
    HIPCK(c0, hipMalloc(&w.d_objs, sizeof(NdtObject) * cap));
    HIPCK(c0, hipMalloc(&w.d_ms, sizeof(ndtomp::NdtMachine) * cap));
    HIPCK(c0, hipHostMalloc(reinterpret_cast<void**>(&w.h_objs), sizeof(NdtObject) * cap, hipHostMallocDefault));
    HIPCK(c0, hipHostMalloc(reinterpret_cast<void**>(&w.h_ms), sizeof(ndtomp::NdtMachine) * cap, hipHostMallocDefault));
    w.cap = cap;
  }
  // an object asks for at most 12 evaluations per Newton iteration (1 + 10 trials + the Hessian pass) and runs max_iterations + 2 of them
  int max_rounds = 2;
  int max_blocks = 1;
  for (int i = 0; i < n; i++) {
    pcm_ctx* c = ctxs[i];
    w.h_objs[i] = make_ndt_object(c->map, c->pleaf, c->pleaf_f, ndt_lists_view(c), c->src.d_pts, (uint32_t)c->src.n, c->ndt_partials);
    max_blocks = std::max(max_blocks, (int)w.h_objs[i].nblocks);
    ndtomp::ndt_machine_start(w.h_ms[i], guesses + 16 * (size_t)i, (double)c->cfg.ndt_step_size, c->cfg.translation_eps, (double)c->cfg.ndt_outlier_ratio,
                              c->cfg.voxel_resolution, c->cfg.max_iterations, c->cfg.num_neighbors);
    max_rounds = std::max(max_rounds, (c->cfg.max_iterations + 3) * 12 + 2);
  }
  const size_t flag_bytes = (size_t)max_rounds * (size_t)n;
  if (flag_bytes > w.cap_flags) {
    if (w.h_flags) hipHostFree(w.h_flags);
    w.h_flags = nullptr; w.d_flags = nullptr; w.cap_flags = 0;
    const size_t bytes = std::max<size_t>(flag_bytes, 65536);
    HIPCK(c0, hipHostMalloc(reinterpret_cast<void**>(&w.h_flags), bytes, hipHostMallocMapped));
    HIPCK(c0, hipHostGetDevicePointer(reinterpret_cast<void**>(&w.d_flags), w.h_flags, 0));
    w.cap_flags = bytes;
  }
  std::memset(w.h_flags, 0, flag_bytes);
  hipStream_t st = c0->stream;
  HIPCK(c0, hipMemcpyAsync(w.d_objs, w.h_objs, sizeof(NdtObject) * n, hipMemcpyHostToDevice, st));
  HIPCK(c0, hipMemcpyAsync(w.d_ms, w.h_ms, sizeof(ndtomp::NdtMachine) * n, hipMemcpyHostToDevice, st));
  HIPCK(c0, hipStreamSynchronize(st));   // the groups below run on their own streams
  // The objects advance in up to four groups, each in lock-step on the stream of its first object: registrations need 6 ... 37
  // Newton iterations on the same map, and a single lock-step batch runs every round at the pace of its largest kernel while most
  // objects have finished.  At most two rounds of a group are in flight (the host confirms a round's status bytes before it queues
  // the one after the next); a group whose objects have all finished sees that one round late and stops.
  struct Group { int lo, hi, max_blocks, launched, confirmed; bool done; hipStream_t st; };
  // Measured at config 4 (100k-point scans, tools/r03_scaling.sh): 8 objects -- four groups 1 175 registrations/s, two 1 122; 16 objects --
  // one group 1 548, two 1 781, four 953; 32 objects -- one 2 340, two 2 753, three 2 217, four 1 990: one group's solver step and the
  // ragged end of its pass overlap the other's pass; more groups only add launches and host-side waiting.
  size_t total_points = 0;
  for (int i = 0; i < n; i++) total_points += ctxs[i]->src.n;
  // (the 27-cell searches -- KDTREE, DIRECT26 -- lose with two groups while they look their cells up one by one: 1 570 -> 1 213 at 32
  // scans, their pass keeps the device busy alone; on the grid's neighbour-leaf lists they gain like the others: 2 760 -> 3 203)
  const bool wide = (c0->cfg.num_neighbors == 0 || c0->cfg.num_neighbors > 7) && ndt_lists_view(c0).pts == nullptr;
  int ngroups = total_points <= 1000000 ? std::min(n, 4) : (wide ? 1 : std::min(n, 2));
  if (const char* e = getenv("PCM_NDT_GROUPS")) ngroups = std::max(1, std::min(std::min(n, 4), atoi(e)));   // measurements only
  std::vector<Group> groups((size_t)ngroups);
  for (int g = 0; g < ngroups; g++) {
    Group& G = groups[(size_t)g];
    G.lo = (int)((long long)n * g / ngroups); G.hi = (int)((long long)n * (g + 1) / ngroups);
    G.launched = 0; G.confirmed = 0; G.done = false; G.max_blocks = 1;
    // Streams of the groups' own, created back to back: HIP deals streams onto a handful of hardware queues in creation order, and two
    // groups whose streams share a queue do not overlap.  (With the streams of the groups' first objects the first batch of a process
    // ran at 3 225 registrations/s and a second batch of objects created later in the same process at 2 016, or the other way round.)
    if (ngroups == 1) G.st = ctxs[G.lo]->stream;
    else {
      if (!w.gst[g]) HIPCK(c0, hipStreamCreateWithFlags(&w.gst[g], hipStreamNonBlocking));
      G.st = w.gst[g];
    }
    for (int i = G.lo; i < G.hi; i++) G.max_blocks = std::max(G.max_blocks, (int)w.h_objs[i].nblocks);
  }
  const auto t_start = std::chrono::steady_clock::now();
  auto next_query = t_start + std::chrono::milliseconds(5);
  int live = ngroups;
  unsigned idle_spins = 0;
  while (live > 0) {
    bool progress = false;
    for (Group& G : groups) {
      if (G.done) continue;
      if (G.confirmed < G.launched) {   // status bytes of the oldest unconfirmed round of this group: all landed?
        volatile unsigned char* row = w.h_flags + (size_t)G.confirmed * n;
        bool ready = true, any_active = false;
        for (int i = G.lo; i < G.hi; i++) { const unsigned char f = row[i]; ready &= f != 0; any_active |= f == 1; }
        if (ready) {
          G.confirmed++;
          progress = true;
          if (!any_active || G.confirmed >= max_rounds) { G.done = true; live--; continue; }
        }
      }
      if (G.launched - G.confirmed < 2 && G.launched < max_rounds) {
        launch_pclndt_batch_round(G.st, w.d_objs + G.lo, w.d_ms + G.lo, G.hi - G.lo, G.max_blocks, w.d_flags + (size_t)G.launched * n + G.lo);
        G.launched++;
        progress = true;
      }
    }
    if (progress) { idle_spins = 0; continue; }
    if ((++idle_spins & 0xfff) != 0) continue;
    const auto now = std::chrono::steady_clock::now();
    auto drain = [&]() { for (Group& G : groups) (void)hipStreamSynchronize(G.st); };
    if (now - t_start > std::chrono::seconds(20)) { drain(); c0->err = "timeout waiting for the GPU round status"; return PCM_ERR_HIP; }
    if (now >= next_query) {   // a dead stream never writes its status bytes (see wait_status_byte for why this is asked rarely)
      next_query = now + std::chrono::milliseconds(5);
      for (Group& G : groups)
        if (!G.done && G.confirmed < G.launched && hipStreamQuery(G.st) == hipSuccess) {
          volatile unsigned char* row = w.h_flags + (size_t)G.confirmed * n;
          bool ready = true;
          for (int i = G.lo; i < G.hi; i++) ready &= row[i] != 0;
          if (!ready) { drain(); c0->err = "stream drained without a round status (kernel fault?)"; return PCM_ERR_HIP; }
        }
    }
  }
  HIPCK(c0, hipGetLastError());
  for (Group& G : groups) HIPCK(c0, hipStreamSynchronize(G.st));
  HIPCK(c0, hipMemcpyAsync(w.h_ms, w.d_ms, sizeof(ndtomp::NdtMachine) * n, hipMemcpyDeviceToHost, st));
  HIPCK(c0, hipStreamSynchronize(st));
  int worst = PCM_OK;
  for (int i = 0; i < n; i++) {
    const ndtomp::NdtMachine& m = w.h_ms[i];
    pcm_result* out = &res[i];
    std::memset(out, 0, sizeof(*out));
    for (int k = 0; k < 16; k++) { out->T[k] = m.P.T[k]; out->T64[k] = (double)m.P.T[k]; }
    std::memcpy(out->H, m.cur.H, sizeof(out->H));   // hessian_eigen_
    out->cost = m.cur.score;                        // trans_probability_ * N
    out->iterations = m.nr;
    out->converged = m.converged;
    out->num_linearize = m.n_deriv;
    out->num_compute_error = m.n_hess;
    out->status = m.request < 0 ? PCM_OK : PCM_ERR_HIP;   // a machine still asking after max_rounds cannot happen (bounded loops)
    if (out->status != PCM_OK) { worst = out->status; c0->err = "pclomp NDT solver did not finish within its round budget"; }
    ctxs[i]->stats.linearize_launches += (uint64_t)(m.n_deriv + m.n_hess);
    ctxs[i]->stats.point_passes += (uint64_t)(m.n_deriv + m.n_hess) * ctxs[i]->src.n;
  }
  return worst;
}

int pclndt_align(pcm_ctx* c, const float guess[16], pcm_result* out) {
  pcm_ctx* one[1] = {c};
  return pclndt_align_batch(one, 1, guess, out);
}

}  // namespace

extern "C" {

int pcm_abi_version(void) { return PCM_ABI_VERSION; }

void pcm_default_config(pcm_config* cfg) {
  if (!cfg) return;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->model = PCM_MODEL_P2PLANE;
  cfg->optimizer = PCM_OPT_LEVENBERG_MARQUARDT;
  cfg->max_iterations = 64;
  cfg->lm_max_iterations = 10;
  cfg->rotation_eps = 2e-3;
  cfg->translation_eps = 5e-4;
  cfg->lm_init_lambda_factor = 1e-9;
  cfg->voxel_resolution = 0.5f;
  cfg->num_neighbors = 27;
  cfg->knn = 5;
  cfg->min_knn = 3;
  cfg->max_range = 5.0f;
  cfg->plane_threshold = 0.1f;
  cfg->max_corr_dist = FLT_MAX;
  cfg->k_correspondences = 20;
  cfg->regularization = PCM_REG_PLANE;
  cfg->sort_source = 1;
  cfg->map_capacity = 1000000;   // IVox Options::capacity_  ivox3d.h:57
  cfg->ndt_step_size = 0.1f;     // ndt_omp_impl.hpp:48
  cfg->ndt_outlier_ratio = 0.55f;
  cfg->covariance_method = PCM_COV_KNN;
  cfg->rbf_kernel_width = 0.25f;   // FastVGICPCudaCore  fast_vgicp_cuda.cu:25-26
  cfg->rbf_max_dist = 3.0f;
}

pcm_ctx* pcm_create(int device, const pcm_config* cfg) {
  pcm_ctx* c = new (std::nothrow) pcm_ctx();
  if (!c) return nullptr;
  c->device = device;
  if (cfg) c->cfg = *cfg; else pcm_default_config(&c->cfg);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    // keep the object so the caller can read the reason, but mark it unusable
    c->err = "no such HIP device (the MI355X path has no CPU fallback)";
    c->device = -1;
    return c;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    c->err = "hipStreamCreate failed";
    c->device = -1;
    return c;
  }
  {   // keep freed scratch in the device's stream-ordered pool instead of returning it to the driver at every sync
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
      uint64_t keep = ~0ull;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
  }
  c->own_stream = true;
  return c;
}

void pcm_destroy(pcm_ctx* c) {
  if (!c) return;
  if (c->device >= 0) {
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    c->src.release();
    c->tgt.release();
    if (c->src_order) hipFree(c->src_order);
    if (c->lio_aux) hipFree(c->lio_aux);
    free_ndt_batch_ws(c->ndt_ws);
    c->ndt_ws = nullptr;
    c->map.release();
    c->srcmap.release();
    c->covfine.release();
    c->nlists.release();
    if (c->corr) hipFree(c->corr);
    if (c->src_cov) hipFree(c->src_cov);
    if (c->tgt_cov) hipFree(c->tgt_cov);
    if (c->vvox) hipFree(c->vvox);
    if (c->cvox) hipFree(c->cvox);
    if (c->maha) hipFree(c->maha);
    if (c->pleaf) hipFree(c->pleaf);
    if (c->pleaf_f) hipFree(c->pleaf_f);
    if (c->pre_arena) hipFree(c->pre_arena);
    if (c->bfgs) hipFree(c->bfgs);
    if (c->bfgs_idx) hipFree(c->bfgs_idx);
    if (c->bfgs_host) hipHostFree(c->bfgs_host);
    if (c->ndt_partials) hipFree(c->ndt_partials);
    if (c->ndt_out) hipFree(c->ndt_out);
    if (c->ndt_out_host) hipHostFree(c->ndt_out_host);
    if (c->planes) hipFree(c->planes);
    if (c->counter) hipFree(c->counter);
    if (c->nn) hipFree(c->nn);
    free_ws(c);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  }
  delete c;
}

const char* pcm_last_error(const pcm_ctx* c) { return c ? c->err.c_str() : "null context"; }

#define CHECK_CTX(c)                                                   \
  do {                                                                 \
    if (!(c)) return PCM_ERR_INVALID_ARGUMENT;                         \
    if ((c)->device < 0) return PCM_ERR_HIP;                           \
  } while (0)

int pcm_get_config(const pcm_ctx* c, pcm_config* out) {
  if (!c || !out) return PCM_ERR_INVALID_ARGUMENT;
  *out = c->cfg;
  return PCM_OK;
}

int pcm_set_config(pcm_ctx* c, const pcm_config* cfg) {
  CHECK_CTX(c);
  if (!cfg) return PCM_ERR_INVALID_ARGUMENT;
  int rc = validate_config(c, *cfg);
  if (rc != PCM_OK) return rc;
  c->cfg = *cfg;
  return PCM_OK;
}

int pcm_set_stream(pcm_ctx* c, void* hip_stream) {
  CHECK_CTX(c);
  if (c->own_stream && c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
  c->stream = static_cast<hipStream_t>(hip_stream);
  c->own_stream = false;
  return PCM_OK;
}

int pcm_set_target(pcm_ctx* c, const void* points, size_t n, size_t stride_bytes, int memory, uint64_t tag) {
  CHECK_CTX(c);
  if (tag != 0 && tag == c->tgt.tag && c->tgt.n == n) return PCM_OK;  // `if (target_ == cloud) return;`  fast_gicp_impl.hpp:83-85
  int rc = set_cloud(c, &c->tgt, points, n, stride_bytes, memory, tag, false);
  c->user_cov[1].clear();   // target_covs_.clear()  fast_gicp_impl.hpp:89
  c->map.valid = false;
  c->map.index_n = 0;   // another log: the sorted index of the old one is of no use
  c->tgt_dynamic = false;
  c->next_seq = (uint32_t)n;
  c->lio_planes_valid = false;
  return rc;
}

namespace {
// residuals_.resize(cur_pts, 0); point_selected_surf_.resize(cur_pts, true)  (laser_mapping.cc:337-338): the first
// min(old, new) entries survive a new scan, appended ones take the default.  plane_coef_ (:339) needs no such care: the
// IEKF's first ObsModel call of a frame always matches (esekfom.hpp:1529) and rewrites every plane it may read later.
static int lio_members_resize(pcm_ctx* c, size_t n) {
  if (n > c->lio_aux_cap) {
    const size_t cap = n + n / 2 + 1024;
    float2* nb = nullptr;
    HIPCK(c, hipMalloc(&nb, sizeof(float2) * cap));
    if (c->lio_aux_n) HIPCK(c, hipMemcpyAsync(nb, c->lio_aux, sizeof(float2) * c->lio_aux_n, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->lio_aux) hipFree(c->lio_aux);
    c->lio_aux = nb;
    c->lio_aux_cap = cap;
  }
  if (n > c->lio_aux_n) launch_lio_members_init(c->stream, c->lio_aux, (uint32_t)c->lio_aux_n, (uint32_t)n);
  c->lio_aux_n = n;
  return PCM_OK;
}
}  // namespace

int pcm_set_source(pcm_ctx* c, const void* points, size_t n, size_t stride_bytes, int memory, uint64_t tag) {
  CHECK_CTX(c);
  if (tag != 0 && tag == c->src.tag && c->src.n == n) return PCM_OK;  // fast_gicp_impl.hpp:72-74
  int rc = set_cloud(c, &c->src, points, n, stride_bytes, memory, tag, true);
  c->user_cov[0].clear();   // source_covs_.clear()  fast_gicp_impl.hpp:78
  c->src_sorted = false;
  c->lio_planes_valid = false;
  c->srcmap.valid = false;
  if (rc == PCM_OK && (c->cfg.flags & PCM_FLAG_LIO_REFERENCE_SEMANTICS)) rc = lio_members_resize(c, n);   // one resize per frame
  return rc;
}

int pcm_swap_source_and_target(pcm_ctx* c) {
  CHECK_CTX(c);
  HIPCK(c, hipStreamSynchronize(c->stream));
  std::swap(c->src, c->tgt);
  std::swap(c->user_cov[0], c->user_cov[1]);   // source_covs_.swap(target_covs_)  fast_gicp_impl.hpp:55
  c->map.valid = false;
  c->srcmap.valid = false;
  c->map.index_n = 0; c->srcmap.index_n = 0;
  c->src_sorted = false;
  return PCM_OK;
}

int pcm_clear_source(pcm_ctx* c) {
  CHECK_CTX(c);
  c->src.n = 0; c->src.tag = 0; c->src_sorted = false; c->srcmap.valid = false;
  return PCM_OK;
}

int pcm_clear_target(pcm_ctx* c) {
  CHECK_CTX(c);
  c->tgt.n = 0; c->tgt.tag = 0; c->map.valid = false; c->map.index_n = 0;
  return PCM_OK;
}

// pclomp::NormalDistributionsTransform::calculateScore (ndt_omp_impl.hpp:835-880) of the source transformed by T
int pcm_ndt_score(pcm_ctx* c, const float T[16], double* score) {
  CHECK_CTX(c);
  if (!T || !score) return PCM_ERR_INVALID_ARGUMENT;
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  if (c->cfg.model != PCM_MODEL_NDT_OMP) { c->err = "pcm_ndt_score needs the NDT_OMP model"; return PCM_ERR_UNSUPPORTED; }
  rc = prepare(c);
  if (rc != PCM_OK) return rc;
  auto solver = make_ndt_solver(c);
  solver.gauss_params();
  std::memcpy(solver.P.T, T, sizeof(float) * 16);
  ndtomp::Eval e{};
  rc = pclndt_eval(c, 3, solver.P, &e, solver.gauss_d3);
  if (rc != PCM_OK) return rc;
  *score = e.score / (double)c->src.n;
  return PCM_OK;
}

// pcl::Registration::getFitnessScore(max_range): mean squared distance of the source points, transformed by T, to their exact
// nearest target points, over the points whose squared distance is <= max_range (call sites: localization.cpp:325-326,
// mapOptmization.cpp:693,719, fast_gicp/src/align.cpp:63)
int pcm_fitness_score(pcm_ctx* c, const float T[16], double max_range, double* score) {
  CHECK_CTX(c);
  if (!T || !score) return PCM_ERR_INVALID_ARGUMENT;
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  rc = prepare(c);
  if (rc != PCM_OK) return rc;
  if (!c->map.valid || !c->map.pts || c->src.n == 0) { c->err = "pcm_fitness_score: set a source and a target first"; return PCM_ERR_INVALID_ARGUMENT; }
  PairDesc d;
  fill_desc(c, &d, nullptr);
  const uint32_t n = (uint32_t)c->src.n;
  const uint32_t nblocks = (n + 255u) / 256u;
  Workspace* w = nullptr;
  rc = ensure_ws(c, &w, 1, (size_t)std::max<uint32_t>(2u * nblocks, kPartialStride), 2);
  if (rc != PCM_OK) return rc;
  launch_fitness(c->stream, d.tgt, coord_mode_for(c->cfg.model), c->src.d_pts, n, T, max_range, w->d_partials);
  HIPCK(c, hipGetLastError());
  std::vector<double> rows(2 * (size_t)nblocks);
  HIPCK(c, hipMemcpyAsync(rows.data(), w->d_partials, sizeof(double) * rows.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  double sum = 0.0, cnt = 0.0;
  for (uint32_t b = 0; b < nblocks; b++) { sum += rows[2 * b]; cnt += rows[2 * b + 1]; }
  *score = cnt > 0.0 ? sum / cnt : DBL_MAX;
  return PCM_OK;
}

// ImuProcess::UndistortPcl backward propagation  (jueying_lio/include/imu_processing.hpp:245-285)
int pcm_undistort(pcm_ctx* c, void* points, size_t n, size_t stride, size_t time_off, int memory, const pcm_imu_pose* poses, int npose, const pcm_lio_state* st) {
  CHECK_CTX(c);
  if ((!points && n) || !poses || !st || npose < 0) return PCM_ERR_INVALID_ARGUMENT;
  if (stride < 16 || (stride % 4) != 0 || time_off + 4 > stride || (time_off % 4) != 0) { c->err = "bad record layout"; return PCM_ERR_INVALID_ARGUMENT; }
  if (n == 0 || npose < 2) return PCM_OK;
  HIPCK(c, hipSetDevice(c->device));
  LioStateD s;
  for (int a = 0; a < 4; a++) { s.rot[a] = st->rot[a]; s.off_R[a] = st->off_R[a]; }
  for (int a = 0; a < 3; a++) { s.pos[a] = st->pos[a]; s.off_T[a] = st->off_T[a]; }
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t need = up(sizeof(pcm_imu_pose) * (size_t)npose) + (memory == PCM_MEM_HOST ? up(n * stride) : 0);
  if (c->pre_arena_cap < need) {   // grow-only arena shared with pcm_voxel_downsample
    if (c->pre_arena) hipFree(c->pre_arena);
    c->pre_arena = nullptr; c->pre_arena_cap = 0;
    HIPCK(c, hipMalloc(&c->pre_arena, need + need / 4));
    c->pre_arena_cap = need + need / 4;
  }
  pcm_imu_pose* d_poses = reinterpret_cast<pcm_imu_pose*>(c->pre_arena);
  HIPCK(c, hipMemcpyAsync(d_poses, poses, sizeof(pcm_imu_pose) * (size_t)npose, hipMemcpyHostToDevice, c->stream));
  void* d_pts = points;
  if (memory == PCM_MEM_HOST) {
    d_pts = c->pre_arena + up(sizeof(pcm_imu_pose) * (size_t)npose);
    HIPCK(c, hipMemcpyAsync(d_pts, points, n * stride, hipMemcpyHostToDevice, c->stream));
  }
  const int rc = undistort_device(c->stream, d_pts, n, stride, time_off, d_poses, npose, s, &c->err);
  if (rc != PCM_OK) return rc;
  if (memory == PCM_MEM_HOST) HIPCK(c, hipMemcpyAsync(points, d_pts, n * stride, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  return PCM_OK;
}

// ---- pclomp GICP-BFGS functor (ndt_omp/include/pclomp/gicp_omp_impl.hpp) ------------------------------------------
namespace {

// applyState (:519-529): t <- R t with R from AngleAxisf(z) * AngleAxisf(y) * AngleAxisf(x) (a float quaternion product), then the translation
void bfgs_apply_state(const float* base, const double* x, float* T) {
  auto axis_quat = [](float ang, int k, float* q) { const float h = 0.5f * ang; q[0] = q[1] = q[2] = 0.f; q[k] = sinf(h); q[3] = cosf(h); };
  auto mul = [](const float* a, const float* b, float* r) {
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  };
  float qz[4], qy[4], qx[4], qa[4], q[4];
  axis_quat((float)x[5], 2, qz); axis_quat((float)x[4], 1, qy); axis_quat((float)x[3], 0, qx);
  mul(qz, qy, qa); mul(qa, qx, q);
  const float tx = 2.f * q[0], ty = 2.f * q[1], tz = 2.f * q[2];
  const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  const float R[9] = {1.f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.f - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.f - (txx + tyy)};
  for (int i = 0; i < 16; i++) T[i] = base[i];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[i * 4 + j] = (R[i * 3] * base[j] + R[i * 3 + 1] * base[4 + j]) + R[i * 3 + 2] * base[8 + j];
    T[i * 4 + 3] = base[i * 4 + 3] + (float)x[i];
  }
}

// computeRDerivative (:125-176): g[3..5] = <dR/dphi, R>, <dR/dtheta, R>, <dR/dpsi, R>  (R row-major)
void bfgs_r_derivative(const double* x, const double* R, double* g) {
  const double cf = cos(x[3]), sf = sin(x[3]), ct = cos(x[4]), st = sin(x[4]), cp = cos(x[5]), sp = sin(x[5]);
  const double d[3][9] = {{0., sf * sp + cf * cp * st, cf * sp - cp * sf * st, 0., -cp * sf + cf * sp * st, -cf * cp - sf * sp * st, 0., cf * ct, -ct * sf},
                          {-cp * st, cp * ct * sf, cf * cp * ct, -sp * st, ct * sf * sp, cf * ct * sp, -ct, -sf * st, -cf * st},
                          {-ct * sp, -cf * cp - sf * sp * st, cp * sf - cf * sp * st, cp * ct, -cf * sp + cp * sf * st, sf * sp + cf * cp * st, 0., 0., 0.}};
  for (int k = 0; k < 3; k++) {
    double r = 0.;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r += d[k][j * 3 + i] * R[i * 3 + j];   // matricesInnerProd (gicp_omp.h:325-334)
    g[3 + k] = r;
  }
}

}  // namespace

int pcm_gicp_bfgs_set_correspondences(pcm_ctx* c, const void* src, size_t n_src, const void* tgt, size_t n_tgt, size_t stride, const int32_t* idx_src, const int32_t* idx_tgt,
                                      size_t m, const float* maha, int memory) {
  CHECK_CTX(c);
  if (m && (!src || !tgt || !idx_src || !idx_tgt || !maha)) return PCM_ERR_INVALID_ARGUMENT;
  if (stride < 12 || (stride % 4) != 0) { c->err = "bad record layout"; return PCM_ERR_INVALID_ARGUMENT; }
  if (m > 0xffffffffull) { c->err = "too many correspondences"; return PCM_ERR_INVALID_ARGUMENT; }
  if (memory == PCM_MEM_HOST)
    for (size_t i = 0; i < m; i++)
      if (idx_src[i] < 0 || (size_t)idx_src[i] >= n_src || idx_tgt[i] < 0 || (size_t)idx_tgt[i] >= n_tgt) { c->err = "correspondence index out of range"; return PCM_ERR_INVALID_ARGUMENT; }
  HIPCK(c, hipSetDevice(c->device));
  c->bfgs_m = 0;
  const size_t need = gicp_bfgs_scratch_bytes(m);
  if (c->bfgs_cap < need) {
    if (c->bfgs) hipFree(c->bfgs);
    c->bfgs = nullptr; c->bfgs_cap = 0;
    HIPCK(c, hipMalloc(&c->bfgs, need + need / 4));
    c->bfgs_cap = need + need / 4;
  }
  if (m == 0) return PCM_OK;
  const void *d_src = src, *d_tgt = tgt;
  const int32_t *d_is = idx_src, *d_it = idx_tgt;
  const float* d_maha = maha;
  char* tmp = nullptr;
  if (memory == PCM_MEM_HOST) {   // staged once per correspondence set; the evaluations then read the packed records only
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_src = up(n_src * stride), b_tgt = up(n_tgt * stride), b_idx = up(m * 4), b_maha = up(n_src * 64);
    HIPCK(c, hipMallocAsync(reinterpret_cast<void**>(&tmp), b_src + b_tgt + 2 * b_idx + b_maha, c->stream));
    char* q = tmp;
    HIPCK(c, hipMemcpyAsync(q, src, n_src * stride, hipMemcpyHostToDevice, c->stream)); d_src = q; q += b_src;
    HIPCK(c, hipMemcpyAsync(q, tgt, n_tgt * stride, hipMemcpyHostToDevice, c->stream)); d_tgt = q; q += b_tgt;
    HIPCK(c, hipMemcpyAsync(q, idx_src, m * 4, hipMemcpyHostToDevice, c->stream)); d_is = reinterpret_cast<const int32_t*>(q); q += b_idx;
    HIPCK(c, hipMemcpyAsync(q, idx_tgt, m * 4, hipMemcpyHostToDevice, c->stream)); d_it = reinterpret_cast<const int32_t*>(q); q += b_idx;
    HIPCK(c, hipMemcpyAsync(q, maha, n_src * 64, hipMemcpyHostToDevice, c->stream)); d_maha = reinterpret_cast<const float*>(q);
  }
  const int rc = gicp_bfgs_pack_device(c->stream, d_src, d_tgt, stride, d_is, d_it, d_maha, m, c->bfgs, &c->err);
  if (tmp) HIPCK(c, hipFreeAsync(tmp, c->stream));
  if (rc != PCM_OK) return rc;
  HIPCK(c, hipStreamSynchronize(c->stream));   // the caller's buffers are free again
  c->bfgs_m = m;
  return PCM_OK;
}

int pcm_gicp_bfgs_fdf(pcm_ctx* c, const float* base_T, const double* x, int mode, double* f, double* g) {
  CHECK_CTX(c);
  if (!base_T || !x || mode < 0 || mode > 2 || (mode != 1 && !f) || (mode != 0 && !g)) return PCM_ERR_INVALID_ARGUMENT;
  if (c->bfgs_m == 0) { c->err = "no correspondences (pcm_gicp_bfgs_set_correspondences)"; return PCM_ERR_NO_INPUT; }
  HIPCK(c, hipSetDevice(c->device));
  float T[16];
  bfgs_apply_state(base_T, x, T);
  const size_t m = c->bfgs_m;
  double* d_partials = reinterpret_cast<double*>(c->bfgs + ((64 * m + 255) & ~(size_t)255));
  if (!c->bfgs_host) HIPCK(c, hipHostMalloc(reinterpret_cast<void**>(&c->bfgs_host), 16 * sizeof(double), hipHostMallocDefault));
  const int rc = gicp_bfgs_fdf_device(c->stream, c->bfgs, m, T, base_T, d_partials, c->bfgs_host, &c->err);   // the finish kernel stores to host memory
  if (rc != PCM_OK) return rc;
  HIPCK(c, hipStreamSynchronize(c->stream));
  const double* s = c->bfgs_host;
  const double dm = (double)m;
  if (f && mode != 1) *f = (mode == 0 ? s[0] : s[1]) / dm;
  if (g && mode != 0) {
    double R[9];
    for (int a = 0; a < 3; a++) g[a] = s[2 + a] * (2.0 / dm);
    for (int a = 0; a < 9; a++) R[a] = s[5 + a] * (2.0 / dm);
    bfgs_r_derivative(x, R, g);
  }
  return PCM_OK;
}

int pcm_gicp_bfgs_update_correspondences(pcm_ctx* c, const float* transformation, const float* guess, size_t* m_out) {
  CHECK_CTX(c);
  if (!transformation || !guess || !m_out) return PCM_ERR_INVALID_ARGUMENT;
  *m_out = 0;
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  if (c->cfg.model != PCM_MODEL_GICP) { c->err = "pcm_gicp_bfgs_update_correspondences needs the GICP model"; return PCM_ERR_UNSUPPORTED; }
  rc = prepare(c);   // target map + its covariances, brick-major copy of the source + its covariances
  if (rc != PCM_OK) return rc;
  const size_t n = c->srcmap.num_points;
  c->bfgs_m = 0;
  const size_t need = gicp_bfgs_scratch_bytes(n);
  if (c->bfgs_cap < need) {
    if (c->bfgs) hipFree(c->bfgs);
    c->bfgs = nullptr; c->bfgs_cap = 0;
    HIPCK(c, hipMalloc(&c->bfgs, need + need / 4));
    c->bfgs_cap = need + need / 4;
  }
  if (c->bfgs_idx_cap < n) {
    if (c->bfgs_idx) hipFree(c->bfgs_idx);
    c->bfgs_idx = nullptr; c->bfgs_idx_cap = 0;
    HIPCK(c, hipMalloc(&c->bfgs_idx, sizeof(int32_t) * 2 * (n + n / 4 + 64)));
    c->bfgs_idx_cap = n + n / 4 + 64;
  }
  uint32_t m = 0;
  rc = gicp_bfgs_correspond_device(c->stream, c->map, coord_mode_for(c->cfg.model), c->srcmap, c->src_cov, c->tgt_cov, guess, transformation, (double)c->cfg.max_corr_dist,
                                   reinterpret_cast<float4*>(c->bfgs), c->bfgs_idx, c->bfgs_idx + c->bfgs_idx_cap, &m, &c->err);
  if (rc != PCM_OK) return rc;
  HIPCK(c, hipStreamSynchronize(c->stream));
  c->bfgs_m = m;
  *m_out = m;
  return PCM_OK;
}

int pcm_gicp_bfgs_get_correspondences(pcm_ctx* c, int32_t* idx_src, int32_t* idx_tgt, float* maha9, size_t capacity) {
  CHECK_CTX(c);
  const size_t m = c->bfgs_m;
  if (!c->bfgs_idx || capacity < m) { c->err = "pcm_gicp_bfgs_get_correspondences: no device-side correspondence set, or the buffers are too small"; return PCM_ERR_INVALID_ARGUMENT; }
  if (m == 0) return PCM_OK;
  if (idx_src) HIPCK(c, hipMemcpy(idx_src, c->bfgs_idx, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
  if (idx_tgt) HIPCK(c, hipMemcpy(idx_tgt, c->bfgs_idx + c->bfgs_idx_cap, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
  if (maha9) {
    std::vector<float4> r(3 * m);   // planes 1..3 of the records hold M
    HIPCK(c, hipMemcpy(r.data(), reinterpret_cast<const float4*>(c->bfgs) + m, sizeof(float4) * 3 * m, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; i++) {
      const float4 a = r[i], b = r[m + i], d = r[2 * m + i];
      float* o = maha9 + 9 * i;
      o[0] = a.z; o[1] = a.w; o[2] = b.x; o[3] = b.y; o[4] = b.z; o[5] = b.w; o[6] = d.x; o[7] = d.y; o[8] = d.z;
    }
  }
  return PCM_OK;
}

// pcl::VoxelGrid::filter of the scan (jueying_lio/src/laser_mapping.cc:323-328)
int pcm_voxel_downsample(pcm_ctx* c, const void* points, size_t n, size_t stride, int memory, float leaf, void* out, size_t capacity_points, size_t* n_out) {
  CHECK_CTX(c);
  if ((!points && n) || !out || !n_out) return PCM_ERR_INVALID_ARGUMENT;
  if (capacity_points < n) { c->err = "the output buffer must hold as many records as the input"; return PCM_ERR_INVALID_ARGUMENT; }
  *n_out = 0;
  if (n == 0) return PCM_OK;
  HIPCK(c, hipSetDevice(c->device));
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t io = memory == PCM_MEM_HOST ? 2 * up(n * stride) : 0;
  const size_t need = voxel_downsample_scratch_bytes(n) + io;
  if (c->pre_arena_cap < need) {   // grow-only arena: no allocation per frame in the steady state
    if (c->pre_arena) hipFree(c->pre_arena);
    c->pre_arena = nullptr; c->pre_arena_cap = 0;
    HIPCK(c, hipMalloc(&c->pre_arena, need + need / 4));
    c->pre_arena_cap = need + need / 4;
  }
  const void* src = points;
  void* dst = out;
  char* scratch = c->pre_arena;
  if (memory == PCM_MEM_HOST) {
    char* d_in = c->pre_arena;
    char* d_out = c->pre_arena + up(n * stride);
    scratch = c->pre_arena + io;
    HIPCK(c, hipMemcpyAsync(d_in, points, n * stride, hipMemcpyHostToDevice, c->stream));
    src = d_in; dst = d_out;
  }
  int rc = voxel_downsample_device(c->stream, src, n, stride, leaf, static_cast<float*>(dst), n_out, scratch, &c->err);
  if (rc != PCM_OK) return rc;
  if (memory == PCM_MEM_HOST && *n_out) {
    HIPCK(c, hipMemcpyAsync(out, dst, *n_out * stride, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
  }
  return PCM_OK;
}

// setSourceCovariances / setTargetCovariances  fast_gicp_impl.hpp:93-100: `covs` = n matrices of `elems` doubles each (9: 3x3, 16: the
// reference's Matrix4d -- its top-left 3x3 block; symmetric, so row- and column-major read the same), input order
int pcm_set_covariances(pcm_ctx* c, int target, const double* covs, size_t n, int elems) {
  CHECK_CTX(c);
  if ((!covs && n) || (elems != 9 && elems != 16)) return PCM_ERR_INVALID_ARGUMENT;
  if (!is_gicp(c->cfg.model) || c->cfg.model == PCM_MODEL_VGICP_CUDA) { c->err = "covariances can be set for the GICP / VGICP models"; return PCM_ERR_UNSUPPORTED; }
  std::vector<double>& u = c->user_cov[target ? 1 : 0];
  u.resize(n * 6);
  const int ld = elems == 9 ? 3 : 4;
  for (size_t i = 0; i < n; i++) {
    const double* m = covs + i * (size_t)elems;
    double* o = &u[i * 6];
    o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[ld + 1]; o[4] = m[ld + 2]; o[5] = m[2 * ld + 2];
  }
  if (target) c->tgt_cov_valid = false; else c->src_cov_valid = false;
  return PCM_OK;
}

// PointCloudPreprocess::AviaHandler  (jueying_lio/src/pointcloud_preprocess.cc:44-88)
int pcm_livox_filter(pcm_ctx* c, const void* custom_points, size_t n, int memory, int num_scans, int point_filter_num, double blind, void* out, size_t capacity_points, size_t* n_out) {
  CHECK_CTX(c);
  if ((!custom_points && n) || !out || !n_out) return PCM_ERR_INVALID_ARGUMENT;
  if (capacity_points < n) { c->err = "the output buffer must hold as many records as the input"; return PCM_ERR_INVALID_ARGUMENT; }
  if (n > 0xffffffffull) { c->err = "too many points"; return PCM_ERR_INVALID_ARGUMENT; }
  *n_out = 0;
  if (n == 0) return PCM_OK;
  HIPCK(c, hipSetDevice(c->device));
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t io = memory == PCM_MEM_HOST ? up(n * 20) + up(n * 48) : 0;
  const size_t need = livox_filter_scratch_bytes(n) + io;
  if (c->pre_arena_cap < need) {   // grow-only arena shared with the other pre-processing operators
    if (c->pre_arena) hipFree(c->pre_arena);
    c->pre_arena = nullptr; c->pre_arena_cap = 0;
    HIPCK(c, hipMalloc(&c->pre_arena, need + need / 4));
    c->pre_arena_cap = need + need / 4;
  }
  const void* src = custom_points;
  void* dst = out;
  char* scratch = c->pre_arena;
  if (memory == PCM_MEM_HOST) {
    char* d_in = c->pre_arena;
    char* d_out = c->pre_arena + up(n * 20);
    scratch = c->pre_arena + io;
    HIPCK(c, hipMemcpyAsync(d_in, custom_points, n * 20, hipMemcpyHostToDevice, c->stream));
    src = d_in; dst = d_out;
  }
  const int rc = livox_filter_device(c->stream, src, n, num_scans, point_filter_num, blind, dst, n_out, scratch, &c->err);
  if (rc != PCM_OK) return rc;
  if (memory == PCM_MEM_HOST && *n_out) HIPCK(c, hipMemcpyAsync(out, dst, *n_out * 48, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  return PCM_OK;
}

// getSourceCovariances / getTargetCovariances  fast_gicp.hpp:64-70  (input order, row-major 3x3 blocks)
int pcm_get_covariances(pcm_ctx* c, int target, double* out, size_t capacity_points, size_t* n) {
  CHECK_CTX(c);
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  if (!is_gicp(c->cfg.model)) { c->err = "covariances exist for the GICP / VGICP models only"; return PCM_ERR_UNSUPPORTED; }
  rc = prepare(c);
  if (rc != PCM_OK) return rc;
  const TargetMap& m = target ? c->map : c->srcmap;
  const double* d_cov = target ? c->tgt_cov : c->src_cov;
  if (n) *n = m.num_points;
  if (!out) return PCM_OK;
  if (capacity_points < m.num_points) { c->err = "output buffer too small"; return PCM_ERR_INVALID_ARGUMENT; }
  std::vector<double> h6((size_t)m.num_points * 6);
  std::vector<uint32_t> ord(m.num_points);
  HIPCK(c, hipMemcpyAsync(h6.data(), d_cov, sizeof(double) * h6.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipMemcpyAsync(ord.data(), m.order, sizeof(uint32_t) * ord.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < ord.size(); i++) {
    const double* s = &h6[i * 6];
    double* o = out + (size_t)ord[i] * 9;
    if (c->cfg.model == PCM_MODEL_VGICP_CUDA) {   // the slot holds the 9 floats of the CUDA-core covariance
      const float* f = reinterpret_cast<const float*>(s);
      for (int a = 0; a < 9; a++) o[a] = (double)f[a];
      continue;
    }
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[1]; o[4] = s[3]; o[5] = s[4]; o[6] = s[2]; o[7] = s[4]; o[8] = s[5];
  }
  return PCM_OK;
}

int pcm_align(pcm_ctx* c, const float guess[16], pcm_result* out) {
  CHECK_CTX(c);
  if (!guess || !out) return PCM_ERR_INVALID_ARGUMENT;
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  if (c->cfg.model == PCM_MODEL_NDT_OMP) return pclndt_align(c, guess, out);
  pcm_ctx* arr[1] = {c};
  return align_batch_impl(arr, 1, guess, out, nullptr);
}

// pclomp NDT: score, gradient, Hessian at the pose vector p = (x, y, z, roll, pitch, yaw) exactly as the line search
// evaluates them (computeDerivatives, ndt_omp_impl.hpp:168-267); pass 2 = computeHessian (:498-559) with the angle
// tables of the previous call
int pcm_ndt_derivatives(pcm_ctx* c, const double p[6], int pass, double* score, double g[6], double H[36]) {
  CHECK_CTX(c);
  if (!p || pass < 0 || pass > 2) return PCM_ERR_INVALID_ARGUMENT;
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  if (c->cfg.model != PCM_MODEL_NDT_OMP) { c->err = "pcm_ndt_derivatives needs the NDT_OMP model"; return PCM_ERR_UNSUPPORTED; }
  rc = prepare(c);
  if (rc != PCM_OK) return rc;
  auto solver = make_ndt_solver(c);
  solver.gauss_params();
  double pp[6];
  std::memcpy(pp, p, sizeof(pp));
  solver.set_pose(pp);
  solver.angle_derivatives(pp);
  ndtomp::Eval e{};
  rc = pclndt_eval(c, pass, solver.P, &e);
  if (rc != PCM_OK) return rc;
  if (score) *score = e.score;
  if (g) std::memcpy(g, e.g, sizeof(e.g));
  if (H) std::memcpy(H, e.H, sizeof(e.H));
  return PCM_OK;
}

int pcm_align_batch(pcm_ctx* const* ctxs, int n, const float* guesses, pcm_result* host_out, void* device_out) {
  if (!ctxs || n <= 0) return PCM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; i++) {
    CHECK_CTX(ctxs[i]);
    int rc = validate_config(ctxs[i], ctxs[i]->cfg);
    if (rc != PCM_OK) return rc;
  }
  if (ctxs[0]->cfg.model == PCM_MODEL_NDT_OMP) {
    std::vector<pcm_result> res((size_t)n);
    for (int i = 0; i < n; i++) {
      if (ctxs[i]->cfg.model != PCM_MODEL_NDT_OMP) { ctxs[0]->err = "all contexts of a batch must share the model"; return PCM_ERR_INVALID_ARGUMENT; }
      if (ctxs[i]->device != ctxs[0]->device) { ctxs[0]->err = "all contexts of a batch must live on one device"; return PCM_ERR_INVALID_ARGUMENT; }
      for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) { ctxs[0]->err = "a context appears twice in the batch"; return PCM_ERR_INVALID_ARGUMENT; }
    }
    const int worst = pclndt_align_batch(ctxs, n, guesses, res.data());
    if (worst != PCM_OK && res.empty()) return worst;
    if (host_out) std::memcpy(host_out, res.data(), sizeof(pcm_result) * (size_t)n);
    if (device_out && hipMemcpy(device_out, res.data(), sizeof(pcm_result) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return PCM_ERR_HIP;
    return worst;
  }
  return align_batch_impl(ctxs, n, guesses, host_out, device_out);
}

int pcm_linearize(pcm_ctx* c, const double T[16], double H[36], double b[6], double* cost, int32_t* num_inliers) {
  CHECK_CTX(c);
  if (!T) return PCM_ERR_INVALID_ARGUMENT;
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  double s[kPartialStride];
  rc = single_pass(c, T, true, s);
  if (rc != PCM_OK) return rc;
  int t = 0;
  for (int a = 0; a < 6; a++) {
    for (int k = a; k < 6; k++) {
      if (H) { H[a * 6 + k] = s[t]; H[k * 6 + a] = s[t]; }
      t++;
    }
  }
  if (b) for (int a = 0; a < 6; a++) b[a] = s[21 + a];
  if (cost) *cost = s[27];
  if (num_inliers) *num_inliers = (int32_t)s[28];
  return PCM_OK;
}

int pcm_compute_error(pcm_ctx* c, const double T[16], double* cost) {
  CHECK_CTX(c);
  if (!T || !cost) return PCM_ERR_INVALID_ARGUMENT;
  double s[kPartialStride];
  int rc = single_pass(c, T, false, s);
  if (rc != PCM_OK) return rc;
  *cost = s[27];
  return PCM_OK;
}

namespace {
void quat_mul_d(const double* a, const double* b, double* r) {   // Eigen quaternion product, (x,y,z,w)
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
void quat_rot_d(const double* q, const double* v, double* r) {   // Eigen _transformVector
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int a = 0; a < 3; a++) r[a] = v[a] + q[3] * uv[a] + c[a];
}
void quat_to_rot_d(const double* q, double* R) {   // Eigen Quaternion::toRotationMatrix, row-major
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
}  // namespace

int pcm_obs_model(pcm_ctx* c, const pcm_lio_state* s, int extrinsic_est_en, int rematch, pcm_obs_result* out) {
  CHECK_CTX(c);
  if (!s || !out) return PCM_ERR_INVALID_ARGUMENT;
  if (c->cfg.model != PCM_MODEL_P2PLANE) { c->err = "pcm_obs_model needs the P2PLANE model"; return PCM_ERR_INVALID_ARGUMENT; }
  int rc = validate_config(c, c->cfg);
  if (rc != PCM_OK) return rc;
  rc = prepare(c);
  if (rc != PCM_OK) return rc;
  if (!rematch && !c->lio_planes_valid) { c->err = "pcm_obs_model(rematch=0) before any matching call"; return PCM_ERR_INVALID_ARGUMENT; }
  // the float state exactly as the reference casts it (laser_mapping.cc:602-603,669-671)
  LioPose L{};
  double qwl[4], twl[3], Rd[9], ORd[9];
  quat_mul_d(s->rot, s->off_R, qwl);
  quat_rot_d(s->rot, s->off_T, twl);
  quat_to_rot_d(s->rot, Rd);
  quat_to_rot_d(s->off_R, ORd);
  for (int a = 0; a < 4; a++) L.q_wl[a] = (float)qwl[a];
  for (int a = 0; a < 3; a++) { L.t_wl[a] = (float)(twl[a] + s->pos[a]); L.off_t[a] = (float)s->off_T[a]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { L.Rt[i * 3 + j] = (float)Rd[j * 3 + i]; L.off_R[i * 3 + j] = (float)ORd[i * 3 + j]; }

  const uint32_t n = (uint32_t)c->src.n;
  const int tiles = (int)((n + 255u) / 256u);
  Workspace* w = nullptr;
  rc = ensure_ws(c, &w, 1, (size_t)tiles * kLioStride, 2);
  if (rc != PCM_OK) return rc;
  hipStream_t st = c->stream;
  if (c->cfg.sort_source && !c->src_sorted) {   // new scan: order it along the world grid at this state's pose
    double Rwl[9];
    quat_to_rot_d(qwl, Rwl);
    float g[16] = {(float)Rwl[0], (float)Rwl[1], (float)Rwl[2], L.t_wl[0], (float)Rwl[3], (float)Rwl[4], (float)Rwl[5], L.t_wl[1],
                   (float)Rwl[6], (float)Rwl[7], (float)Rwl[8], L.t_wl[2], 0.f, 0.f, 0.f, 1.f};
    SortJob j{c->src.d_pts, c->src_order, n, 0, 0, 0};
    HIPCK(c, hipMemcpyAsync(w->d_guesses, g, sizeof(g), hipMemcpyHostToDevice, st));
    HIPCK(c, hipMemcpyAsync(w->d_jobs, &j, sizeof(j), hipMemcpyHostToDevice, st));
    rc = sort_sources_batched(st, w->d_jobs, 1, n, n, w->d_guesses, c->cfg.voxel_resolution, &w->sort, &c->err);
    if (rc != PCM_OK) return rc;
    c->src_sorted = true;
    c->lio_planes_valid = false;
    if (!rematch) { c->err = "pcm_obs_model(rematch=0) on a new scan"; return PCM_ERR_INVALID_ARGUMENT; }
  }
  const bool ref = (c->cfg.flags & PCM_FLAG_LIO_REFERENCE_SEMANTICS) != 0;
  if (ref) {
    rc = lio_members_resize(c, n);
    if (rc != PCM_OK) return rc;
  }
  Geom geom = pick_geom(n, 1);
  KernelParams kp = kernel_params(c->cfg, geom);
  kp.lio_rematch = rematch ? 1 : 0;
  kp.lio_extrinsic = extrinsic_est_en ? 1 : 0;
  kp.lio_ref = !ref ? 0 : ((c->cfg.sort_source && c->src_sorted) ? 2 : 1);
  PairDesc d;
  fill_desc(c, &d, w->d_partials);
  d.lio = L;
  d.lio_aux = c->lio_aux;
  PairState ps;
  const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  init_state(ps, ident);
  HIPCK(c, hipMemcpyAsync(w->d_descs, &d, sizeof(d), hipMemcpyHostToDevice, st));
  HIPCK(c, hipMemcpyAsync(w->d_states, &ps, sizeof(ps), hipMemcpyHostToDevice, st));
  launch_lio_obs(st, w->d_descs, w->d_states, kp);
  launch_lio_finish(st, w->d_partials, tiles, w->d_sums);
  HIPCK(c, hipGetLastError());
  double sums[kLioStride];
  HIPCK(c, hipMemcpyAsync(sums, w->d_sums, sizeof(double) * kLioStride, hipMemcpyDeviceToHost, st));
  HIPCK(c, hipStreamSynchronize(st));
  if (rematch) c->lio_planes_valid = true;
  int t = 0;
  for (int a = 0; a < 12; a++) for (int b = a; b < 12; b++) { out->HTH[a * 12 + b] = sums[t]; out->HTH[b * 12 + a] = sums[t]; t++; }
  for (int a = 0; a < 12; a++) out->HTh[a] = sums[78 + a];
  out->sum_h2 = sums[90];
  out->n_eff = (int32_t)sums[91];
  out->valid = out->n_eff >= 1 ? 1 : 0;
  c->stats.linearize_launches += 1;
  c->stats.point_passes += n;
  return PCM_OK;
}

namespace {
// grow the target point log to hold `need` points (keeps the content)
int reserve_target(pcm_ctx* c, size_t need) {
  if (need <= c->tgt.cap && !c->tgt.borrowed) return PCM_OK;
  const size_t cap = std::max(need, c->tgt.cap + c->tgt.cap / 2 + 1024);
  float4* nb = nullptr;
  HIPCK(c, hipMalloc(&nb, sizeof(float4) * cap));
  if (c->tgt.n) HIPCK(c, hipMemcpyAsync(nb, c->tgt.d_pts, sizeof(float4) * c->tgt.n, hipMemcpyDeviceToDevice, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  const size_t n = c->tgt.n;
  const uint64_t tag = c->tgt.tag;
  c->tgt.drop_buffer();
  c->tgt.d_pts = nb; c->tgt.cap = cap; c->tgt.n = n; c->tgt.tag = tag;
  return PCM_OK;
}
}  // namespace

int pcm_target_insert(pcm_ctx* c, const void* points, size_t n, size_t stride_bytes, int memory) {
  CHECK_CTX(c);
  if (!points && n) { c->err = "null point buffer"; return PCM_ERR_INVALID_ARGUMENT; }
  if (stride_bytes < 3 * sizeof(float) || (stride_bytes % sizeof(float)) != 0) { c->err = "stride must be a multiple of 4 and >= 12 bytes"; return PCM_ERR_INVALID_ARGUMENT; }
  if (n == 0) return PCM_OK;
  HIPCK(c, hipSetDevice(c->device));
  int rc = reserve_target(c, c->tgt.n + n);
  if (rc != PCM_OK) return rc;
  rc = load_points_to_device(c->stream, points, n, stride_bytes, memory, c->next_seq, c->tgt.d_pts + c->tgt.n, &c->err);
  if (rc != PCM_OK) return rc;
  HIPCK(c, hipStreamSynchronize(c->stream));
  c->tgt.n += n;
  c->next_seq += (uint32_t)n;
  c->tgt.tag = 0;
  c->map.valid = false;
  c->tgt_dynamic = true;
  return PCM_OK;
}

int pcm_map_incremental(pcm_ctx* c, const pcm_lio_state* s, float filter_size_map, int ekf_inited, size_t* num_added) {
  CHECK_CTX(c);
  if (!s) return PCM_ERR_INVALID_ARGUMENT;
  if (c->src.n == 0) { c->err = "pcm_map_incremental without a source scan"; return PCM_ERR_NO_INPUT; }
  HIPCK(c, hipSetDevice(c->device));
  const uint32_t n = (uint32_t)c->src.n;
  int rc = reserve_target(c, c->tgt.n + n);
  if (rc != PCM_OK) return rc;
  const bool have_nn = c->lio_planes_valid && c->nn != nullptr && c->map.valid && ekf_inited;
  LioStateD L;
  for (int a = 0; a < 4; a++) { L.rot[a] = s->rot[a]; L.off_R[a] = s->off_R[a]; }
  for (int a = 0; a < 3; a++) { L.pos[a] = s->pos[a]; L.off_T[a] = s->off_T[a]; }
  const bool reordered = c->cfg.sort_source && c->src_sorted;
  const float4* scan = reordered ? c->src_order : c->src.d_pts;
  uint32_t added = 0;
  rc = map_incremental_device(c->stream, scan, reordered, n, L, filter_size_map, have_nn ? c->nn : nullptr, have_nn ? c->map.pts : nullptr, c->next_seq,
                              c->tgt.d_pts + c->tgt.n, &added, &c->err);
  if (rc != PCM_OK) return rc;
  c->tgt.n += added;
  c->next_seq += added;
  c->tgt.tag = 0;
  if (added) { c->map.valid = false; c->tgt_dynamic = true; }
  if (num_added) *num_added = added;
  return PCM_OK;
}

// ---------------------------------------------------------------------------
// One LiDAR frame of LaserMapping::Run on device buffers (jueying_lio/src/laser_mapping.cc:323-347 front end, :525-583 back end).
// ---------------------------------------------------------------------------
int pcm_lio_frame_begin(pcm_ctx* c, const void* custom_points, size_t n, int memory, const pcm_lio_frame_params* prm, const pcm_imu_pose* poses, int npose,
                        const pcm_lio_state* end_state, size_t* n_scan) {
  CHECK_CTX(c);
  if ((!custom_points && n) || !prm || !n_scan || (npose >= 2 && (!poses || !end_state))) return PCM_ERR_INVALID_ARGUMENT;
  if (n > 0xffffffffull) { c->err = "too many points"; return PCM_ERR_INVALID_ARGUMENT; }
  if (!(prm->leaf_size >= 0.f)) { c->err = "leaf_size must be >= 0"; return PCM_ERR_INVALID_ARGUMENT; }
  *n_scan = 0;
  if (n == 0) { c->err = "empty frame"; return PCM_ERR_NO_INPUT; }
  HIPCK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  // frame arena: [raw message | filtered records | down-sampled records | IMU poses | scratch of the operators]; grow-only, so the
  // steady state makes no allocation.  The raw message is the only host -> device copy of the frame.
  const size_t o_raw = 0, o_flt = o_raw + up(n * 20), o_ds = o_flt + up(n * 48), o_pose = o_ds + up(n * 48);
  const size_t o_scr = o_pose + up(sizeof(pcm_imu_pose) * (size_t)std::max(npose, 1));
  const size_t need = o_scr + std::max(livox_filter_scratch_bytes(n), voxel_downsample_scratch_bytes(n));
  if (c->pre_arena_cap < need) {
    if (c->pre_arena) hipFree(c->pre_arena);
    c->pre_arena = nullptr; c->pre_arena_cap = 0;
    HIPCK(c, hipMalloc(&c->pre_arena, need + need / 4));
    c->pre_arena_cap = need + need / 4;
  }
  char* A = c->pre_arena;
  const void* d_raw = custom_points;
  if (memory == PCM_MEM_HOST) {
    HIPCK(c, hipMemcpyAsync(A + o_raw, custom_points, n * 20, hipMemcpyHostToDevice, st));
    d_raw = A + o_raw;
  }
  // 1. PointCloudPreprocess::AviaHandler  (pointcloud_preprocess.cc:44-88)
  size_t n_flt = 0;
  int rc = livox_filter_device(st, d_raw, n, prm->num_scans, prm->point_filter_num, prm->blind, A + o_flt, &n_flt, A + o_scr, &c->err);
  if (rc != PCM_OK) return rc;
  if (n_flt == 0) { c->err = "no point of the frame passed the driver-message filter"; return PCM_ERR_NO_INPUT; }
  // 2. ImuProcess::UndistortPcl backward loop  (imu_processing.hpp:245-285): in place on the filtered records (time stamp = curvature).
  //    The reference sorts the scan by time first (:177-178); a Livox message is time-ordered, and the compensation of a point
  //    depends on its own stamp only, so the message order is kept.
  if (npose >= 2) {
    LioStateD s;
    for (int a = 0; a < 4; a++) { s.rot[a] = end_state->rot[a]; s.off_R[a] = end_state->off_R[a]; }
    for (int a = 0; a < 3; a++) { s.pos[a] = end_state->pos[a]; s.off_T[a] = end_state->off_T[a]; }
    HIPCK(c, hipMemcpyAsync(A + o_pose, poses, sizeof(pcm_imu_pose) * (size_t)npose, hipMemcpyHostToDevice, st));
    rc = undistort_device(st, A + o_flt, n_flt, 48, 36, reinterpret_cast<const pcm_imu_pose*>(A + o_pose), npose, s, &c->err);   // PointXYZINormal::curvature: byte 36
    if (rc != PCM_OK) return rc;
  }
  // 3. voxel_scan_.filter()  (laser_mapping.cc:323-328); leaf 0 = no down-sampling
  const char* d_scan = A + o_flt;
  size_t n_ds = n_flt;
  if (prm->leaf_size > 0.f) {
    rc = voxel_downsample_device(st, A + o_flt, n_flt, 48, prm->leaf_size, reinterpret_cast<float*>(A + o_ds), &n_ds, A + o_scr, &c->err);
    if (rc != PCM_OK) return rc;
    d_scan = A + o_ds;
  }
  if (n_ds == 0) { c->err = "empty scan after down-sampling"; return PCM_ERR_NO_INPUT; }
  // 4. the down-sampled scan (scan_down_body_) becomes the source of this object: device -> device, no host copy
  if (c->cfg.flags & PCM_FLAG_LIO_REFERENCE_SEMANTICS) {   // one resize of residuals_ / point_selected_surf_ per frame  laser_mapping.cc:335-339
    rc = lio_members_resize(c, n_ds);
    if (rc != PCM_OK) return rc;
  }
  rc = set_cloud(c, &c->src, d_scan, n_ds, 48, PCM_MEM_DEVICE, 0, false);
  if (rc != PCM_OK) return rc;
  c->src_sorted = false;
  c->lio_planes_valid = false;
  c->srcmap.valid = false;
  c->src_cov_valid = false;
  c->user_cov[0].clear();
  *n_scan = n_ds;
  return PCM_OK;
}

int pcm_lio_frame_end(pcm_ctx* c, const pcm_lio_state* s, float filter_size_map, int ekf_inited, size_t* num_added) { return pcm_map_incremental(c, s, filter_size_map, ekf_inited, num_added); }

int pcm_get_source(pcm_ctx* c, float* out_xyz, size_t capacity_points, size_t* n) {
  CHECK_CTX(c);
  if (!n) return PCM_ERR_INVALID_ARGUMENT;
  *n = c->src.n;
  if (!out_xyz) return PCM_OK;
  if (capacity_points < c->src.n) { c->err = "pcm_get_source: buffer too small"; return PCM_ERR_INVALID_ARGUMENT; }
  std::vector<float4> tmp(c->src.n);
  HIPCK(c, hipStreamSynchronize(c->stream));
  HIPCK(c, hipMemcpy(tmp.data(), c->src.d_pts, sizeof(float4) * c->src.n, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < c->src.n; i++) { out_xyz[3 * i] = tmp[i].x; out_xyz[3 * i + 1] = tmp[i].y; out_xyz[3 * i + 2] = tmp[i].z; }
  return PCM_OK;
}

int pcm_get_target(pcm_ctx* c, float* out_xyz, size_t capacity_points, size_t* n) {
  CHECK_CTX(c);
  if (!n) return PCM_ERR_INVALID_ARGUMENT;
  if (c->tgt.n && c->cfg.map_capacity > 0 && !c->map.valid && c->src.n) {
    int rc = prepare(c);   // apply a pending LRU eviction so the log is the current map
    if (rc != PCM_OK) return rc;
  }
  *n = c->tgt.n;
  if (!out_xyz) return PCM_OK;
  if (capacity_points < c->tgt.n) { c->err = "pcm_get_target: buffer too small"; return PCM_ERR_INVALID_ARGUMENT; }
  std::vector<float4> tmp(c->tgt.n);
  HIPCK(c, hipMemcpy(tmp.data(), c->tgt.d_pts, sizeof(float4) * c->tgt.n, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < c->tgt.n; i++) { out_xyz[3 * i] = tmp[i].x; out_xyz[3 * i + 1] = tmp[i].y; out_xyz[3 * i + 2] = tmp[i].z; }
  return PCM_OK;
}

int pcm_get_planes(pcm_ctx* c, float* out, size_t n) {
  CHECK_CTX(c);
  if (!out || n != c->src.n || !c->planes) { c->err = "pcm_get_planes: call pcm_linearize first; n must equal the source size"; return PCM_ERR_INVALID_ARGUMENT; }
  HIPCK(c, hipMemcpy(out, c->planes, sizeof(float4) * n, hipMemcpyDeviceToHost));
  return PCM_OK;
}

int pcm_get_lio_members(pcm_ctx* c, float* residuals, uint8_t* selected, size_t n) {
  CHECK_CTX(c);
  if (!(c->cfg.flags & PCM_FLAG_LIO_REFERENCE_SEMANTICS) || !c->lio_aux || n != c->lio_aux_n || n != c->src.n) {
    c->err = "pcm_get_lio_members: needs PCM_FLAG_LIO_REFERENCE_SEMANTICS and n equal to the source size";
    return PCM_ERR_INVALID_ARGUMENT;
  }
  std::vector<float2> tmp(n);
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (n) HIPCK(c, hipMemcpy(tmp.data(), c->lio_aux, sizeof(float2) * n, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++) {
    if (residuals) residuals[i] = tmp[i].x;
    if (selected) selected[i] = tmp[i].y != 0.f ? 1 : 0;
  }
  return PCM_OK;
}

int pcm_debug_phase_cycles(pcm_ctx* c, uint64_t out[8]) {
  if (!c || !out) return PCM_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < 8; k++) { out[k] = c->phase_cycles[k]; c->phase_cycles[k] = 0; }
  return PCM_OK;
}

int pcm_get_stats(pcm_ctx* c, pcm_stats* out) {
  if (!c || !out) return PCM_ERR_INVALID_ARGUMENT;
  *out = c->stats;
  return PCM_OK;
}

int pcm_reset_stats(pcm_ctx* c) {
  if (!c) return PCM_ERR_INVALID_ARGUMENT;
  const uint64_t v = c->stats.target_voxels, s = c->stats.target_slots;
  c->stats = pcm_stats{};
  c->stats.target_voxels = v;
  c->stats.target_slots = s;
  return PCM_OK;
}

int pcm_set_profiling(pcm_ctx* c, int on) {
  if (!c) return PCM_ERR_INVALID_ARGUMENT;
  c->profiling = on;
  return PCM_OK;
}

}  // extern "C"

// voxel_hash.hip -- device-side build of the submap voxel hash (pcm_set_target) and
// the batched re-ordering of new scans.
//
// Replaces, for the MI355X path:
//   - IVox::AddPoints bulk insert        /root/reference/src/jueying_lio/include/ivox3d/ivox3d.h:256-281
//   - GaussianVoxelMap::create_bucket_table (linear-probed buckets, atomicCAS insert)
//                                        /root/reference/src/pointcloud_match/fast_gicp/src/fast_gicp/cuda/gaussian_voxelmap.cu:21-58,258-289
// Design (not a port): the reference inserts every POINT with an atomicCAS into a
// table of one bucket per voxel and doubles the table until <1 % of points fail.
// Here the points are radix-sorted once by (brick, voxel-in-brick) key (stable, so a
// voxel keeps its points in input order), the hash holds one slot per 8x8x8 BRICK
// with a 512-bit occupancy mask, the table is sized up front from the brick count
// (load <= 0.25, unbounded probing, no insertion ever fails), and points and voxels
// are laid out brick-major so everything under a search tile is a few contiguous
// runs (layout: pcm_device.h).
#include "pcm_device.h"
#include "pcm_host.h"
#include "dev_linalg.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace pcm {

// strided xyz records -> compact float4; .w = insertion sequence number (raw uint bits): the
// "last touched" order of the sliding map's LRU eviction is derived from it
__global__ void k_load_points(const char* __restrict__ base, size_t stride, uint32_t n, uint32_t seq0, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = reinterpret_cast<const float*>(base + (size_t)i * stride);
  out[i] = make_float4(p[0], p[1], p[2], __uint_as_float(seq0 + i));
}

__global__ void k_point_keys(const float4* __restrict__ pts, uint32_t n, float res, float inv_res, int mode,
                             uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, int* __restrict__ oor) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int c[3] = {voxel_coord(p.x, res, inv_res, mode), voxel_coord(p.y, res, inv_res, mode), voxel_coord(p.z, res, inv_res, mode)};
  bool bad = !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z));
  for (int a = 0; a < 3; a++) {
    // keep a few cells of slack so that neighbour offsets never leave the key range
    if (c[a] < -kCoordBias + 16 || c[a] > kCoordBias - 17) { bad = true; c[a] = 0; }
  }
  if (bad) atomicOr(oor, 1);
  keys[i] = point_key(c[0], c[1], c[2]);
  idx[i] = i;
}

constexpr int kBrickCountShards = 32;   // counters of the brick heads (k_head_flags), summed on the host
constexpr int kCtrInts = 8 + kBrickCountShards;   // the build's counter record: 8 scalars + the shards
// voxel-head flags (input of the rank scan) + number of brick heads
__global__ void k_head_flags(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ vflag, unsigned int* __restrict__ nbricks) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool bhead = false;
  if (i < n) {
    const uint64_t k = keys[i];
    const uint64_t kp = i ? keys[i - 1] : ~k;
    vflag[i] = k != kp ? 1u : 0u;
    bhead = (k >> 9) != (kp >> 9);
  }
  __shared__ unsigned int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const unsigned long long m = __ballot(bhead);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt, (unsigned int)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(nbricks + (blockIdx.x & (kBrickCountShards - 1)), s_cnt);   // sharded: one word took 130 us for 13 000 blocks
}

// one insertion per brick: the first point of each brick claims a slot
__global__ void k_insert_bricks(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vrank, uint32_t n, BrickSlot* __restrict__ slots, uint32_t mask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t bkey = keys[i] >> 9;
  if (i != 0 && (keys[i - 1] >> 9) == bkey) return;
  const int bx = (int)(bkey >> 36) - kBrickBias, by = (int)((bkey >> 18) & 0x3ffff) - kBrickBias, bz = (int)(bkey & 0x3ffff) - kBrickBias;
  uint32_t h = hash_coord(bx, by, bz) & mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&slots[h].key), (unsigned long long)kEmptyKey, (unsigned long long)bkey);
    if (prev == kEmptyKey) {  // brick keys are unique, so a claimed slot is ours alone
      slots[h].vox_base = vrank[i];
      slots[h].nvox = 0;
      slots[h].pt_start = 0;
      slots[h].npts = 0;
      return;
    }
    h = (h + 1) & mask;
  }
}

// one pass per voxel: first point index, occupancy bit, voxel count of the brick
__global__ void k_fill_voxels(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ vrank, uint32_t n,
                              uint32_t nvox_total, uint32_t* __restrict__ vox_start, BrickSlot* __restrict__ slots, uint32_t* __restrict__ bmask, uint32_t mask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) vox_start[nvox_total] = n;
  if (i >= n || !vflag[i]) return;
  vox_start[vrank[i]] = i;
  const uint64_t bkey = keys[i] >> 9;
  const uint32_t li = (uint32_t)(keys[i] & 511u);
  const int bx = (int)(bkey >> 36) - kBrickBias, by = (int)((bkey >> 18) & 0x3ffff) - kBrickBias, bz = (int)(bkey & 0x3ffff) - kBrickBias;
  uint32_t h = hash_coord(bx, by, bz) & mask;
  while (slots[h].key != bkey) h = (h + 1) & mask;  // inserted by the previous kernel
  atomicOr(&bmask[(size_t)h * 16 + (li >> 5)], 1u << (li & 31));
  atomicAdd(&slots[h].nvox, 1u);
}

// rank prefix of every mask word; point range of the brick
__global__ void k_finalize_bricks(BrickSlot* __restrict__ slots, const uint32_t* __restrict__ bmask, uint16_t* __restrict__ bpref, const uint32_t* __restrict__ vox_start,
                                  uint32_t cap) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= cap || slots[h].key == kEmptyKey) return;
  const uint32_t p0 = vox_start[slots[h].vox_base], p1 = vox_start[slots[h].vox_base + slots[h].nvox];
  slots[h].pt_start = p0;
  slots[h].npts = p1 - p0;
  uint32_t run = 0;
  for (int w = 0; w < 16; w++) {
    bpref[(size_t)h * 16 + w] = (uint16_t)run;
    run += (uint32_t)__popc(bmask[(size_t)h * 16 + w]);
  }
}

// points into sorted order, tagged in .w (raw bits): voxel-in-brick in bits 0..8; the FIRST point of a voxel's run also carries
// bit 31 and, in bits 9..30, the number of points of its voxel (clamped to kMaxTagCount) -- a reader of a staged brick finds the
// voxel heads and the length of every voxel's run without comparing neighbouring tags or walking to the next head
__global__ void k_gather_points(const float4* __restrict__ in, const uint32_t* __restrict__ idx, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vrank,
                                const uint32_t* __restrict__ vox_start, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = in[idx[i]];
  uint32_t tag = (uint32_t)(keys[i] & 511u);
  if (i == 0 || keys[i - 1] != keys[i]) {   // vrank = exclusive scan of the head flags: at a head it is the voxel's index
    const uint32_t cnt = vox_start[vrank[i] + 1u] - i;
    tag |= 0x80000000u | ((cnt < kMaxTagCount ? cnt : kMaxTagCount) << 9);
  }
  p.w = __int_as_float((int)tag);
  out[i] = p;
}

// most points in one voxel (PCM_FLAG_REFERENCE_KNN_ORDER sizes a private array by it)
__global__ void __launch_bounds__(256) k_max_voxel_points(const uint32_t* __restrict__ vox_start, uint32_t nvox, unsigned int* __restrict__ out) {
  unsigned int c = 0;
  for (uint32_t v = blockIdx.x * 1024u + threadIdx.x; v < nvox && v < (blockIdx.x + 1u) * 1024u; v += 256u) c = max(c, vox_start[v + 1] - vox_start[v]);   // 4 voxels per lane
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c = max(c, (unsigned int)__shfl_xor((int)c, off, 64));
  __shared__ unsigned int s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { c = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])); if (c) atomicMax(out, c); }   // one atomic per 1024 voxels
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
// scratch arrays come from the device's stream-ordered memory pool (hipMallocAsync): after the first frames an allocation
// is a pool hit, no driver call and no implicit device synchronisation -- the sliding map rebuilds every frame
static inline void sfree(hipStream_t stream, void* p) { if (p) (void)hipFreeAsync(p, stream); }

// ---------------------------------------------------------------------------
// Sliding-map eviction: IVox keeps its voxels in a least-recently-touched list and
// drops the tail when a new voxel brings the count to `capacity`
// (/root/reference/src/jueying_lio/include/ivox3d/ivox3d.h:256-281).  A voxel's place
// in that list is decided by its LAST insertion, i.e. by the largest insertion
// sequence number among its points; so the survivors of a batch are the
// (capacity - 1) voxels with the largest such number.  (Difference from the
// sequential list: a voxel evicted and re-created inside one batch would restart
// empty in the reference; here it keeps its points.)
// ---------------------------------------------------------------------------
__global__ void k_voxel_last(const float4* __restrict__ log, const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ vrank,
                             uint32_t n, uint32_t* __restrict__ vlast) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicMax(&vlast[vrank[i] + vflag[i] - 1u], __float_as_uint(log[idx_s[i]].w));
}

__global__ void k_mark_alive(const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ vrank, const uint32_t* __restrict__ vlast,
                             uint32_t cutoff, uint32_t n, uint32_t* __restrict__ alive) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  alive[idx_s[i]] = vlast[vrank[i] + vflag[i] - 1u] >= cutoff ? 1u : 0u;
}

__global__ void k_compact_log(const float4* __restrict__ in, const uint32_t* __restrict__ alive, const uint32_t* __restrict__ pos, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && alive[i]) out[pos[i]] = in[i];
}

// ---------------------------------------------------------------------------
// Gaussian voxel statistics of the NDT models:
//   sums .......... gaussian_voxelmap.cu:122-148 (x and x x^T per voxel)
//   mean / cov .... gaussian_voxelmap.cu:178-198  cov = (sum x x^T - mean * sum x^T) / n
//   MIN_EIG ....... covariance_regularization.cu:83-97  eigenvalues clamped to >= 1e-3
// The reference adds the 13 numbers of every point with float atomics (run-to-run
// order dependent); here a voxel's points are one contiguous run in input order and
// one lane sums them in double -- deterministic, no atomics.
// ---------------------------------------------------------------------------
__global__ void k_gauss_voxels(const float4* __restrict__ pts, const uint32_t* __restrict__ vox_start, uint32_t nvox, GaussVoxel* __restrict__ out) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t p0 = vox_start[v], p1 = vox_start[v + 1];
  double sx[3] = {0.0, 0.0, 0.0}, sxx[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (uint32_t k = p0; k < p1; k++) {
    const float4 p = pts[k];
    const float x[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      sx[a] += (double)x[a];
#pragma unroll
      for (int b = 0; b < 3; b++) sxx[a * 3 + b] += (double)(x[a] * x[b]);   // the product is a float, as in the reference
    }
  }
  const double nn = (double)(p1 - p0);
  double mean[3], cov[9];
#pragma unroll
  for (int a = 0; a < 3; a++) mean[a] = sx[a] / nn;
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) cov[a * 3 + b] = (sxx[a * 3 + b] - mean[a] * sx[b]) / nn;
  }
  // covariance_regularization_mineig (covariance_regularization.cu:83-97) on the float voxel covariance: computeDirect (closed
  // form, reads the lower triangle; dev_linalg.h), eigenvalues clamped at 1e-3, V diag V^-1 with the general 3x3 inverse
  float cf[9], w[3], Vf[9], Vi[9], VD[9], c[9];
#pragma unroll
  for (int a = 0; a < 9; a++) cf[a] = (float)cov[a];
  selfadjoint3_direct(cf, w, Vf);
#pragma unroll
  for (int k = 0; k < 3; k++) w[k] = fmaxf(1e-3f, w[k]);
  inv3<float>(Vf, Vi);
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) VD[a * 3 + b] = Vf[a * 3 + b] * w[b];
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) c[a * 3 + b] = (VD[a * 3 + 0] * Vi[0 * 3 + b] + VD[a * 3 + 1] * Vi[1 * 3 + b]) + VD[a * 3 + 2] * Vi[2 * 3 + b];
  }
  GaussVoxel g;
  g.mx = (float)mean[0]; g.my = (float)mean[1]; g.mz = (float)mean[2];
  g.n = (int32_t)(p1 - p0);
  g.c00 = c[0]; g.c01 = c[1]; g.c02 = c[2]; g.c11 = c[4]; g.c12 = c[5]; g.c22 = c[8];
  g.pad0 = 0.f; g.pad1 = 0.f;
  out[v] = g;
}

// ---------------------------------------------------------------------------
// Sliding map: the sorted index of the point log is PERSISTENT (TargetMap::keys_s / idx_s), a batch of appended points is
// merged into it instead of re-sorting the whole log, the tables are rebuilt from the merged index in capacity-managed arrays
// (no allocation per frame), and an LRU eviction compacts log and index in place instead of building everything a second time.
//   IVox::AddPoints + LRU cache   /root/reference/src/jueying_lio/include/ivox3d/ivox3d.h:256-281
// Policy of the linear-probed brick table under insert + evict: the table is rebuilt from the merged index every
// batch (one slot per occupied brick, load <= 0.25) -- bricks never need a tombstone because no slot outlives a batch; what is
// incremental is everything that is expensive: the 63-bit sort of the log (replaced by a sort of the batch + two merge passes)
// and the second build an eviction used to trigger.  The result is identical to a full build of the same log, structure for structure.
// ---------------------------------------------------------------------------
__global__ void k_point_keys_at(const float4* __restrict__ pts, uint32_t first, uint32_t m, float res, float inv_res, int mode, uint64_t* __restrict__ keys,
                                uint32_t* __restrict__ idx, int* __restrict__ oor) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const float4 p = pts[first + j];
  int c[3] = {voxel_coord(p.x, res, inv_res, mode), voxel_coord(p.y, res, inv_res, mode), voxel_coord(p.z, res, inv_res, mode)};
  bool bad = !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z));
  for (int a = 0; a < 3; a++) {
    if (c[a] < -kCoordBias + 16 || c[a] > kCoordBias - 17) { bad = true; c[a] = 0; }
  }
  if (bad) atomicOr(oor, 1);
  keys[j] = point_key(c[0], c[1], c[2]);
  idx[j] = first + j;
}

// stable merge of the sorted batch (keys_b, m) into the sorted index (keys_a, n): an old entry goes behind the batch entries with
// a SMALLER key, a batch entry behind every old entry with a smaller OR EQUAL key (a voxel keeps its points in insertion order)
__global__ void k_merge_old(const uint64_t* __restrict__ keys_a, const uint32_t* __restrict__ idx_a, uint32_t n, const uint64_t* __restrict__ keys_b, uint32_t m,
                            uint64_t* __restrict__ keys_o, uint32_t* __restrict__ idx_o) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys_a[i];
  uint32_t lo = 0, hi = m;   // lower_bound: first batch key >= k
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys_b[mid] < k) lo = mid + 1; else hi = mid; }
  keys_o[i + lo] = k;
  idx_o[i + lo] = idx_a[i];
}
__global__ void k_merge_new(const uint64_t* __restrict__ keys_a, uint32_t n, const uint64_t* __restrict__ keys_b, const uint32_t* __restrict__ idx_b, uint32_t m,
                            uint64_t* __restrict__ keys_o, uint32_t* __restrict__ idx_o) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint64_t k = keys_b[j];
  uint32_t lo = 0, hi = n;   // upper_bound: first old key > k
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys_a[mid] <= k) lo = mid + 1; else hi = mid; }
  keys_o[j + lo] = k;
  idx_o[j + lo] = idx_b[j];
}

// counters for the host in ONE record (read back with one copy into pinned memory): [4] voxels = rank + flag of the last point
__global__ void k_count_voxels(const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ vrank, uint32_t n, int* __restrict__ ctr) { ctr[4] = (int)(vrank[n - 1] + vflag[n - 1]); }
__global__ void k_count_alive(const uint32_t* __restrict__ alive, const uint32_t* __restrict__ pos, uint32_t n, int* __restrict__ ctr) { ctr[5] = (int)(pos[n - 1] + alive[n - 1]); }
// last touch of every voxel = sequence number of its last point (a voxel's points are in insertion order): no atomics
__global__ void k_voxel_last_from_firsts(const float4* __restrict__ log, const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ vox_first, uint32_t nvox, uint32_t n,
                                         uint32_t* __restrict__ vlast) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t p1 = v + 1 < nvox ? vox_first[v + 1] : n;
  vlast[v] = __float_as_uint(log[idx_s[p1 - 1]].w);
}
// alive[log position] = its voxel's last touch is not older than the cut-off (read on the device: *cutoff)
__global__ void k_mark_alive_dev(const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ vrank, const uint32_t* __restrict__ vlast,
                                 const uint32_t* __restrict__ cutoff, uint32_t n, uint32_t* __restrict__ alive) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  alive[idx_s[i]] = vlast[vrank[i] + vflag[i] - 1u] >= *cutoff ? 1u : 0u;
}

// eviction: survivors of the sorted index, re-pointed at the compacted log
__global__ void k_compact_index(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in, const uint32_t* __restrict__ alive_log, const uint32_t* __restrict__ pos_log,
                                const uint32_t* __restrict__ pos_idx, uint32_t n, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t l = idx_in[i];
  if (!alive_log[l]) return;
  keys_out[pos_idx[i]] = keys_in[i];
  idx_out[pos_idx[i]] = pos_log[l];
}
__global__ void k_alive_sorted(const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ alive_log, uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = alive_log[idx_s[i]];
}
// A voxel that this batch touched, that existed before the batch and whose last touch BEFORE the batch is older than the eviction
// cut-off: the reference's sequential list (ivox3d.h:256-281) may have dropped it before the batch reached it and re-created it
// with the batch's points alone; the batch rule here keeps it whole.  Counted, reported (pcm_stats.lru_batch_hazards), not hidden.
__global__ void k_lru_hazards(const float4* __restrict__ log, const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ vox_first, const uint32_t* __restrict__ vlast,
                              uint32_t nvox, uint32_t n, const float4* __restrict__ first_of_batch, const uint32_t* __restrict__ cutoff_p, unsigned int* __restrict__ out) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t seq0 = __float_as_uint(first_of_batch->w), cutoff = *cutoff_p;
  if (vlast[v] < seq0) return;                                        // not touched by this batch
  const uint32_t p0 = vox_first[v], p1 = v + 1 < nvox ? vox_first[v + 1] : n;
  uint32_t k = p1 - 1;
  while (k > p0 && __float_as_uint(log[idx_s[k]].w) >= seq0) k--;     // a voxel's points are in insertion order
  const uint32_t before = __float_as_uint(log[idx_s[k]].w);
  if (before < seq0 && before < cutoff) atomicAdd(out, 1u);
}
__global__ void k_voxel_firsts(const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ vrank, uint32_t n, uint32_t* __restrict__ vox_first) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && vflag[i]) vox_first[vrank[i]] = i;
}

// Sub-voxel order (build_target_map(..., subsort)): a voxel's points are one run of the sorted index in INPUT order, i.e. scattered
// all over the voxel; a consumer that works on 64 consecutive map points at a time (k_covariances) wants them to be one small patch
// instead.  Key = (voxel rank, Morton code of the point's cell on a grid 8x finer); one more radix sort permutes the index inside
// the voxels only (the voxel keys stay where they are).  Only for maps whose within-voxel order nothing else depends on.
__device__ inline uint32_t spread3(uint32_t v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4); }
__global__ void k_subvoxel_keys(const float4* __restrict__ pts, const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ vrank, uint32_t n, float inv_res,
                                uint64_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[idx_s[i]];
  const int cx = (int)floorf(p.x * inv_res * 8.f), cy = (int)floorf(p.y * inv_res * 8.f), cz = (int)floorf(p.z * inv_res * 8.f);
  const uint32_t m = spread3((uint32_t)cx & 7u) | (spread3((uint32_t)cy & 7u) << 1) | (spread3((uint32_t)cz & 7u) << 2);
  keys[i] = ((uint64_t)(vrank[i] + vflag[i] - 1u) << 9) | m;   // vrank is the EXCLUSIVE scan of the head flags: a head holds its voxel's index, the rest of the run one more
}

// Scratch of one build: a sliding map keeps a grow-only arena (an update then makes no allocation call at all: two dozen
// hipMallocAsync / hipFreeAsync pairs were a third of an update's wall time); any other build takes from the stream-ordered pool.
struct BuildScratch {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  std::vector<void*> pooled;
  hipError_t get(void** p, size_t bytes, hipStream_t stream) {
    const size_t b = (bytes + 255) & ~(size_t)255;
    if (base && used + b <= cap) { *p = base + used; used += b; return hipSuccess; }
    const hipError_t e = hipMallocAsync(p, bytes ? bytes : 1, stream);
    if (e == hipSuccess) pooled.push_back(*p);
    return e;
  }
  void release(hipStream_t stream) {
    for (void* q : pooled) (void)hipFreeAsync(q, stream);
    pooled.clear();
    used = 0;
  }
};

template <typename T>
static int grow(T** p, size_t* cap, size_t need, size_t keep_elems, hipStream_t stream, std::string* err, const char* what) {
  if (need <= *cap) return PCM_OK;
  const size_t nc = need + need / 4 + 1024;
  T* q = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&q), sizeof(T) * nc) != hipSuccess) { *err = std::string("hipMalloc(") + what + ")"; return PCM_ERR_HIP; }
  if (*p && keep_elems) {
    if (hipMemcpyAsync(q, *p, sizeof(T) * keep_elems, hipMemcpyDeviceToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { hipFree(q); *err = std::string("copy(") + what + ")"; return PCM_ERR_HIP; }
  } else if (*p) {
    (void)hipStreamSynchronize(stream);   // queued kernels may still read the old array
  }
  if (*p) hipFree(*p);
  *p = q; *cap = nc;
  return PCM_OK;
}

// Build the voxel hash of the point log `d_pts` into `map`, or -- `n_indexed` > 0: the first n_indexed log points are what
// map->keys_s / idx_s index -- merge the points appended since.  Host syncs: voxel / brick counts (array sizes), once more after
// an eviction.
int build_target_map(hipStream_t stream, float4* d_pts, uint32_t* n_inout, float res, int coord_mode, bool want_gauss, uint32_t capacity_voxels, TargetMap* map,
                     std::string* err, bool keep_order, uint32_t n_indexed, uint32_t* lru_hazards, bool subsort) {
  uint32_t n = *n_inout;
  if (n == 0) { map->release(); *err = "empty target cloud"; return PCM_ERR_NO_INPUT; }
  const bool incremental = n_indexed > 0 && n_indexed <= n && map->keys_s && map->idx_s && map->index_n == n_indexed && map->res == res && map->coord_mode == coord_mode && !want_gauss && !keep_order;
  uint64_t *keys_b = nullptr, *keys_bs = nullptr;
  uint32_t *idx_b = nullptr, *idx_bs = nullptr, *vflag = nullptr, *vrank = nullptr;
  int* d_flags = nullptr;  // [0] out-of-range flag, [1] brick count, [2] most points in one voxel, [3] LRU hazards, [4] voxels, [5] log points alive
  void *tmp = nullptr, *tmp2 = nullptr;
  size_t tmp_bytes = 0, tmp2_bytes = 0;
  int rc = PCM_OK;
  uint32_t hazards = 0;
  BuildScratch sc;
  // ivox3d.h:67  inv_resolution_ = 1.0 / resolution_ (float);  pcl::VoxelGrid: inverse_leaf_size_ = 1 / leaf_size_ in float
  const float inv_res = coord_mode == COORD_FLOOR_MUL ? 1.0f / res : (float)(1.0 / res);
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); rc = PCM_ERR_HIP; goto done; } \
  } while (0)
#define RC(x) do { rc = (x); if (rc != PCM_OK) goto done; } while (0)
  map->valid = false;
  if (!map->h_ctr && hipHostMalloc(reinterpret_cast<void**>(&map->h_ctr), kCtrInts * sizeof(int)) != hipSuccess) { *err = "hipHostMalloc(counters)"; return PCM_ERR_HIP; }
  if (incremental) {   // everything an update with eviction takes, rounded up: 24 B per new point, 52 B per log point, sort / scan temporaries
    const size_t need = 24 * (size_t)(n - n_indexed) + 56 * (size_t)n + ((size_t)16 << 20);
    if (need > map->arena_cap) {
      (void)hipStreamSynchronize(stream);
      if (map->arena) hipFree(map->arena);
      map->arena = nullptr; map->arena_cap = 0;
      const size_t nc = need + need / 4;
      CK(hipMalloc(reinterpret_cast<void**>(&map->arena), nc));
      map->arena_cap = nc;
    }
    sc.base = map->arena; sc.cap = map->arena_cap;
  }
  CK(sc.get(reinterpret_cast<void**>(&d_flags), kCtrInts * sizeof(int), stream));
  CK(hipMemsetAsync(d_flags, 0, kCtrInts * sizeof(int), stream));
  {
    // ---- 1. the sorted index (key, log position) of all n log points in map->keys_s / idx_s ------------------------------
    const uint32_t first = incremental ? n_indexed : 0u, m = n - first;
    RC(grow(&map->keys_s, &map->keys_cap, n, incremental ? n_indexed : 0, stream, err, "keys_s"));
    RC(grow(&map->idx_s, &map->idx_cap, n, incremental ? n_indexed : 0, stream, err, "idx_s"));
    RC(grow(&map->keys_t, &map->keys_t_cap, n, 0, stream, err, "keys_t"));
    RC(grow(&map->idx_t, &map->idx_t_cap, n, 0, stream, err, "idx_t"));
    if (m > 0) {
      CK(sc.get(reinterpret_cast<void**>(&keys_b), sizeof(uint64_t) * m, stream));
      CK(sc.get(reinterpret_cast<void**>(&keys_bs), sizeof(uint64_t) * m, stream));
      CK(sc.get(reinterpret_cast<void**>(&idx_b), sizeof(uint32_t) * m, stream));
      CK(sc.get(reinterpret_cast<void**>(&idx_bs), sizeof(uint32_t) * m, stream));
      k_point_keys_at<<<cdiv(m, 256), 256, 0, stream>>>(d_pts, first, m, res, inv_res, coord_mode, keys_b, idx_b, d_flags);
      CK(hipGetLastError());
      uint64_t* ks = incremental ? keys_bs : map->keys_s;   // a full build sorts straight into the persistent index
      uint32_t* is = incremental ? idx_bs : map->idx_s;
      CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_b, ks, idx_b, is, m, 0, 63, stream));
      CK(sc.get(reinterpret_cast<void**>(&tmp), tmp_bytes, stream));
      CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_b, ks, idx_b, is, m, 0, 63, stream));
      if (incremental) {
        k_merge_old<<<cdiv(n_indexed, 256), 256, 0, stream>>>(map->keys_s, map->idx_s, n_indexed, keys_bs, m, map->keys_t, map->idx_t);
        k_merge_new<<<cdiv(m, 256), 256, 0, stream>>>(map->keys_s, n_indexed, keys_bs, idx_bs, m, map->keys_t, map->idx_t);
        CK(hipGetLastError());
        std::swap(map->keys_s, map->keys_t); std::swap(map->keys_cap, map->keys_t_cap);
        std::swap(map->idx_s, map->idx_t); std::swap(map->idx_cap, map->idx_t_cap);
      }
    }
    map->index_n = 0;   // until this build is through
    // ---- 2. voxel heads, ranks; counts to the host -----------------------------------------------------------------------
    CK(sc.get(reinterpret_cast<void**>(&vflag), sizeof(uint32_t) * n, stream));
    CK(sc.get(reinterpret_cast<void**>(&vrank), sizeof(uint32_t) * n, stream));
    uint32_t nvox = 0, nbricks = 0;
    auto heads = [&](uint32_t cnt) -> int {
      CK(hipMemsetAsync(d_flags + 8, 0, kBrickCountShards * sizeof(int), stream));
      k_head_flags<<<cdiv(cnt, 256), 256, 0, stream>>>(map->keys_s, cnt, vflag, reinterpret_cast<unsigned int*>(d_flags + 8));
      CK(hipGetLastError());
      if (!tmp2) {
        CK(rocprim::exclusive_scan(nullptr, tmp2_bytes, vflag, vrank, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
        CK(sc.get(reinterpret_cast<void**>(&tmp2), tmp2_bytes, stream));
      }
      {
        size_t tb = tmp2_bytes;
        CK(rocprim::exclusive_scan(tmp2, tb, vflag, vrank, 0u, (size_t)cnt, rocprim::plus<uint32_t>(), stream));
      }
      k_count_voxels<<<1, 1, 0, stream>>>(vflag, vrank, cnt, d_flags);
      CK(hipMemcpyAsync(map->h_ctr, d_flags, kCtrInts * sizeof(int), hipMemcpyDeviceToHost, stream));   // one copy into pinned memory
      CK(hipStreamSynchronize(stream));
      if (map->h_ctr[0]) { *err = "target point outside the +-2^20 voxel range (or not finite)"; rc = PCM_ERR_OUT_OF_RANGE; goto done; }
      nvox = (uint32_t)map->h_ctr[4];
      nbricks = 0;
      for (int k = 0; k < kBrickCountShards; k++) nbricks += (uint32_t)map->h_ctr[8 + k];
      return PCM_OK;
    done:
      return rc;
    };
    RC(heads(n));
    if (subsort && !incremental) {   // permute the index inside the voxels (the heads and ranks above stay valid: same keys at the same places)
      uint64_t *k2 = nullptr, *k2s = nullptr;
      void* tmp5 = nullptr;
      size_t tmp5_bytes = 0;
      int bits = 9;
      while (bits < 41 && (1ull << (bits - 9)) < (unsigned long long)nvox) bits++;
      CK(sc.get(reinterpret_cast<void**>(&k2), sizeof(uint64_t) * n, stream));
      CK(sc.get(reinterpret_cast<void**>(&k2s), sizeof(uint64_t) * n, stream));
      k_subvoxel_keys<<<cdiv(n, 256), 256, 0, stream>>>(d_pts, map->idx_s, vflag, vrank, n, inv_res, k2);
      CK(hipGetLastError());
      CK(rocprim::radix_sort_pairs(nullptr, tmp5_bytes, k2, k2s, map->idx_s, map->idx_t, n, 0, bits, stream));
      CK(sc.get(&tmp5, tmp5_bytes, stream));
      CK(rocprim::radix_sort_pairs(tmp5, tmp5_bytes, k2, k2s, map->idx_s, map->idx_t, n, 0, bits, stream));
      std::swap(map->idx_s, map->idx_t); std::swap(map->idx_cap, map->idx_t_cap);
    }
    if (capacity_voxels > 1 && nvox > capacity_voxels - 1) {
      // ---- 3. LRU eviction: keep the (capacity - 1) most recently touched voxels; log and index are compacted in place ------
      const uint32_t keep = capacity_voxels - 1;
      uint32_t *vlast = nullptr, *vsorted = nullptr, *alive = nullptr, *pos = nullptr, *alive_s = nullptr, *pos_s = nullptr, *vfirst = nullptr;
      float4* tmp_log = nullptr;
      void *tmp3 = nullptr, *tmp4 = nullptr;
      size_t tmp3_bytes = 0, tmp4_bytes = 0;
      const uint32_t* d_cutoff = nullptr;
      int rc2 = PCM_OK;
#define CK2(x)                                                                   \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); rc2 = PCM_ERR_HIP; goto evict_done; } \
  } while (0)
      CK2(sc.get(reinterpret_cast<void**>(&vlast), sizeof(uint32_t) * nvox, stream));
      CK2(sc.get(reinterpret_cast<void**>(&vsorted), sizeof(uint32_t) * nvox, stream));
      CK2(sc.get(reinterpret_cast<void**>(&vfirst), sizeof(uint32_t) * nvox, stream));
      CK2(sc.get(reinterpret_cast<void**>(&alive), sizeof(uint32_t) * n, stream));
      CK2(sc.get(reinterpret_cast<void**>(&pos), sizeof(uint32_t) * n, stream));
      CK2(sc.get(reinterpret_cast<void**>(&alive_s), sizeof(uint32_t) * n, stream));
      CK2(sc.get(reinterpret_cast<void**>(&pos_s), sizeof(uint32_t) * n, stream));
      CK2(sc.get(reinterpret_cast<void**>(&tmp_log), sizeof(float4) * n, stream));
      k_voxel_firsts<<<cdiv(n, 256), 256, 0, stream>>>(vflag, vrank, n, vfirst);
      k_voxel_last_from_firsts<<<cdiv(nvox, 256), 256, 0, stream>>>(d_pts, map->idx_s, vfirst, nvox, n, vlast);
      CK2(hipGetLastError());
      CK2(rocprim::radix_sort_keys(nullptr, tmp3_bytes, vlast, vsorted, nvox, 0, 32, stream));
      CK2(sc.get(reinterpret_cast<void**>(&tmp3), tmp3_bytes, stream));
      CK2(rocprim::radix_sort_keys(tmp3, tmp3_bytes, vlast, vsorted, nvox, 0, 32, stream));
      d_cutoff = vsorted + (nvox - keep);   // the smallest surviving last-touch stamp, read on the device
      if (incremental && n_indexed < n) {   // the batch = the log points from n_indexed on
        k_lru_hazards<<<cdiv(nvox, 256), 256, 0, stream>>>(d_pts, map->idx_s, vfirst, vlast, nvox, n, d_pts + n_indexed, d_cutoff, reinterpret_cast<unsigned int*>(d_flags + 3));
        CK2(hipGetLastError());
      }
      k_mark_alive_dev<<<cdiv(n, 256), 256, 0, stream>>>(map->idx_s, vflag, vrank, vlast, d_cutoff, n, alive);
      CK2(hipGetLastError());
      CK2(rocprim::exclusive_scan(nullptr, tmp4_bytes, alive, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
      CK2(sc.get(reinterpret_cast<void**>(&tmp4), tmp4_bytes, stream));
      { size_t tb = tmp4_bytes; CK2(rocprim::exclusive_scan(tmp4, tb, alive, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream)); }
      k_alive_sorted<<<cdiv(n, 256), 256, 0, stream>>>(map->idx_s, alive, n, alive_s);
      { size_t tb = tmp4_bytes; CK2(rocprim::exclusive_scan(tmp4, tb, alive_s, pos_s, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream)); }
      k_compact_log<<<cdiv(n, 256), 256, 0, stream>>>(d_pts, alive, pos, n, tmp_log);
      k_compact_index<<<cdiv(n, 256), 256, 0, stream>>>(map->keys_s, map->idx_s, alive, pos, pos_s, n, map->keys_t, map->idx_t);
      k_count_alive<<<1, 1, 0, stream>>>(alive, pos, n, d_flags);
      CK2(hipGetLastError());
      CK2(hipMemcpyAsync(map->h_ctr, d_flags, kCtrInts * sizeof(int), hipMemcpyDeviceToHost, stream));
      CK2(hipStreamSynchronize(stream));
      n = (uint32_t)map->h_ctr[5];
      hazards = (uint32_t)map->h_ctr[3];
      CK2(hipMemcpyAsync(d_pts, tmp_log, sizeof(float4) * (size_t)n, hipMemcpyDeviceToDevice, stream));
      std::swap(map->keys_s, map->keys_t); std::swap(map->keys_cap, map->keys_t_cap);
      std::swap(map->idx_s, map->idx_t); std::swap(map->idx_cap, map->idx_t_cap);
      *n_inout = n;
    evict_done:
#undef CK2
      if (rc2 != PCM_OK) { rc = rc2; goto done; }
      RC(heads(n));   // the survivors' voxel heads and ranks
    }
    // ---- 4. tables from the sorted index (capacity-managed arrays: a sliding map allocates nothing in the steady state) ----
    uint32_t cap = 1024;
    while (cap < 4ull * nbricks) cap <<= 1;
    if (cap > kMaxBrickSlots) { *err = "too many occupied bricks"; rc = PCM_ERR_OUT_OF_RANGE; goto done; }
    if (cap > map->bricks_cap) {   // the table only grows; a smaller table uses the front of the allocation
      (void)hipStreamSynchronize(stream);
      if (map->bricks) hipFree(map->bricks);
      if (map->bmask) hipFree(map->bmask);
      if (map->bpref) hipFree(map->bpref);
      map->bricks = nullptr; map->bmask = nullptr; map->bpref = nullptr; map->bricks_cap = 0;
      CK(hipMalloc(&map->bricks, sizeof(BrickSlot) * (size_t)cap));
      CK(hipMalloc(&map->bmask, sizeof(uint32_t) * 16 * (size_t)cap));
      CK(hipMalloc(&map->bpref, sizeof(uint16_t) * 16 * (size_t)cap));
      map->bricks_cap = cap;
    }
    RC(grow(&map->vox_start, &map->vox_cap, (size_t)nvox + 1, 0, stream, err, "vox_start"));
    RC(grow(&map->pts, &map->pts_cap, (size_t)n, 0, stream, err, "pts"));
    CK(hipMemsetAsync(map->bricks, 0xFF, sizeof(BrickSlot) * (size_t)cap, stream));
    CK(hipMemsetAsync(map->bmask, 0, sizeof(uint32_t) * 16 * (size_t)cap, stream));
    CK(hipMemsetAsync(map->bpref, 0, sizeof(uint16_t) * 16 * (size_t)cap, stream));
    k_insert_bricks<<<cdiv(n, 256), 256, 0, stream>>>(map->keys_s, vrank, n, map->bricks, cap - 1);
    CK(hipGetLastError());
    k_fill_voxels<<<cdiv(n, 256), 256, 0, stream>>>(map->keys_s, vflag, vrank, n, nvox, map->vox_start, map->bricks, map->bmask, cap - 1);
    CK(hipGetLastError());
    k_finalize_bricks<<<cdiv(cap, 256), 256, 0, stream>>>(map->bricks, map->bmask, map->bpref, map->vox_start, cap);
    CK(hipGetLastError());
    k_gather_points<<<cdiv(n, 256), 256, 0, stream>>>(d_pts, map->idx_s, map->keys_s, vrank, map->vox_start, n, map->pts);
    CK(hipGetLastError());
    k_max_voxel_points<<<cdiv(nvox, 1024), 256, 0, stream>>>(map->vox_start, nvox, reinterpret_cast<unsigned int*>(d_flags + 2));
    CK(hipGetLastError());
    CK(hipMemcpyAsync(map->h_ctr, d_flags, kCtrInts * sizeof(int), hipMemcpyDeviceToHost, stream));   // complete at the synchronize below
    if (map->gvox) { (void)hipStreamSynchronize(stream); hipFree(map->gvox); map->gvox = nullptr; }
    if (map->order && !keep_order) { (void)hipStreamSynchronize(stream); hipFree(map->order); map->order = nullptr; map->order_cap = 0; }
    if (keep_order) {   // a persistent copy of the input index of every map point (capacity-managed: a scan per registration re-uses it)
      RC(grow(&map->order, &map->order_cap, (size_t)n, 0, stream, err, "order"));
      CK(hipMemcpyAsync(map->order, map->idx_s, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    }
    if (want_gauss) {
      CK(hipMalloc(&map->gvox, sizeof(GaussVoxel) * ((size_t)nvox + 1)));
      k_gauss_voxels<<<cdiv(nvox, 128), 128, 0, stream>>>(map->pts, map->vox_start, nvox, map->gvox);
      CK(hipGetLastError());
    }
    CK(hipStreamSynchronize(stream));
    map->max_voxel_points = (uint32_t)map->h_ctr[2];
    map->cap = cap;
    map->num_voxels = nvox;
    map->num_bricks = nbricks;
    map->num_points = n;
    map->res = res;
    map->inv_res = inv_res;
    map->coord_mode = coord_mode;
    map->index_n = n;
    map->valid = true;
  }
done:
  sc.release(stream);
  if (lru_hazards) *lru_hazards = hazards;
  if (rc != PCM_OK) map->release();
  return rc;
#undef CK
#undef RC
}

// Re-order the scans of a whole batch along a Morton (Z-order) curve of their WORLD
// voxel coordinates at the initial guess, so that the 256 points of a
// k_linearize tile fall into a few neighbouring voxels of the submap grid
// (their voxel box then fits the LDS grid).  One key kernel + one radix sort +
// one gather for all pairs of the batch.  Speed only: the normal equations are a
// sum over points, so the order never changes a result beyond floating-point
// summation order.
__device__ inline uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// grid = (ceil(max_n / 256), npairs)
__global__ void k_batch_morton_keys(const SortJob* __restrict__ jobs, const float* __restrict__ guesses, float inv_res, uint64_t* __restrict__ keys,
                                    uint32_t* __restrict__ vals) {
  const SortJob job = jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= job.n) return;
  const float* T = guesses + 16 * job.guess_index;
  const float4 p = gload4(job.src + i);
  float c[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float q = (T[a * 4 + 0] * p.x + T[a * 4 + 1] * p.y) + T[a * 4 + 2] * p.z + T[a * 4 + 3];
    // cell index relative to the sensor's cell, clamped to +-512 cells (beyond that only locality is lost)
    const float r = roundf(q * inv_res) - roundf(T[a * 4 + 3] * inv_res);
    c[a] = r == r ? fminf(fmaxf(r, -512.f), 511.f) : 0.f;
  }
  const uint32_t m = spread10((uint32_t)((int)c[0] + 512)) | (spread10((uint32_t)((int)c[1] + 512)) << 1) | (spread10((uint32_t)((int)c[2] + 512)) << 2);
  keys[job.offset + i] = ((uint64_t)blockIdx.y << 32) | m;
  vals[job.offset + i] = job.offset + i;
}

// grid = ceil(total / 256)
__global__ void k_batch_gather(const SortJob* __restrict__ jobs, const uint64_t* __restrict__ keys_sorted, const uint32_t* __restrict__ vals_sorted, uint32_t total) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const SortJob job = jobs[(uint32_t)(keys_sorted[g] >> 32)];
  const uint32_t from = vals_sorted[g] - job.offset;
  float4 v = gload4(job.src + from);
  v.w = __uint_as_float(from);   // position in the caller's scan: per-point state that outlives a frame is kept in that order
  gstore4(job.dst + (g - job.offset), v);
}

// The same with 32-bit keys, for batches of up to 256 scans: [scan | Morton code] in one word -- four 8-bit radix passes over 8 bytes
// per point instead of five over 12 (the re-ordering is timed with every registration, and with the candidate-list kernel it was a
// fifth of a pass's device time).  The Morton code keeps its cell size and gives up RANGE instead: `mb` = 32 - scan bits are split
// over the axes (z takes the floor of a third, x and y the rest), coordinates beyond an axis' range clamp (only locality is lost).
// grid = (ceil(max_n / 256), npairs)
__global__ void k_batch_morton_keys32(const SortJob* __restrict__ jobs, const float* __restrict__ guesses, float inv_res, int mb, uint32_t* __restrict__ keys,
                                      uint32_t* __restrict__ vals) {
  const SortJob job = jobs[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= job.n) return;
  const float* T = guesses + 16 * job.guess_index;
  const float4 p = gload4(job.src + i);
  const int bz = mb / 3, rem = mb - 3 * bz;
  const int nb[3] = {bz + (rem >= 1 ? 1 : 0), bz + (rem >= 2 ? 1 : 0), bz};   // bits of x, y, z
  uint32_t c[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float q = (T[a * 4 + 0] * p.x + T[a * 4 + 1] * p.y) + T[a * 4 + 2] * p.z + T[a * 4 + 3];
    const float half = (float)(1 << (nb[a] - 1));
    const float r = roundf(q * inv_res) - roundf(T[a * 4 + 3] * inv_res);   // cell index relative to the sensor's cell
    c[a] = (uint32_t)((int)(r == r ? fminf(fmaxf(r, -half), half - 1.f) : 0.f) + (1 << (nb[a] - 1)));
  }
  const uint32_t lowmask = (1u << bz) - 1u;
  uint32_t m = spread10(c[0] & lowmask) | (spread10(c[1] & lowmask) << 1) | (spread10(c[2] & lowmask) << 2);   // bz <= 10
  m |= ((c[0] >> bz) | ((c[1] >> bz) << (nb[0] - bz))) << (3 * bz);                                             // the extra top bits of x and y
  keys[job.offset + i] = (blockIdx.y << mb) | m;
  vals[job.offset + i] = job.offset + i;
}

// grid = ceil(total / 256)
__global__ void k_batch_gather32(const SortJob* __restrict__ jobs, const uint32_t* __restrict__ keys_sorted, const uint32_t* __restrict__ vals_sorted, uint32_t total, int mb) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const SortJob job = jobs[keys_sorted[g] >> mb];
  const uint32_t from = vals_sorted[g] - job.offset;
  float4 v = gload4(job.src + from);
  v.w = __uint_as_float(from);   // position in the caller's scan: per-point state that outlives a frame is kept in that order
  gstore4(job.dst + (g - job.offset), v);
}

int sort_sources_batched(hipStream_t stream, const SortJob* d_jobs, int njobs, uint32_t max_n, uint32_t total, const float* d_guesses, float res,
                         SortScratch* ws, std::string* err) {
  if (total == 0 || njobs == 0) return PCM_OK;
  int rc = PCM_OK;
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); rc = PCM_ERR_HIP; goto done; } \
  } while (0)
  {
    int pair_bits = 1;
    while ((1 << pair_bits) < njobs) pair_bits++;
    size_t need = 0;
    if (total > ws->cap) {
      hipFree(ws->keys); hipFree(ws->vals); hipFree(ws->tmp);
      ws->keys = nullptr; ws->vals = nullptr; ws->tmp = nullptr; ws->cap = 0; ws->tmp_bytes = 0;
      CK(hipMalloc(&ws->keys, sizeof(uint64_t) * 2 * (size_t)total));
      CK(hipMalloc(&ws->vals, sizeof(uint32_t) * 2 * (size_t)total));
      ws->cap = total;
    }
    uint64_t* keys_s = ws->keys + ws->cap;
    uint32_t* vals_s = ws->vals + ws->cap;
    if (pair_bits <= 8) {   // one 32-bit word per point (the key buffers are re-used as 32-bit arrays)
      const int mb = 32 - pair_bits;
      uint32_t* k32 = reinterpret_cast<uint32_t*>(ws->keys);
      uint32_t* k32_s = k32 + ws->cap;
      CK(rocprim::radix_sort_pairs(nullptr, need, k32, k32_s, ws->vals, vals_s, total, 0, 32, stream));
      if (need > ws->tmp_bytes) {
        hipFree(ws->tmp);
        ws->tmp = nullptr; ws->tmp_bytes = 0;
        CK(hipMalloc(&ws->tmp, need));
        ws->tmp_bytes = need;
      }
      k_batch_morton_keys32<<<dim3(cdiv(max_n, 256), (unsigned)njobs), 256, 0, stream>>>(d_jobs, d_guesses, (float)(1.0 / res), mb, k32, ws->vals);
      CK(hipGetLastError());
      need = ws->tmp_bytes;
      CK(rocprim::radix_sort_pairs(ws->tmp, need, k32, k32_s, ws->vals, vals_s, total, 0, 32, stream));
      k_batch_gather32<<<cdiv(total, 256), 256, 0, stream>>>(d_jobs, k32_s, vals_s, total, mb);
      CK(hipGetLastError());
      goto done;
    }
    CK(rocprim::radix_sort_pairs(nullptr, need, ws->keys, keys_s, ws->vals, vals_s, total, 0, 32 + pair_bits, stream));
    if (need > ws->tmp_bytes) {
      hipFree(ws->tmp);
      ws->tmp = nullptr; ws->tmp_bytes = 0;
      CK(hipMalloc(&ws->tmp, need));
      ws->tmp_bytes = need;
    }
    k_batch_morton_keys<<<dim3(cdiv(max_n, 256), (unsigned)njobs), 256, 0, stream>>>(d_jobs, d_guesses, (float)(1.0 / res), ws->keys, ws->vals);
    CK(hipGetLastError());
    need = ws->tmp_bytes;
    CK(rocprim::radix_sort_pairs(ws->tmp, need, ws->keys, keys_s, ws->vals, vals_s, total, 0, 32 + pair_bits, stream));
    k_batch_gather<<<cdiv(total, 256), 256, 0, stream>>>(d_jobs, keys_s, vals_s, total);
    CK(hipGetLastError());
  }
done:
  return rc;
#undef CK
}

// ---------------------------------------------------------------------------
// LaserMapping::MapIncremental (/root/reference/src/jueying_lio/src/laser_mapping.cc:525-583)
// ---------------------------------------------------------------------------
__device__ inline void qrot_d(const double* q, const double* v, double* r) {   // Eigen _transformVector, (x,y,z,w)
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int a = 0; a < 3; a++) r[a] = v[a] + q[3] * uv[a] + c[a];
}

// flag[i]: 0 = not added, 1 = points_to_add, 2 = point_no_need_downsample; world[i] = PointBodyToWorld(scan[i])
__global__ void k_map_filter(const float4* __restrict__ scan, uint32_t n, LioStateD s, float fs, const uint32_t* __restrict__ nn, const float4* __restrict__ map_pts,
                             float4* __restrict__ world, uint32_t* __restrict__ f1, uint32_t* __restrict__ f2, int scan_reordered) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 pb = scan[i];
  // the verdicts are laid out in the order of the caller's scan (a re-ordered scan carries that position in w): the order in
  // which points enter the map decides their voxels' places in the LRU list
  const uint32_t o = scan_reordered ? __float_as_uint(pb.w) : i;
  // p_global = rot * (offset_R_L_I * p_body + offset_T_L_I) + pos   in double   laser_mapping.cc:855-864
  const double vb[3] = {(double)pb.x, (double)pb.y, (double)pb.z};
  double v1[3], v2[3];
  qrot_d(s.off_R, vb, v1);
  for (int a = 0; a < 3; a++) v1[a] += s.off_T[a];
  qrot_d(s.rot, v1, v2);
  const float pw[3] = {(float)(v2[0] + s.pos[0]), (float)(v2[1] + s.pos[1]), (float)(v2[2] + s.pos[2])};
  world[o] = make_float4(pw[0], pw[1], pw[2], 0.f);
  uint32_t flag = 1;
  if (nn != nullptr && nn[(size_t)i * 5] != ~0u) {   // !nearest_points_[i].empty() && flg_EKF_inited_
    float center[3];
#pragma unroll
    for (int a = 0; a < 3; a++) center[a] = (floorf(pw[a] / fs) + 0.5f) * fs;   // :547-548
    const float4 n0 = map_pts[nn[(size_t)i * 5]];
    const double half = 0.5 * (double)fs;
    if ((double)fabsf(n0.x - center[0]) > half && (double)fabsf(n0.y - center[1]) > half && (double)fabsf(n0.z - center[2]) > half) {
      flag = 2;                                                                  // :552-557
    } else {
      const float dx = pw[0] - center[0], dy = pw[1] - center[1], dz = pw[2] - center[2];
      const float dist = dx * dx + dy * dy + dz * dz;
      bool need_add = true;
      if (nn[(size_t)i * 5 + 4] != ~0u) {                                       // points_near.size() >= NUM_MATCH_POINTS
        for (int k = 0; k < 5; k++) {
          const float4 q = map_pts[nn[(size_t)i * 5 + k]];
          const float ex = q.x - center[0], ey = q.y - center[1], ez = q.z - center[2];
          if ((double)(ex * ex + ey * ey + ez * ez) < (double)dist + 1e-6) { need_add = false; break; }   // :563-566
        }
      }
      flag = need_add ? 1u : 0u;
    }
  }
  f1[o] = flag == 1 ? 1u : 0u;
  f2[o] = flag == 2 ? 1u : 0u;
}

__global__ void k_map_append(const float4* __restrict__ world, const uint32_t* __restrict__ f1, const uint32_t* __restrict__ f2, const uint32_t* __restrict__ p1,
                             const uint32_t* __restrict__ p2, uint32_t n, uint32_t n1, uint32_t seq0, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (f1[i]) { float4 v = world[i]; v.w = __uint_as_float(seq0 + p1[i]); out[p1[i]] = v; }                 // ivox_->AddPoints(points_to_add)
  if (f2[i]) { float4 v = world[i]; v.w = __uint_as_float(seq0 + n1 + p2[i]); out[n1 + p2[i]] = v; }       // then point_no_need_downsample
}

int map_incremental_device(hipStream_t stream, const float4* scan, bool scan_reordered, uint32_t n, const LioStateD& s, float filter_size_map, const uint32_t* nn,
                           const float4* map_pts, uint32_t seq0, float4* out_append, uint32_t* num_added, std::string* err) {
  float4* world = nullptr;
  uint32_t* buf = nullptr;   // f1, f2, p1, p2
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = PCM_OK;
  uint32_t tails[4] = {0, 0, 0, 0};
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); rc = PCM_ERR_HIP; goto done; } \
  } while (0)
  CK(hipMallocAsync(&world, sizeof(float4) * n, stream));
  CK(hipMallocAsync(&buf, sizeof(uint32_t) * 4 * (size_t)n, stream));
  {
    uint32_t *f1 = buf, *f2 = buf + n, *p1 = buf + 2 * (size_t)n, *p2 = buf + 3 * (size_t)n;
    k_map_filter<<<cdiv(n, 256), 256, 0, stream>>>(scan, n, s, filter_size_map, nn, map_pts, world, f1, f2, scan_reordered ? 1 : 0);
    CK(hipGetLastError());
    CK(rocprim::exclusive_scan(nullptr, tmp_bytes, f1, p1, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    CK(hipMallocAsync(&tmp, tmp_bytes, stream));
    CK(rocprim::exclusive_scan(tmp, tmp_bytes, f1, p1, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    CK(rocprim::exclusive_scan(tmp, tmp_bytes, f2, p2, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    CK(hipMemcpyAsync(&tails[0], f1 + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(&tails[1], p1 + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(&tails[2], f2 + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(&tails[3], p2 + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    const uint32_t n1 = tails[0] + tails[1], n2 = tails[2] + tails[3];
    k_map_append<<<cdiv(n, 256), 256, 0, stream>>>(world, f1, f2, p1, p2, n, n1, seq0, out_append);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(stream));
    *num_added = n1 + n2;
  }
done:
  sfree(stream, world); sfree(stream, buf); sfree(stream, tmp);
  return rc;
#undef CK
}

int load_points_to_device(hipStream_t stream, const void* points, size_t n, size_t stride, int memory, uint32_t seq0, float4* d_out, std::string* err) {
  if (n == 0) return PCM_OK;
  hipError_t e;
  if (memory == PCM_MEM_DEVICE) {
    k_load_points<<<cdiv((uint32_t)n, 256), 256, 0, stream>>>(static_cast<const char*>(points), stride, (uint32_t)n, seq0, d_out);
    e = hipGetLastError();
    if (e != hipSuccess) { *err = std::string("k_load_points: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
    return PCM_OK;
  }
  // host records: copy the 12-byte xyz prefix of every record into 16-byte rows, then normalise w
  e = hipMemcpy2DAsync(d_out, sizeof(float4), points, stride, 3 * sizeof(float), n, hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) { *err = std::string("hipMemcpy2DAsync: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  k_load_points<<<cdiv((uint32_t)n, 256), 256, 0, stream>>>(reinterpret_cast<const char*>(d_out), sizeof(float4), (uint32_t)n, seq0, d_out);
  e = hipGetLastError();
  if (e != hipSuccess) { *err = std::string("k_load_points: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

}  // namespace pcm

// kernels.hip -- the per-round kernels of the scan-to-submap registration path (gfx950):
//
//   k_corr_search      correspondence search: 5-NN of every scan point in the
//                      linear-probed voxel hash of the submap + plane fit
//   k_residual_reduce  point-to-plane residual / Jacobian + 6x6 normal-equation
//                      accumulation (wave64 reduction, deterministic partials)
//   k_lsq_step         fixed-order sum of the partials + GN / LM state machine
//
// Replaces, for the MI355X path (paths relative to /root/reference/src):
//   - LaserMapping::ObsModel matcher loop            jueying_lio/src/laser_mapping.cc:606-637
//   - IVox::GetClosestPoint / KNNPointByCondition    jueying_lio/include/ivox3d/ivox3d.h:132-204, ivox3d_node.hpp:140-205
//   - common::esti_plane                             jueying_lio/include/common_lib.h:186-243
//   - HTH = h_x^T h_x  ("J^T J")                      jueying_lio/include/IKFoM_toolkit/esekfom/esekfom.hpp:1687
//   - find_voxel_correspondences + compute_derivatives + transform_reduce of the
//     reference's CUDA path (materialised pair list, 43-float tuple reduction)
//                                                    pointcloud_match/fast_gicp/src/fast_gicp/cuda/{find_voxel_correspondences,compute_derivatives}.cu
//   - LsqRegistration step_gn / step_lm              pointcloud_match/fast_gicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp:105-172
//
// Shape (not a port).  The scan is ordered along a Morton curve once per scan,
// so the 256 points of a workgroup tile sit in a few neighbouring voxels.  The
// tile's voxel bounding box (+1 halo) is resolved against the global hash ONCE
// per tile -- one probe per cell instead of 27 per point -- into a dense LDS
// grid, and the map points of those cells are staged through LDS with coalesced
// 16-byte loads; each lane then runs its 27-cell / 5-NN search and the plane fit
// entirely out of LDS and registers.  Tiles whose box does not fit fall back to
// per-lane probing of the global table with the first probes of 9 cells in
// flight at once.  No correspondence list is written to HBM: the only per-point
// output is the fitted plane (one float4), which the second kernel turns into
// the 28 unique normal-equation terms (float geometry, double accumulation,
// wave64 cross-lane reduction, one partial row per workgroup).  The GN/LM state
// stays on the device for the whole align().
//
// Compiled with -ffp-contract=off: the float geometry that feeds discrete
// decisions (voxel key, kNN order, plane test) must round like the reference's
// plain x86 build; the double accumulations use explicit fma().
#include "pcm_device.h"
#include "pcm_host.h"
#include "plane_fit.h"

namespace pcm {

// neighbour cells in the reference's order (ivox3d.h:211-235): CENTER, NEARBY6, NEARBY18, NEARBY26 are prefixes
__constant__ int8_t c_nearby[27][4] = {
  {0, 0, 0, 0},   {-1, 0, 0, 0},  {1, 0, 0, 0},   {0, 1, 0, 0},   {0, -1, 0, 0},  {0, 0, -1, 0},  {0, 0, 1, 0},
  {1, 1, 0, 0},   {-1, 1, 0, 0},  {1, -1, 0, 0},  {-1, -1, 0, 0}, {1, 0, 1, 0},   {-1, 0, 1, 0},  {1, 0, -1, 0},
  {-1, 0, -1, 0}, {0, 1, 1, 0},   {0, -1, 1, 0},  {0, 1, -1, 0},  {0, -1, -1, 0}, {1, 1, 1, 0},   {-1, 1, 1, 0},
  {1, -1, 1, 0},  {1, 1, -1, 0},  {-1, -1, 1, 0}, {-1, 1, -1, 0}, {1, -1, -1, 0}, {-1, -1, -1, 0}};

struct PoseF {
  float r[9];
  float t[3];
};

__device__ inline PoseF load_pose(const double* T) {
  PoseF p;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) p.r[i * 3 + j] = (float)T[i * 4 + j];  // trans.cast<float>()  laser_mapping.cc:602-603
    p.t[i] = (float)T[i * 4 + 3];
  }
  return p;
}

__device__ inline void transform(const PoseF& P, const float4& p, float (&q)[3]) {
#pragma unroll
  for (int a = 0; a < 3; a++) q[a] = (P.r[a * 3 + 0] * p.x + P.r[a * 3 + 1] * p.y) + P.r[a * 3 + 2] * p.z + P.t[a];
}

// running 5-best list: ascending distance, equal distances keep visit order
struct Best {
  float d[K];
  uint32_t i[K];
  int m;
};

__device__ inline void best_init(Best& b) {
#pragma unroll
  for (int j = 0; j < K; j++) { b.d[j] = __builtin_inff(); b.i[j] = 0xffffffffu; }
  b.m = 0;
}

// Offer one map point.  `max_r2f` is the smallest float >= max_range^2, so
// `d2 < max_r2f` is exactly the reference's `double(d2) < max_range * max_range`
// (ivox3d_node.hpp:162) without a double-precision compare per candidate.
// Sorted insert with strict '<' (an equal distance goes behind the entries
// already there = visit order); new d[j] = median(d2, d[j-1], d[j]).
__device__ inline void best_offer(Best& b, const float4& mp, const float (&q)[3], uint32_t id, float max_r2f) {
  const float dx = mp.x - q[0], dy = mp.y - q[1], dz = mp.z - q[2];
  const float d2 = dx * dx + dy * dy + dz * dz;  // distance2()  ivox3d_node.hpp:13-16
  if (d2 < max_r2f) {
    b.m = b.m < K ? b.m + 1 : K;
    if (d2 < b.d[4]) {
      const bool c0 = d2 < b.d[0], c1 = d2 < b.d[1], c2 = d2 < b.d[2], c3 = d2 < b.d[3];
      b.i[4] = c3 ? b.i[3] : id;
      b.d[4] = fmaxf(d2, b.d[3]);
      b.i[3] = c2 ? b.i[2] : (c3 ? id : b.i[3]);
      b.d[3] = __builtin_amdgcn_fmed3f(d2, b.d[2], b.d[3]);
      b.i[2] = c1 ? b.i[1] : (c2 ? id : b.i[2]);
      b.d[2] = __builtin_amdgcn_fmed3f(d2, b.d[1], b.d[2]);
      b.i[1] = c0 ? b.i[0] : (c1 ? id : b.i[1]);
      b.d[1] = __builtin_amdgcn_fmed3f(d2, b.d[0], b.d[1]);
      b.i[0] = c0 ? id : b.i[0];
      b.d[0] = fminf(d2, b.d[0]);
    }
  }
}

__device__ inline uint64_t slot_key(const uint4& s) { return ((uint64_t)s.y << 32) | s.x; }

// find a brick in the linear-probed brick table: slot index (or ~0u) and its first voxel
template <bool STATS>
__device__ inline uint32_t brick_find(const TargetView& tg, int bx, int by, int bz, uint32_t& vox_base, uint32_t& n_probe) {
  const uint64_t key = pack_brick(bx, by, bz);
  uint32_t h = hash_coord(bx, by, bz) & tg.mask;
  for (;;) {
    const uint4 s = gload4u(&tg.bricks[h]);
    if (STATS) n_probe++;
    const uint64_t sk = slot_key(s);
    if (sk == key) { vox_base = s.z; return h; }
    if (sk == kEmptyKey) { vox_base = 0; return ~0u; }
    h = (h + 1) & tg.mask;
  }
}

// Per-lane search straight against the global structures (tiles whose voxel box
// does not fit the LDS grid): brick probe (re-used while consecutive cells stay in
// one brick) -> occupancy bit -> rank -> vox_start -> the voxel's points.
template <bool STATS>
__device__ inline void knn_global(const TargetView& tg, const float (&q)[3], int cx, int cy, int cz, int num_neighbors, float max_range_sq, Best& best,
                                  uint32_t& n_cand, uint32_t& n_probe) {
  int cbx = 0x7fffffff, cby = 0, cbz = 0;
  uint32_t slot = ~0u, vox_base = 0;
  for (int g = 0; g < num_neighbors; g++) {
    const int vx = cx + c_nearby[g][0], vy = cy + c_nearby[g][1], vz = cz + c_nearby[g][2];
    const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
    if (bx != cbx || by != cby || bz != cbz) {
      slot = brick_find<STATS>(tg, bx, by, bz, vox_base, n_probe);
      cbx = bx; cby = by; cbz = bz;
    }
    if (slot == ~0u) continue;
    const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
    const uint32_t m = gload_u(&tg.bmask[(size_t)slot * 16 + w]);
    if (!((m >> bit) & 1u)) continue;
    const uint32_t v = vox_base + gload_u16(&tg.bpref[(size_t)slot * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u));
    const uint32_t start = gload_u(&tg.vox_start[v]), end = gload_u(&tg.vox_start[v + 1]);
    for (uint32_t k = start; k < end; k++) {
      const float4 mp = gload4(tg.pts + k);
      if (STATS) n_cand++;
      best_offer(best, mp, q, k, max_range_sq);
    }
  }
}

// ---------------------------------------------------------------------------
// k_corr_search: grid = (tiles_per_pair, npairs), block = 256, one scan point per lane
// ---------------------------------------------------------------------------
constexpr int kCapCells = 2048;   // LDS voxel grid of a tile (one uint16 per cell)
constexpr int kCapPts = 1792;     // map points staged per tile (float4 each); keeps the workgroup under 40 KB of LDS (4 per CU)
constexpr int kCapBricks = 64;    // bricks under a tile box
constexpr uint16_t kNoCell = 0xffffu;

// TIMING (diagnostic build only): lane 0 of every tile stamps s_memtime at the phase
// boundaries and adds the differences to stats[8..15]; nothing is computed from them.
#define PCM_STAMP(slot)                                                        \
  if (TIMING) {                                                                \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();             \
    if (threadIdx.x == 0) atomicAdd(&stats[8 + (slot)], t_now - t_prev);       \
    t_prev = t_now;                                                            \
  }

template <bool STATS, bool TIMING>
__global__ void __launch_bounds__(256) k_corr_search(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp,
                                                     unsigned long long* __restrict__ stats) {
  const int pair = blockIdx.y;
  if (states[pair].mode != MODE_LINEARIZE) return;
  const PairDesc d = descs[pair];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x * 256u >= d.src.num_points) return;
  const bool live = i < d.src.num_points;
  const PoseF P = load_pose(states[pair].x0);
  const TargetView tg = d.tgt;

  __shared__ int s_red[4][6];
  __shared__ int s_box[8];                 // origin xyz, dims xyz, ncell, dense flag
  __shared__ int s_bbox[8];                // first brick xyz, brick dims xyz, nbricks
  __shared__ uint32_t s_bps[kCapBricks];   // first map point of each brick under the box
  __shared__ uint32_t s_boff[kCapBricks + 1];  // its offset in s_pts (exclusive scan of the point counts)
  __shared__ uint32_t s_njobs;
  __shared__ uint16_t s_cell[kCapCells];   // first staged point of the voxel in that cell, kNoCell when empty
  __shared__ float4 s_pts[kCapPts];        // the bricks' map points, .w = voxel tag
  __shared__ uint32_t s_job[256];          // owner tid | m << 16
  __shared__ uint32_t s_jobid[256][4];     // the 3 or 4 neighbour ids of the job

  uint32_t n_cand = 0, n_probe = 0;
  unsigned long long t_prev = 0;
  if (TIMING) t_prev = __builtin_amdgcn_s_memtime();
  float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
  float q[3] = {0.f, 0.f, 0.f};
  int cx = 0, cy = 0, cz = 0;
  bool search = false;  // lanes whose query voxel lies inside the key range of the table
  if (live) {
    p = gload4(d.src.pts + i);
    transform(P, p, q);
    const float fx = roundf(q[0] * tg.inv_res), fy = roundf(q[1] * tg.inv_res), fz = roundf(q[2] * tg.inv_res);  // Pos2Grid  ivox3d.h:283-286
    const float lim = (float)(kCoordBias - 32);
    search = fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim;  // also false for NaN
    if (search) { cx = (int)fx; cy = (int)fy; cz = (int)fz; }
  }

  // ---- voxel bounding box of the tile ---------------------------------------------------------
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int big = 0x3fffffff;
    int mn[3] = {search ? cx : big, search ? cy : big, search ? cz : big};
    int mx[3] = {search ? cx : -big, search ? cy : -big, search ? cz : -big};
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        mn[a] = min(mn[a], __shfl_xor(mn[a], off, 64));
        mx[a] = max(mx[a], __shfl_xor(mx[a], off, 64));
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; a++) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
    }
  }
  if (threadIdx.x == 0) s_njobs = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int ncell = 1;
    bool ok = kp.use_lds != 0;
    for (int a = 0; a < 3; a++) {
      const int mn = min(min(s_red[0][a], s_red[1][a]), min(s_red[2][a], s_red[3][a]));
      const int mx = max(max(s_red[0][3 + a], s_red[1][3 + a]), max(s_red[2][3 + a], s_red[3][3 + a]));
      if (mx < mn) { ok = false; s_box[a] = 0; s_box[3 + a] = 0; continue; }  // no searchable lane in this tile
      const long long dim = (long long)mx - mn + 3;  // +-1 halo for the 27-cell neighbourhood
      s_box[a] = mn - 1;
      s_box[3 + a] = (int)(dim < 4096 ? dim : 4096);
      if (dim > kCapCells) ok = false;
      ncell = ok ? ncell * (int)dim : ncell;
      if (ncell > kCapCells) ok = false;
    }
    if (ok) {  // bricks under the box
      int nb = 1;
      for (int a = 0; a < 3; a++) {
        const int blo = s_box[a] >> kBrickShift, bhi = (s_box[a] + s_box[3 + a] - 1) >> kBrickShift;
        s_bbox[a] = blo;
        s_bbox[3 + a] = bhi - blo + 1;
        nb *= bhi - blo + 1;
      }
      s_bbox[6] = nb;
      if (nb > kCapBricks) ok = false;
    }
    s_box[6] = ncell;
    s_box[7] = ok ? 1 : 0;
  }
  __syncthreads();
  bool use_lds = s_box[7] != 0;   // uniform over the workgroup
  PCM_STAMP(0)   // load + transform + tile box

  Best best;
  best_init(best);

  if (use_lds) {
    const int ox0 = s_box[0], oy0 = s_box[1], oz0 = s_box[2];
    const int Dx = s_box[3], Dy = s_box[4], Dz = s_box[5], ncell = s_box[6];
    const int bx0 = s_bbox[0], by0 = s_bbox[1], bz0 = s_bbox[2], nby = s_bbox[4], nbz = s_bbox[5], nb = s_bbox[6];
    // ---- one probe per BRICK under the box (wave 0), exclusive scan of their point counts -------
    if (wave == 0) {
      uint32_t npts = 0, ps = 0;
      if (lane < nb) {
        const int z = lane % nbz, xy = lane / nbz, y = xy % nby, x = xy / nby;
        const uint64_t key = pack_brick(bx0 + x, by0 + y, bz0 + z);
        uint32_t h = hash_coord(bx0 + x, by0 + y, bz0 + z) & tg.mask;
        for (;;) {
          const uint4 s0 = gload4u(&tg.bricks[h]);
          if (STATS) n_probe++;
          const uint64_t sk = slot_key(s0);
          if (sk == key) { const uint4 s1 = gload4u(reinterpret_cast<const char*>(&tg.bricks[h]) + 16); ps = s1.x; npts = s1.y; break; }
          if (sk == kEmptyKey) break;
          h = (h + 1) & tg.mask;
        }
      }
      uint32_t incl = npts;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      if (lane < nb) { s_bps[lane] = ps; s_boff[lane] = incl - npts; }
      if (lane == 63) s_boff[kCapBricks] = incl;   // total
    }
    // meanwhile everybody clears the cell grid
#pragma unroll
    for (int j = 0; j < kCapCells / 256; j++) s_cell[threadIdx.x + 256 * j] = kNoCell;
    __syncthreads();
    const uint32_t total = s_boff[kCapBricks];
    use_lds = total <= (uint32_t)kCapPts;   // still uniform
    PCM_STAMP(1)   // brick probes
    if (use_lds) {
      // ---- stage the bricks' map points through LDS: flat, coalesced, all loads in flight --------
      float4 v[kCapPts / 256];
      int vb[kCapPts / 256];
#pragma unroll
      for (int r = 0; r < kCapPts / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        vb[r] = -1;
        if (k < total) {
          int b = 0;
          while (b + 1 < nb && s_boff[b + 1] <= k) b++;   // nb is small (typically 1..8)
          vb[r] = b;
          v[r] = gload4(tg.pts + s_bps[b] + (k - s_boff[b]));
        }
      }
#pragma unroll
      for (int r = 0; r < kCapPts / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        if (k < total) s_pts[k] = v[r];
      }
      __syncthreads();
      PCM_STAMP(2)   // stage map points
      // ---- every voxel head among the staged points registers itself in the cell grid ------------
#pragma unroll
      for (int r = 0; r < kCapPts / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        if (k < total) {
          const int tag = __float_as_int(v[r].w);
          const bool head = k == s_boff[vb[r]] || __float_as_int(s_pts[k - 1].w) != tag;
          if (head) {
            const int b = vb[r], bz = b % nbz, bxy = b / nbz, by = bxy % nby, bx = bxy / nby;
            const int li = tag & 511;
            const int x = ((bx0 + bx) << kBrickShift) + (li >> 6) - ox0, y = ((by0 + by) << kBrickShift) + ((li >> 3) & 7) - oy0,
                      z = ((bz0 + bz) << kBrickShift) + (li & 7) - oz0;
            if (x >= 0 && x < Dx && y >= 0 && y < Dy && z >= 0 && z < Dz) s_cell[(x * Dy + y) * Dz + z] = (uint16_t)k;
          }
        }
      }
      __syncthreads();
      PCM_STAMP(3)   // cell grid
      // ---- per-lane 27-cell / 5-NN search out of LDS (reference cell order) ----------------------
      if (search) {
        const int rx = cx - ox0, ry = cy - oy0, rz = cz - oz0;
        for (int g = 0; g < kp.num_neighbors; g++) {
          const int cell = ((rx + c_nearby[g][0]) * Dy + (ry + c_nearby[g][1])) * Dz + (rz + c_nearby[g][2]);
          uint32_t k = s_cell[cell];
          if (k == kNoCell) continue;
          float4 mp = s_pts[k];
          const int tag = __float_as_int(mp.w);
          for (;;) {   // the voxel's points: consecutive staged points with the same tag
            if (STATS) n_cand++;
            best_offer(best, mp, q, k, kp.max_range_sq);
            if (++k >= total) break;
            mp = s_pts[k];
            if (__float_as_int(mp.w) != tag) break;
          }
        }
      }
    }
  }
  if (!use_lds && search) knn_global<STATS>(tg, q, cx, cy, cz, kp.num_neighbors, kp.max_range_sq, best, n_cand, n_probe);
  PCM_STAMP(4)   // 27-cell / 5-NN search

  // ---- plane fit on the <= 5 neighbours  (laser_mapping.cc:619-623) -----------------------------
  // 5 neighbours (the float path, almost every lane) is solved in place; the rare
  // 3- and 4-neighbour cases (double path) are queued and solved by the first lanes
  // of the workgroup afterwards, so one straggler does not drag its whole wave
  // through the double-precision QR.
  if (live) {
    float4 pl = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
    if (best.m == K) {
      float px[K], py[K], pz[K];
#pragma unroll
      for (int j = 0; j < K; j++) {
        const float4 mp = use_lds ? s_pts[best.i[j]] : gload4(tg.pts + best.i[j]);
        px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
      }
      float4 fit;
      if (esti_plane(px, py, pz, K, kp.plane_threshold, &fit)) pl = fit;
    } else if (best.m >= KMIN) {
      const uint32_t job = atomicAdd(&s_njobs, 1u);
      s_job[job] = threadIdx.x | ((uint32_t)best.m << 16);
#pragma unroll
      for (int j = 0; j < 4; j++) s_jobid[job][j] = best.i[j];
    }
    gstore4(d.planes + i, pl);
  }
  PCM_STAMP(5)   // float plane fit
  __syncthreads();
  for (uint32_t job = threadIdx.x; job < s_njobs; job += 256) {
    const uint32_t owner = s_job[job] & 0xffffu, m = s_job[job] >> 16;
    float px[K], py[K], pz[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
      float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < (int)m) mp = use_lds ? s_pts[s_jobid[job][j]] : gload4(tg.pts + s_jobid[job][j]);
      px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
    }
    float4 fit;
    if (esti_plane(px, py, pz, (int)m, kp.plane_threshold, &fit)) gstore4(d.planes + blockIdx.x * 256u + owner, fit);
  }
  PCM_STAMP(6)   // queued double-precision fits
  if (TIMING && threadIdx.x == 0) atomicAdd(&stats[15], 1ull);
  if (STATS) {
    unsigned long long c = n_cand, pr = n_probe;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off, 64); pr += __shfl_xor(pr, off, 64); }
    if (lane == 0) {
      atomicAdd(&stats[0], c);
      atomicAdd(&stats[1], pr);
      if (wave == 0) { atomicAdd(&stats[2], use_lds ? 1ull : 0ull); atomicAdd(&stats[3], 1ull); }
    }
  }
}

void launch_corr_search(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, unsigned long long* d_stats,
                        bool timing) {
  dim3 grid((unsigned)kp.tiles_per_pair, (unsigned)npairs);
  if (d_stats && timing) k_corr_search<false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  else if (d_stats) k_corr_search<true, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  else k_corr_search<false, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
}

// ---------------------------------------------------------------------------
// k_residual_reduce: LINEARIZE -> H, b, cost, #inliers of the planes fitted by
// k_corr_search; TRIAL -> cost of the same correspondences at the trial pose
// (compute_error contract, fast_gicp_impl.hpp:213-237).
// grid = (blocks_per_pair, npairs), block = 256
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <bool WRITE_SEL>
__global__ void __launch_bounds__(256) k_residual_reduce(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp) {
  const int pair = blockIdx.y;
  const int mode = states[pair].mode;
  if (mode == MODE_DONE) return;
  const PairDesc d = descs[pair];
  const PoseF P = load_pose(mode == MODE_LINEARIZE ? states[pair].x0 : states[pair].xi);

  double acc[kNumSums];
#pragma unroll
  for (int j = 0; j < kNumSums; j++) acc[j] = 0.0;

  const uint32_t begin = blockIdx.x * (uint32_t)kp.points_per_block;
  uint32_t end = begin + (uint32_t)kp.points_per_block;
  end = end < d.src.num_points ? end : d.src.num_points;

  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    const float4 pl = gload4(d.planes + i);
    if (pl.x != pl.x) continue;  // no plane / not selected
    const float4 p = gload4(d.src.pts + i);
    float q[3];
    transform(P, p, q);
    const float pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;  // laser_mapping.cc:627-629
    const double e = (double)pd2;
    if (mode == MODE_LINEARIZE) {
      const float pn = pcm_sqrtf_rn(p.x * p.x + p.y * p.y + p.z * p.z);
      const bool sel = pn > 81.f * pd2 * pd2;                          // :631
      if (!sel) {
        if (WRITE_SEL) gstore_f(&d.planes[i].x, __builtin_nanf(""));            // dropped for the following trial passes too
        continue;
      }
      // left-perturbation Jacobian of e = n.(T p) + d :  [ (q x n)^T , n^T ]
      const float jf[6] = {q[1] * pl.z - q[2] * pl.y, q[2] * pl.x - q[0] * pl.z, q[0] * pl.y - q[1] * pl.x, pl.x, pl.y, pl.z};
      double J[6];
#pragma unroll
      for (int a = 0; a < 6; a++) J[a] = (double)jf[a];
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = a; c < 6; c++) { acc[t] = fma(J[a], J[c], acc[t]); t++; }
      }
#pragma unroll
      for (int a = 0; a < 6; a++) acc[21 + a] = fma(J[a], e, acc[21 + a]);
    }
    acc[27] = fma(e, e, acc[27]);
    acc[28] += 1.0;
  }

  // wave reduce -> LDS -> one partial row per workgroup
  __shared__ double s_part[4][kPartialStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < kNumSums; j++) {
    const double v = wave_sum(acc[j]);
    if (lane == 0) s_part[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    const double v = ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) + s_part[3][threadIdx.x];
    gstore_d(d.partials + (size_t)blockIdx.x * kPartialStride + threadIdx.x, v);
  }
}

void launch_residual(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_sel) {
  dim3 grid((unsigned)kp.blocks_per_pair, (unsigned)npairs);
  if (write_sel) k_residual_reduce<true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  else k_residual_reduce<false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
}

// ---------------------------------------------------------------------------
// second stage: fixed-order sum of the workgroup partials + GN/LM state machine
// grid = npairs, block = 256
// ---------------------------------------------------------------------------
__device__ inline void sum_partials(const double* __restrict__ partials, int blocks_per_pair, double* s_sum /* [8][32] then [0][*] holds the result */) {
  const int j = threadIdx.x & 31, r = threadIdx.x >> 5;  // 8 row groups x 32 columns
  double v = 0.0;
  if (j < kNumSums) {
    for (int b = r; b < blocks_per_pair; b += 8) v += gload_d(partials + (size_t)b * kPartialStride + j);
  }
  s_sum[r * kPartialStride + j] = v;
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    double t = 0.0;
    for (int k = 0; k < 8; k++) t += s_sum[k * kPartialStride + threadIdx.x];
    s_sum[8 * kPartialStride + threadIdx.x] = t;
  }
  __syncthreads();
}

__device__ inline void unpack_sums(const double* s, double* H, double* b, double* cost, int* inliers) {
  int t = 0;
  for (int a = 0; a < 6; a++) {
    for (int c = a; c < 6; c++) { H[a * 6 + c] = s[t]; H[c * 6 + a] = s[t]; t++; }
  }
  for (int a = 0; a < 6; a++) b[a] = s[21 + a];
  *cost = s[27];
  *inliers = (int)s[28];
}

__global__ void __launch_bounds__(256) k_lsq_step(const PairDesc* __restrict__ descs, PairState* __restrict__ states, LsqParams lp, int blocks_per_pair,
                                                  int* __restrict__ active_slot) {
  const int pair = blockIdx.x;
  __shared__ double s_sum[9 * kPartialStride];
  const int mode = states[pair].mode;
  if (mode == MODE_DONE) return;
  sum_partials(descs[pair].partials, blocks_per_pair, s_sum);
  if (threadIdx.x == 0) {
    const double* s = s_sum + 8 * kPartialStride;
    PairState& st = states[pair];
    if (mode == MODE_LINEARIZE) {
      double H[36], b[6], cost;
      int inl;
      unpack_sums(s, H, b, &cost, &inl);
      after_linearize(st, lp, H, b, cost, inl);
    } else {
      after_trial(st, lp, s[27]);
    }
    if (st.mode != MODE_DONE) atomicAdd(active_slot, 1);
  }
}

void launch_lsq_step(hipStream_t stream, const PairDesc* d_descs, PairState* d_states, const LsqParams& lp, int blocks_per_pair, int npairs, int* d_active_slot) {
  k_lsq_step<<<npairs, 256, 0, stream>>>(d_descs, d_states, lp, blocks_per_pair, d_active_slot);
}

__global__ void __launch_bounds__(256) k_reduce_only(const PairDesc* __restrict__ descs, int blocks_per_pair, double* __restrict__ sums) {
  __shared__ double s_sum[9 * kPartialStride];
  sum_partials(descs[blockIdx.x].partials, blocks_per_pair, s_sum);
  if (threadIdx.x < kNumSums) sums[blockIdx.x * kPartialStride + threadIdx.x] = s_sum[8 * kPartialStride + threadIdx.x];
}

void launch_reduce_only(hipStream_t stream, const PairDesc* d_descs, int blocks_per_pair, int npairs, double* d_sums) {
  k_reduce_only<<<npairs, 256, 0, stream>>>(d_descs, blocks_per_pair, d_sums);
}

__global__ void k_init_states(PairState* __restrict__ states, const float* __restrict__ guesses, int npairs, int max_iterations) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= npairs) return;
  PairState s;
  init_state(s, guesses + pair * 16);
  if (max_iterations <= 0) s.mode = MODE_DONE;
  states[pair] = s;
}

void launch_init_states(hipStream_t stream, PairState* d_states, const float* d_guesses, int npairs, int max_iterations) {
  k_init_states<<<(npairs + 63) / 64, 64, 0, stream>>>(d_states, d_guesses, npairs, max_iterations);
}

__global__ void k_pack_results(const PairState* __restrict__ states, pcm_result* __restrict__ out, int npairs) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= npairs) return;
  const PairState& s = states[pair];
  pcm_result r;
  for (int i = 0; i < 16; i++) { r.T[i] = (float)s.x0[i]; r.T64[i] = s.x0[i]; }  // final_transformation_ = x0.cast<float>()
  for (int i = 0; i < 36; i++) r.H[i] = s.final_hessian[i];
  r.cost = s.last_cost;
  r.iterations = s.iter;
  r.converged = s.converged;
  r.num_linearize = s.num_linearize;
  r.num_compute_error = s.num_compute_error;
  r.num_inliers = s.num_inliers;
  r.status = s.status;
  out[pair] = r;
}

void launch_pack_results(hipStream_t stream, const PairState* d_states, pcm_result* d_results, int npairs) {
  k_pack_results<<<(npairs + 63) / 64, 64, 0, stream>>>(d_states, d_results, npairs);
}

}  // namespace pcm

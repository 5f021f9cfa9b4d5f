// linearize_common.h -- device helpers shared by the point-to-plane search kernels (kernels.hip: the per-tile LDS search and
// the LIO variant; neighbour_lists.hip: the walk of per-voxel candidate lists; linearize_counted.hip, linearize_reforder.hip):
// neighbour-cell tables in the reference's order
// (jueying_lio/include/ivox3d/ivox3d.h:211-235), the float pose transform of laser_mapping.cc:602-612, the running 5-best list
// of IVoxNode::KNNPointByCondition / IVox::GetClosestPoint (ivox3d_node.hpp:140-205, ivox3d.h:132-204) and the per-lane search
// against the global brick hash.
#pragma once

#include "pcm_device.h"
#include "plane_fit.h"

namespace pcm {

// neighbour cells in the reference's order (ivox3d.h:211-235): CENTER, NEARBY6, NEARBY18, NEARBY26 are prefixes
// (compile-time copy for the unrolled LDS search: the offsets become immediates)
constexpr int kNearby[27][3] = {
  {0, 0, 0},   {-1, 0, 0},  {1, 0, 0},   {0, 1, 0},   {0, -1, 0},  {0, 0, -1},  {0, 0, 1},
  {1, 1, 0},   {-1, 1, 0},  {1, -1, 0},  {-1, -1, 0}, {1, 0, 1},   {-1, 0, 1},  {1, 0, -1},
  {-1, 0, -1}, {0, 1, 1},   {0, -1, 1},  {0, 1, -1},  {0, -1, -1}, {1, 1, 1},   {-1, 1, 1},
  {1, -1, 1},  {1, 1, -1},  {-1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};
// IEKF reduction terms (LIO rows = 12 Jacobian columns, h, selected): 78 x HTH upper triangle, 12 x H^T h, sum h^2, count
static __constant__ int8_t c_lio_a[96] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 8, 8, 8, 8, 9, 9, 9, 10, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 0, 0, 0, 0};
static __constant__ int8_t c_lio_b[96] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 3, 4, 5, 6, 7, 8, 9, 10, 11, 4, 5, 6, 7, 8, 9, 10, 11, 5, 6, 7, 8, 9, 10, 11, 6, 7, 8, 9, 10, 11, 7, 8, 9, 10, 11, 8, 9, 10, 11, 9, 10, 11, 10, 11, 11, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 13, 0, 0, 0, 0};
// term j of the normal equations = row[c_term_a[j]] * row[c_term_b[j]] with row = (J0..J5, e, selected)
static __constant__ int8_t c_term_a[32] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6, 7, 0, 0, 0};
static __constant__ int8_t c_term_b[32] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6, 7, 0, 0, 0};
static __constant__ int8_t c_nearby[27][4] = {
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
  float thr;   // min(d[K - 1], max_range^2): a candidate changes the list iff its distance is below this
  int m;       // candidates within max_range, at most K (= finite entries of d; set by best_finish)
};

__device__ inline void best_init(Best& b, float max_r2f) {
#pragma unroll
  for (int j = 0; j < K; j++) { b.d[j] = __builtin_inff(); b.i[j] = 0xffffffffu; }
  b.thr = max_r2f;
  b.m = 0;
}

// every in-range candidate enters the list while it has a free (infinite) slot, so the reference's
// count min(K, #in range) is the number of finite entries
__device__ inline void best_finish(Best& b) {
  int m = 0;
#pragma unroll
  for (int j = 0; j < K; j++) m += b.d[j] < __builtin_inff() ? 1 : 0;
  b.m = m;
}

// Offer one map point.  `max_r2f` is the smallest float >= max_range^2, so
// `d2 < max_r2f` is exactly the reference's `double(d2) < max_range * max_range`
// (ivox3d_node.hpp:162) without a double-precision compare per candidate.
// Sorted insert with strict '<' (an equal distance goes behind the entries
// already there = visit order); new d[j] = median(d2, d[j-1], d[j]).
__device__ inline void best_offer(Best& b, const float4& mp, const float (&q)[3], uint32_t id, float max_r2f) {
  const float dx = mp.x - q[0], dy = mp.y - q[1], dz = mp.z - q[2];
  const float d2 = dx * dx + dy * dy + dz * dz;  // distance2()  ivox3d_node.hpp:13-16
  if (d2 < b.thr) {   // in range (d2 < max_r2f) and ahead of the current K-th (strict: an equal distance stays behind)
    {
      const bool c0 = d2 < b.d[0], c1 = d2 < b.d[1], c2 = d2 < b.d[2], c3 = d2 < b.d[3];
      b.i[4] = c3 ? b.i[3] : id;
      // distances are non-negative and never NaN here: unsigned integer min / max of the bit patterns order them
      // exactly and skip the NaN canonicalisation fminf / fmaxf carry
      b.d[4] = __uint_as_float(max(__float_as_uint(d2), __float_as_uint(b.d[3])));
      b.i[3] = c2 ? b.i[2] : (c3 ? id : b.i[3]);
      b.d[3] = __builtin_amdgcn_fmed3f(d2, b.d[2], b.d[3]);
      b.i[2] = c1 ? b.i[1] : (c2 ? id : b.i[2]);
      b.d[2] = __builtin_amdgcn_fmed3f(d2, b.d[1], b.d[2]);
      b.i[1] = c0 ? b.i[0] : (c1 ? id : b.i[1]);
      b.d[1] = __builtin_amdgcn_fmed3f(d2, b.d[0], b.d[1]);
      b.i[0] = c0 ? id : b.i[0];
      b.d[0] = __uint_as_float(min(__float_as_uint(d2), __float_as_uint(b.d[0])));
      b.thr = __uint_as_float(min(__float_as_uint(b.d[4]), __float_as_uint(max_r2f)));
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

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- wave64 data-parallel primitives on the DPP path (no LDS round trip: __shfl_* compiles to ds_bpermute_b32, one LDS-latency
// hop per step).  gfx9 DPP controls: row_shr:n = 0x110 + n (shift inside a row of 16 lanes), row_bcast:15 = 0x142 (lane 15 of a
// row to the next row), row_bcast:31 = 0x143 (lane 31 to rows 2 and 3).  A lane without a source keeps `old`.
template <int CTRL, int ROW_MASK = 0xf>
__device__ inline int dpp_mov(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }

// inclusive prefix sum inside each half (lanes 0..31, 32..63) of the wave
__device__ inline uint32_t scan_add_half(uint32_t v) {
  int x = (int)v;
  x += dpp_mov<0x111>(0, x);
  x += dpp_mov<0x112>(0, x);
  x += dpp_mov<0x114>(0, x);
  x += dpp_mov<0x118>(0, x);
  x += dpp_mov<0x142, 0xa>(0, x);   // rows 1 and 3 add the total of rows 0 and 2
  return (uint32_t)x;
}
// inclusive prefix sum over the wave
__device__ inline uint32_t scan_add_wave(uint32_t v) {
  int x = (int)scan_add_half(v);
  x += dpp_mov<0x143, 0xc>(0, x);   // rows 2 and 3 add the total of the lower half
  return (uint32_t)x;
}
// minimum / maximum over the wave, returned to every lane
__device__ inline int wave_min_i32(int x) {
  x = min(x, dpp_mov<0x111>(x, x));
  x = min(x, dpp_mov<0x112>(x, x));
  x = min(x, dpp_mov<0x114>(x, x));
  x = min(x, dpp_mov<0x118>(x, x));
  x = min(x, dpp_mov<0x142, 0xa>(x, x));
  x = min(x, dpp_mov<0x143, 0xc>(x, x));
  return __builtin_amdgcn_readlane(x, 63);
}
__device__ inline int wave_max_i32(int x) {
  x = max(x, dpp_mov<0x111>(x, x));
  x = max(x, dpp_mov<0x112>(x, x));
  x = max(x, dpp_mov<0x114>(x, x));
  x = max(x, dpp_mov<0x118>(x, x));
  x = max(x, dpp_mov<0x142, 0xa>(x, x));
  x = max(x, dpp_mov<0x143, 0xc>(x, x));
  return __builtin_amdgcn_readlane(x, 63);
}

// ---- tail of a linearize tile: residual / Jacobian row of every lane -> 29 sums of the tile -> one partial row.
// e = n.(T p) + d (laser_mapping.cc:627-629), valid iff |p_body| > 81 e^2 (:631), left-perturbation Jacobian [ (q x n)^T , n^T ];
// products formed in double from the float values like h_x (esekfom.hpp:1687), 8 groups of 32 rows each summed in row order, then
// the groups in order: the fixed order k_finish_round continues.  All 256 threads call it; `smem` is 16-byte aligned LDS of at
// least kReduceLdsBytes that nobody else touches any more (the caller's last barrier lies behind every other use); two barriers inside.
constexpr int kReduceLdsBytes = 256 * 8 * 8 + 8 * kPartialStride * 8;
template <bool WRITE_PLANES>
__device__ inline void residual_and_reduce(const PairDesc& d, uint32_t i, uint32_t tile_x, bool live, const float4& pl, const float (&q)[3], float pn_body, void* smem) {
  double* const s_row = reinterpret_cast<double*>(smem);   // [256][8]: J0..J5, e, selected
  double* const s_grp = s_row + 256 * 8;                   // [8][32] group partials
  {
    float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
      bool sel = !(pl.x != pl.x);
      if (sel) {
        const float pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;
        sel = pn_body > 81.f * pd2 * pd2;
        if (sel) {
          row[0] = q[1] * pl.z - q[2] * pl.y;
          row[1] = q[2] * pl.x - q[0] * pl.z;
          row[2] = q[0] * pl.y - q[1] * pl.x;
          row[3] = pl.x; row[4] = pl.y; row[5] = pl.z;
          row[6] = pd2;
          row[7] = 1.f;
        }
      }
      if (WRITE_PLANES) gstore4(d.planes + i, sel ? pl : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f));   // the selected set, for trial passes / parity hooks
    }
    double2* dst = reinterpret_cast<double2*>(s_row + threadIdx.x * 8);
#pragma unroll
    for (int a = 0; a < 4; a++) dst[a] = make_double2((double)row[2 * a], (double)row[2 * a + 1]);
  }
  __syncthreads();
  {
    const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
    double v = 0.0;
    if (j < kNumSums) {
      const int ia = c_term_a[j], ib = c_term_b[j];
      const double* r0 = s_row + (g * 32) * 8;
#pragma unroll 8
      for (int k = 0; k < 32; k++) v = fma(r0[k * 8 + ia], r0[k * 8 + ib], v);
    }
    s_grp[g * kPartialStride + j] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    double v = 0.0;
#pragma unroll
    for (int g = 0; g < 8; g++) v += s_grp[g * kPartialStride + threadIdx.x];
    gstore_d(d.partials + (size_t)tile_x * kPartialStride + threadIdx.x, v);
  }
}

}  // namespace pcm

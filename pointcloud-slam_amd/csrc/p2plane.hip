// p2plane.hip -- fused correspondence search + point-to-plane residual/Jacobian +
// 6x6 normal-equation reduction for gfx950 (MI355X), and the per-pair GN/LM step.
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
// Shape (not a port): one lane = one scan point per pass; the kNN, plane fit,
// residual and the 28 unique normal-equation terms stay in registers; float
// geometry, double accumulation; wave64 cross-lane reduction -> LDS across the
// 4 waves of a 256-thread workgroup -> one partial row per workgroup; a
// deterministic fixed-order second stage and the 6x6 solve run in a tiny
// kernel per round.  No correspondence list is ever written to HBM and there
// is no host round trip inside the GN/LM loop.
//
// This translation unit is compiled with -ffp-contract=off: the float geometry
// that feeds discrete decisions (voxel key, kNN order, plane test) must round
// like the reference's plain x86 build; the double accumulations use explicit fma().
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

// ---------------------------------------------------------------------------
// wave64 / workgroup reduction of the per-lane double accumulators
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

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

// 5-NN of q among the points of the `num_neighbors` cells around it (global-memory probing).
// Returns m = number found (<= K); idx/d sorted by ascending distance, ties in visit order.
template <bool STATS>
__device__ inline int knn_global(const TargetView& tg, const float (&q)[3], int num_neighbors, double max_range_sq, float (&bd)[K], uint32_t (&bi)[K],
                                 uint32_t& n_cand, uint32_t& n_probe) {
  const int cx = (int)roundf(q[0] * tg.inv_res), cy = (int)roundf(q[1] * tg.inv_res), cz = (int)roundf(q[2] * tg.inv_res);  // Pos2Grid ivox3d.h:283-286
#pragma unroll
  for (int j = 0; j < K; j++) { bd[j] = __builtin_inff(); bi[j] = 0xffffffffu; }
  int m = 0;
  const uint32_t hbase = hash_part_x(cx) + hash_part_y(cy) + hash_part_z(cz);
  const int64_t kbase = (int64_t)pack_key(cx, cy, cz);
  for (int g = 0; g < num_neighbors; g++) {
    const int ox = c_nearby[g][0], oy = c_nearby[g][1], oz = c_nearby[g][2];
    const uint32_t hsum = hbase + (uint32_t)ox * 0x9E3779B1u + (uint32_t)oy * 0x85EBCA77u + (uint32_t)oz * 0xC2B2AE3Du;
    const uint64_t key = (uint64_t)(kbase + ((int64_t)ox << 42) + ((int64_t)oy << 21) + (int64_t)oz);
    uint32_t h = hash_finish(hsum) & tg.mask;
    uint32_t start = 0, count = 0;
    for (;;) {
      const uint4 s = *reinterpret_cast<const uint4*>(&tg.slots[h]);
      if (STATS) n_probe++;
      const uint64_t sk = ((uint64_t)s.y << 32) | s.x;
      if (sk == key) { start = s.z; count = s.w; break; }
      if (sk == kEmptyKey) break;
      h = (h + 1) & tg.mask;
    }
    for (uint32_t s = start; s < start + count; s++) {
      const float4 mp = tg.pts[s];
      if (STATS) n_cand++;
      const float dx = mp.x - q[0], dy = mp.y - q[1], dz = mp.z - q[2];
      float d2 = dx * dx + dy * dy + dz * dz;  // distance2()  ivox3d_node.hpp:13-16
      if ((double)d2 < max_range_sq) {         // d < max_range * max_range  ivox3d_node.hpp:162
        uint32_t id = s;
        m = m < K ? m + 1 : K;
        // stable sorted insert (strict <: equal distances keep visit order)
        bool shifting = false;  // once the new entry is placed, the displaced ones just move down
#pragma unroll
        for (int j = 0; j < K; j++) {
          const bool lt = shifting | (d2 < bd[j]);
          shifting = lt;
          const float td = lt ? bd[j] : d2;
          const uint32_t ti = lt ? bi[j] : id;
          bd[j] = lt ? d2 : bd[j];
          bi[j] = lt ? id : bi[j];
          d2 = td;
          id = ti;
        }
      }
    }
  }
  return m;
}

// ---------------------------------------------------------------------------
// residual kernel: LINEARIZE (kNN + plane fit + J^T J) or TRIAL (cost only on
// the stored planes) per pair, selected by the pair's state.
// grid = (blocks_per_pair, npairs), block = 256
// ---------------------------------------------------------------------------
template <bool WRITE_PLANES, bool STATS>
__global__ void __launch_bounds__(256) k_p2plane(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp,
                                                 unsigned long long* __restrict__ stats) {
  const int pair = blockIdx.y;
  const int mode = states[pair].mode;
  if (mode == MODE_DONE) return;
  const PairDesc d = descs[pair];
  const PoseF P = load_pose(mode == MODE_LINEARIZE ? states[pair].x0 : states[pair].xi);

  double acc[kNumSums];
#pragma unroll
  for (int j = 0; j < kNumSums; j++) acc[j] = 0.0;
  uint32_t n_cand = 0, n_probe = 0;

  const uint32_t begin = blockIdx.x * (uint32_t)kp.points_per_block;
  uint32_t end = begin + (uint32_t)kp.points_per_block;
  end = end < d.src.num_points ? end : d.src.num_points;

  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    const float4 p = d.src.pts[i];
    float q[3];
    transform(P, p, q);
    float4 pl;
    bool sel;
    if (mode == MODE_LINEARIZE) {
      float bd[K];
      uint32_t bi[K];
      const int m = knn_global<STATS>(d.tgt, q, kp.num_neighbors, kp.max_range_sq, bd, bi, n_cand, n_probe);
      sel = m >= KMIN;
      if (sel) {
        float px[K], py[K], pz[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
          if (j < m) { const float4 mp = d.tgt.pts[bi[j]]; px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z; }
          else { px[j] = 0.f; py[j] = 0.f; pz[j] = 0.f; }
        }
        sel = esti_plane(px, py, pz, m, kp.plane_threshold, &pl);
      }
      float pd2 = 0.f;
      if (sel) {
        pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;                      // laser_mapping.cc:627-629
        const float pn = pcm_sqrtf_rn(p.x * p.x + p.y * p.y + p.z * p.z);
        sel = pn > 81.f * pd2 * pd2;                                               // :631
      }
      if (WRITE_PLANES) d.planes[i] = sel ? pl : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
      if (sel) {
        // left-perturbation Jacobian of e = n.(T p) + d :  [ (q x n)^T , n^T ]
        const float jf[6] = {q[1] * pl.z - q[2] * pl.y, q[2] * pl.x - q[0] * pl.z, q[0] * pl.y - q[1] * pl.x, pl.x, pl.y, pl.z};
        double J[6];
#pragma unroll
        for (int a = 0; a < 6; a++) J[a] = (double)jf[a];
        const double e = (double)pd2;
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
          for (int c = a; c < 6; c++) { acc[t] = fma(J[a], J[c], acc[t]); t++; }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] = fma(J[a], e, acc[21 + a]);
        acc[27] = fma(e, e, acc[27]);
        acc[28] += 1.0;
      }
    } else {
      pl = d.planes[i];
      if (!(pl.x != pl.x)) {  // selected in the last linearize
        const float pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;
        const double e = (double)pd2;
        acc[27] = fma(e, e, acc[27]);
        acc[28] += 1.0;
      }
    }
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
    d.partials[(size_t)blockIdx.x * kPartialStride + threadIdx.x] = v;
  }
  if (STATS) {
    unsigned long long c = n_cand, pr = n_probe;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off, 64); pr += __shfl_xor(pr, off, 64); }
    if (lane == 0) { atomicAdd(&stats[0], c); atomicAdd(&stats[1], pr); }
  }
}

void launch_p2plane(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes,
                    unsigned long long* d_stats) {
  dim3 grid((unsigned)kp.blocks_per_pair, (unsigned)npairs);
  if (d_stats) {
    if (write_planes) k_p2plane<true, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
    else k_p2plane<false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  } else {
    if (write_planes) k_p2plane<true, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
    else k_p2plane<false, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  }
}

// ---------------------------------------------------------------------------
// second stage: fixed-order sum of the workgroup partials + GN/LM state machine
// grid = npairs, block = 64
// ---------------------------------------------------------------------------
__device__ inline void sum_partials(const double* __restrict__ partials, int blocks_per_pair, double* s_sum) {
  const int j = threadIdx.x;
  if (j < kNumSums) {
    double v = 0.0;
    for (int b = 0; b < blocks_per_pair; b++) v += partials[(size_t)b * kPartialStride + j];
    s_sum[j] = v;
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

__global__ void __launch_bounds__(64) k_lsq_step(const PairDesc* __restrict__ descs, PairState* __restrict__ states, LsqParams lp, int blocks_per_pair,
                                                 int* __restrict__ active_slot) {
  const int pair = blockIdx.x;
  __shared__ double s_sum[kPartialStride];
  const int mode = states[pair].mode;
  if (mode == MODE_DONE) return;
  sum_partials(descs[pair].partials, blocks_per_pair, s_sum);
  if (threadIdx.x == 0) {
    PairState& st = states[pair];
    if (mode == MODE_LINEARIZE) {
      double H[36], b[6], cost;
      int inl;
      unpack_sums(s_sum, H, b, &cost, &inl);
      after_linearize(st, lp, H, b, cost, inl);
    } else {
      after_trial(st, lp, s_sum[27]);
    }
    if (st.mode != MODE_DONE) atomicAdd(active_slot, 1);
  }
}

void launch_lsq_step(hipStream_t stream, const PairDesc* d_descs, PairState* d_states, const LsqParams& lp, int blocks_per_pair, int npairs, int* d_active_slot) {
  k_lsq_step<<<npairs, 64, 0, stream>>>(d_descs, d_states, lp, blocks_per_pair, d_active_slot);
}

__global__ void __launch_bounds__(64) k_reduce_only(const PairDesc* __restrict__ descs, int blocks_per_pair, double* __restrict__ sums) {
  __shared__ double s_sum[kPartialStride];
  sum_partials(descs[blockIdx.x].partials, blocks_per_pair, s_sum);
  if (threadIdx.x < kNumSums) sums[blockIdx.x * kPartialStride + threadIdx.x] = s_sum[threadIdx.x];
}

void launch_reduce_only(hipStream_t stream, const PairDesc* d_descs, int blocks_per_pair, int npairs, double* d_sums) {
  k_reduce_only<<<npairs, 64, 0, stream>>>(d_descs, blocks_per_pair, d_sums);
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

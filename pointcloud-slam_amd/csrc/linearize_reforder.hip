// linearize_reforder.hip -- k_linearize_reforder: the linearize pass of the point-to-plane path with the neighbours handed to the
// plane fit in the REFERENCE'S OWN ROW ORDER (PCM_FLAG_REFERENCE_KNN_ORDER).  A compatibility mode, not the fast path.
//
// IVox::GetClosestPoint (/root/reference/src/jueying_lio/include/ivox3d/ivox3d.h:132-204) collects, voxel by voxel in nearby_grids_
// order, every point within max_range as DistPoint{dist, idx}; a voxel that contributed more than K keeps the K that
// std::nth_element(begin + old, begin + old + K - 1, end) leaves in front (ivox3d_node.hpp:176-181); at the end
// nth_element(begin, begin + K - 1, end) + resize(K) if more than K were collected, then nth_element(begin, begin, end)
// (ivox3d.h:173-178).  The neighbour SET is the K nearest either way (ties at the K-th distance aside), but the ORDER of the rows of
// the 5 x 3 system esti_plane factorises (common_lib.h:199-208) is whatever libstdc++'s introselect leaves, and a row-permuted
// float QR rounds differently: planes differ in the last bits, marginal `|n.p + d| > 0.1` verdicts flip, poses move by up to
// ~1e-4 m (profiles/r03_knn_order_sensitivity.json).  The default kernels sort ascending; this kernel reproduces the reference
// order exactly: the candidate array of every scan point is built in the reference's visit order in private memory and put
// through the same three selections by nth_select.h, a restatement of libstdc++'s std::nth_element checked against the real
// one permutation for permutation (tests/test_knn_order.py).  The CPU check has a "libstdcxx" order mode of its own, which
// calls the container's std::nth_element itself; GPU == oracle bit for bit (tests/test_gpu_reforder.py).
//
// Shape: one scan point per lane, candidates straight from the global brick hash (no LDS staging: the per-lane candidate array
// lives in scratch memory and dominates anyway), one plane fit per lane, then the shared residual / reduction tail.  Private
// array: 27 x K survivors + one voxel's points; the host refuses maps whose voxels hold more than kRefMaxVoxelPoints.
//
// Compiled with -ffp-contract=off (see kernels.hip).
#include "pcm_device.h"
#include "pcm_host.h"
#include "plane_fit.h"
#include "linearize_common.h"
#include "nth_select.h"

namespace pcm {

constexpr int kRefCap = 27 * K + kRefMaxVoxelPoints;

template <bool WRITE_PLANES>
__global__ void __launch_bounds__(256) k_linearize_reforder(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp) {
  const uint32_t tile_x = blockIdx.x;
  const int pair = PCM_PAIR_OF(kp, blockIdx.y);
  if (states[pair].mode != MODE_LINEARIZE) return;
  const PairDesc d = descs[pair];
  const uint32_t i = tile_x * 256u + threadIdx.x;
  if (tile_x * 256u >= d.src.num_points) return;
  const bool live = i < d.src.num_points;
  const PoseF P = load_pose(states[pair].x0);
  const TargetView tg = d.tgt;
  __shared__ __align__(16) unsigned char s_mem[kReduceLdsBytes];

  float4 pl = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
  float q[3] = {0.f, 0.f, 0.f};
  float pn_body = 0.f;
  if (live) {
    const float4 p = gload4(d.src.pts + i);
    pn_body = pcm_sqrtf_rn(p.x * p.x + p.y * p.y + p.z * p.z);
    transform(P, p, q);
    const float fx = roundf(q[0] * tg.inv_res), fy = roundf(q[1] * tg.inv_res), fz = roundf(q[2] * tg.inv_res);  // Pos2Grid  ivox3d.h:283-286
    const float lim = (float)(kCoordBias - 32);
    if (fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim) {
      const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
      DistId cand[kRefCap];
      int n = 0;
      int cbx = 0x7fffffff, cby = 0, cbz = 0;
      uint32_t slot = ~0u, vox_base = 0, n_probe = 0;
      for (int g = 0; g < kp.num_neighbors; g++) {   // nearby_grids_ order  ivox3d.h:211-235
        const int vx = cx + c_nearby[g][0], vy = cy + c_nearby[g][1], vz = cz + c_nearby[g][2];
        const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
        if (bx != cbx || by != cby || bz != cbz) {
          slot = brick_find<false>(tg, bx, by, bz, vox_base, n_probe);
          cbx = bx; cby = by; cbz = bz;
        }
        if (slot == ~0u) continue;
        const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
        const uint32_t m = gload_u(&tg.bmask[(size_t)slot * 16 + w]);
        if (!((m >> bit) & 1u)) continue;
        const uint32_t v = vox_base + gload_u16(&tg.bpref[(size_t)slot * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u));
        const uint32_t start = gload_u(&tg.vox_start[v]), end = gload_u(&tg.vox_start[v + 1]);
        const int old = n;
        for (uint32_t k = start; k < end && n < kRefCap; k++) {   // the voxel's points in insertion order (points_)  ivox3d_node.hpp:158-166
          const float4 mp = gload4(tg.pts + k);
          const float dx = mp.x - q[0], dy = mp.y - q[1], dz = mp.z - q[2];
          const float d2 = dx * dx + dy * dy + dz * dz;
          if (d2 < kp.max_range_sq) { cand[n].d = d2; cand[n].id = k; n++; }   // kp.max_range_sq: see best_offer (exactly the reference's double compare)
        }
        if (n - old > K) {   // ivox3d_node.hpp:176-181
          nth_element_libstdcxx(cand + old, K - 1, n - old);
          n = old + K;
        }
      }
      if (n > K) {           // ivox3d.h:173-176
        nth_element_libstdcxx(cand, K - 1, n);
        n = K;
      }
      if (n > 0) nth_element_libstdcxx(cand, 0, n);   // :178
      if (n >= KMIN) {       // laser_mapping.cc:619-623
        float px[K], py[K], pz[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
          float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < n) mp = gload4(tg.pts + cand[j].id);
          px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
        }
        float4 fit;
        if (esti_plane(px, py, pz, n, kp.plane_threshold, &fit)) pl = fit;
      }
    }
  }
  residual_and_reduce<WRITE_PLANES>(d, i, tile_x, live, pl, q, pn_body, s_mem);
}

void launch_linearize_reforder(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes) {
  dim3 grid((unsigned)kp.tiles_per_pair, (unsigned)npairs);
  if (write_planes) k_linearize_reforder<true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  else k_linearize_reforder<false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
}

}  // namespace pcm

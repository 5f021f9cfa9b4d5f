// ndt.hip -- NDT residual models (point-to-distribution, distribution-to-distribution)
// on the brick voxel hash, gfx950.
//
// Replaces, for the MI355X path (paths relative to
// /root/reference/src/pointcloud_match/fast_gicp):
//   find_voxel_correspondences ......... src/fast_gicp/cuda/find_voxel_correspondences.cu:16-111
//   p2d / d2d_ndt_compute_derivatives ... src/fast_gicp/cuda/ndt_compute_derivatives.cu:33-231
//   NDTCudaCore update / compute_error .. src/fast_gicp/cuda/ndt_cuda.cu:142-177
// Shape (not a port): the reference materialises an (source, voxel) pair list with one
// thrust::async::transform per neighbour offset, compacts it with remove_if, then
// transform_reduces 43-float tuples and uploads two Isometry3f per call.  Here one
// kernel per round looks the <= 27 cells up (brick probe re-used while consecutive
// cells stay in one brick -> occupancy bit -> rank), forms the per-correspondence
// terms in float exactly as the reference writes them, accumulates the 28 unique
// normal-equation terms per lane in double, and reduces per workgroup; the matched
// voxel indices (4 B per (element, offset)) are kept for the LM trial passes, which
// must re-use the correspondences of the last linearize.
// Compiled with -ffp-contract=off (see kernels.hip).
#include "pcm_device.h"
#include "dev_linalg.h"
#include "pcm_host.h"

namespace pcm {

// neighbour offsets in the reference's order (ndt_cuda.cu:35-88)
__constant__ int8_t c_direct7[7][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {-1, 0, 0, 0}, {0, 1, 0, 0}, {0, -1, 0, 0}, {0, 0, 1, 0}, {0, 0, -1, 0}};

// DIRECT_RADIUS: entry k of the cube [-range, range]^3 (i, j, k loops); false when the reference's list does not hold it
__device__ inline bool ndt_offset_radius(int range, double radius, int k, int& ox, int& oy, int& oz) {
  const int D = 2 * range + 1;
  ox = k / (D * D) - range; oy = (k / D) % D - range; oz = k % D - range;
  return sqrt((double)(ox * ox + oy * oy + oz * oz)) <= radius + 1e-3;   // offset.cast<double>().norm() <= radius + 1e-3
}

__device__ inline void ndt_offset(int nO, int k, int& ox, int& oy, int& oz) {
  if (nO == 27) { ox = k / 9 - 1; oy = (k / 3) % 3 - 1; oz = k % 3 - 1; }   // i, j, k loops of DIRECT27
  else { ox = c_direct7[k][0]; oy = c_direct7[k][1]; oz = c_direct7[k][2]; }
}

__device__ inline uint64_t slot_key2(const uint4& s) { return ((uint64_t)s.y << 32) | s.x; }

// voxel index of cell (vx,vy,vz) or -1; (cbx..,slot,vox_base) cache the last brick
__device__ inline int voxel_lookup(const TargetView& tg, int vx, int vy, int vz, int& cbx, int& cby, int& cbz, uint32_t& slot, uint32_t& vox_base) {
  const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
  if (bx != cbx || by != cby || bz != cbz) {
    const uint64_t key = pack_brick(bx, by, bz);
    uint32_t h = hash_coord(bx, by, bz) & tg.mask;
    slot = ~0u;
    for (;;) {
      const uint4 s = gload4u(&tg.bricks[h]);
      const uint64_t sk = slot_key2(s);
      if (sk == key) { slot = h; vox_base = s.z; break; }
      if (sk == kEmptyKey) break;
      h = (h + 1) & tg.mask;
    }
    cbx = bx; cby = by; cbz = bz;
  }
  if (slot == ~0u) return -1;
  const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
  const uint32_t m = gload_u(&tg.bmask[(size_t)slot * 16 + w]);
  if (!((m >> bit) & 1u)) return -1;
  return (int)(vox_base + gload_u16(&tg.bpref[(size_t)slot * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u)));
}

__device__ inline void load_gvox(const GaussVoxel* g, float (&mean)[3], float (&C)[9], int& n) {
  const float4 a = gload4(reinterpret_cast<const float4*>(g)), b = gload4(reinterpret_cast<const float4*>(g) + 1), c = gload4(reinterpret_cast<const float4*>(g) + 2);
  mean[0] = a.x; mean[1] = a.y; mean[2] = a.z;
  n = __float_as_int(a.w);
  C[0] = b.x; C[1] = b.y; C[2] = b.z; C[3] = b.y; C[4] = b.w; C[5] = c.x; C[6] = b.z; C[7] = c.x; C[8] = c.y;
}

// Matrix3f::inverse(): Eigen's fixed-size 3x3 inverse (dev_linalg.h: cofactors of column 0, det along that column)
__device__ inline void inv3f(const float (&m)[9], float (&inv)[9]) { inv3<float>(m, inv); }

__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------
// k_ndt: TRIAL = false -> correspondences at x0 + H, b, cost (linearize)
//        TRIAL = true  -> cost of the remembered correspondences at xi (compute_error)
// grid = (blocks_per_pair, npairs), block = 256, kp.lin_points_per_block elements per workgroup
// ---------------------------------------------------------------------------
template <int KIND, bool TRIAL>   // KIND 0: NDT P2D, 1: NDT D2D, 2: VGICP of the CUDA core (compute_derivatives.cu:49-92)
__global__ void __launch_bounds__(256, 3) k_ndt(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp) {
  const int pair = PCM_PAIR_OF(kp, blockIdx.y);
  const int mode = states[pair].mode;
  if (mode != (TRIAL ? MODE_TRIAL : MODE_LINEARIZE)) return;
  constexpr bool D2D = KIND == 1, VGC = KIND == 2;
  const PairDesc d = descs[pair];
  const uint32_t per = (uint32_t)(TRIAL ? kp.points_per_block : kp.lin_points_per_block);
  const uint32_t begin = blockIdx.x * per;
  if (begin >= d.src.num_points) return;
  uint32_t end = begin + per;
  end = end < d.src.num_points ? end : d.src.num_points;
  const TargetView tg = d.tgt;
  const int nO = kp.nb_range > 0 ? (2 * kp.nb_range + 1) * (2 * kp.nb_range + 1) * (2 * kp.nb_range + 1) : kp.num_neighbors;
  const double* T = TRIAL ? states[pair].xi : states[pair].x0;
  float R[9], t[3], Re[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      R[i * 3 + j] = (float)T[i * 4 + j];
      Re[i * 3 + j] = (float)states[pair].x0[i * 4 + j];   // linearized_x (ndt_cuda.cu:149): x0 is unchanged during the trial passes
    }
    t[i] = (float)T[i * 4 + 3];
  }

  double acc[kNumSums];
#pragma unroll
  for (int j = 0; j < kNumSums; j++) acc[j] = 0.0;

  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    float pa[3], CA[9];
    if (D2D) {
      int na;
      load_gvox(d.src.gvox + i, pa, CA, na);
    } else if (VGC) {   // source element = point i of the brick-major copy, its 9-float covariance in the 48-byte slot
      const float4 p = gload4(d.src.pts + i);
      pa[0] = p.x; pa[1] = p.y; pa[2] = p.z;
      const float* cf = reinterpret_cast<const float*>(d.src_cov + (size_t)i * 6);
#pragma unroll
      for (int a = 0; a < 9; a++) CA[a] = *(const PCM_GLOBAL float*)(cf + a);
    } else {
      const float4 p = gload4(d.src.pts + i);
      pa[0] = p.x; pa[1] = p.y; pa[2] = p.z;
    }
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) q[a] = (R[a * 3 + 0] * pa[0] + R[a * 3 + 1] * pa[1]) + R[a * 3 + 2] * pa[2] + t[a];
    int cx = 0, cy = 0, cz = 0;
    bool inrange = true;
    if (!TRIAL) {
      // calc_voxel_coord: floor(x / resolution - 0.5)   vector3_hash.cuh:35-38
      const float fx = floorf(q[0] / tg.res - 0.5f), fy = floorf(q[1] / tg.res - 0.5f), fz = floorf(q[2] / tg.res - 0.5f);
      const float lim = (float)(kCoordBias - 32);
      inrange = fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim;
      if (inrange) { cx = (int)fx; cy = (int)fy; cz = (int)fz; }
    }
    float RCR[9];
    if (D2D || VGC) {   // RCR = R_eval cov_A R_eval^T   ndt_compute_derivatives.cu:145 | compute_derivatives.cu:75
      float RC[9];
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) RC[a * 3 + b] = (Re[a * 3 + 0] * CA[0 * 3 + b] + Re[a * 3 + 1] * CA[1 * 3 + b]) + Re[a * 3 + 2] * CA[2 * 3 + b];
      }
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) RCR[a * 3 + b] = (RC[a * 3 + 0] * Re[b * 3 + 0] + RC[a * 3 + 1] * Re[b * 3 + 1]) + RC[a * 3 + 2] * Re[b * 3 + 2];
      }
    }
    int cbx = 0x7fffffff, cby = 0, cbz = 0;
    uint32_t slot = ~0u, vox_base = 0;
    // a target that is registered against again keeps, for every voxel a query can fall into, the row of its nO neighbour voxels
    // (neighbour_lists.hip): one probe of that index instead of nO cell look-ups one after the other
    const bool rows = !TRIAL && d.nl.pts != nullptr && kp.nb_range == 0;
    const int32_t* row = nullptr;
    if (rows && inrange) {
      int rbx = 0x7fffffff, rby = 0, rbz = 0;
      uint32_t rslot = ~0u, rbase = 0;
      const int lr = voxel_lookup(d.nl, cx, cy, cz, rbx, rby, rbz, rslot, rbase);
      if (lr >= 0) row = reinterpret_cast<const int32_t*>(d.nl.pts) + (size_t)lr * nO;
    }
    for (int k = 0; k < nO; k++) {
      int v;
      if (TRIAL) {
        v = *(const PCM_GLOBAL int32_t*)(d.corr + (size_t)i * nO + k);
      } else if (rows) {
        v = row ? *(const PCM_GLOBAL int32_t*)(row + k) : -1;
        *(PCM_GLOBAL int32_t*)(d.corr + (size_t)i * nO + k) = v;
      } else {
        int ox, oy, oz;
        bool listed = true;
        if (kp.nb_range > 0) listed = ndt_offset_radius(kp.nb_range, kp.nb_radius, k, ox, oy, oz);
        else ndt_offset(nO, k, ox, oy, oz);
        v = (inrange && listed) ? voxel_lookup(tg, cx + ox, cy + oy, cz + oz, cbx, cby, cbz, slot, vox_base) : -1;
        *(PCM_GLOBAL int32_t*)(d.corr + (size_t)i * nO + k) = v;
      }
      if (v < 0) continue;
      float mb[3], C[9];
      int nb;
      if (VGC) {
        const VgcVoxel* cv = d.cvox + v;
        const float4 a0 = gload4(reinterpret_cast<const float4*>(cv)), a1 = gload4(reinterpret_cast<const float4*>(cv) + 1), a2 = gload4(reinterpret_cast<const float4*>(cv) + 2);
        mb[0] = a0.x; mb[1] = a0.y; mb[2] = a0.z; nb = __float_as_int(a0.w);
        C[0] = a1.x; C[1] = a1.y; C[2] = a1.z; C[3] = a1.w; C[4] = a2.x; C[5] = a2.y; C[6] = a2.z; C[7] = a2.w;
        C[8] = *(const PCM_GLOBAL float*)&cv->cov[8];
        if (nb <= 0) continue;   // compute_derivatives.cu:62-64
      } else {
        load_gvox(tg.gvox + v, mb, C, nb);
        if (nb <= 6) continue;   // ndt_compute_derivatives.cu:61,132
      }
      if (D2D || VGC) {
#pragma unroll
        for (int a = 0; a < 9; a++) C[a] += RCR[a];
      }
      float M[9];
      inv3f(C, M);
      float e[3];
#pragma unroll
      for (int a = 0; a < 3; a++) e[a] = mb[a] - q[a];
      const float en = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
      const float ksq = tg.res * tg.res;
      const float w = VGC ? sqrtf((float)nb) : ksq / (ksq + en * en);   // cauchy(resolution, |e|)  :15-18,78,150 | sqrtf(num_points)  compute_derivatives.cu:78
      float Me[3];
#pragma unroll
      for (int a = 0; a < 3; a++) Me[a] = (M[a * 3 + 0] * e[0] + M[a * 3 + 1] * e[1]) + M[a * 3 + 2] * e[2];
      const float err = w * ((e[0] * Me[0] + e[1] * Me[1]) + e[2] * Me[2]);
      acc[27] += (double)err;
      acc[28] += 1.0;
      if (TRIAL) continue;
      // J = [skew(q), -I];  H = w J^T M J ;  b = w J^T M e
      const float J[3][6] = {{0.f, -q[2], q[1], -1.f, 0.f, 0.f}, {q[2], 0.f, -q[0], 0.f, -1.f, 0.f}, {-q[1], q[0], 0.f, 0.f, 0.f, -1.f}};
      float JtM[6][3];
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) JtM[r][c] = (w * J[0][r] * M[0 * 3 + c] + w * J[1][r] * M[1 * 3 + c]) + w * J[2][r] * M[2 * 3 + c];
      }
      int tt = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int c = r; c < 6; c++) { acc[tt] += (double)((JtM[c][0] * J[0][r] + JtM[c][1] * J[1][r]) + JtM[c][2] * J[2][r]); tt++; }   // entry (c, r) of the float product: the LOWER triangle is what the LDLT of the reference reads
      }
#pragma unroll
      for (int r = 0; r < 6; r++) acc[21 + r] += (double)((JtM[r][0] * e[0] + JtM[r][1] * e[1]) + JtM[r][2] * e[2]);
    }
  }

  __shared__ double s_part[4][kPartialStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < kNumSums; j++) {
    const double v = wave_sum_d(acc[j]);
    if (lane == 0) s_part[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    const double v = ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) + s_part[3][threadIdx.x];
    gstore_d(d.partials + (size_t)blockIdx.x * kPartialStride + threadIdx.x, v);
  }
}

void launch_ndt(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, int kind, bool trial) {
  dim3 grid((unsigned)(trial ? kp.blocks_per_pair : kp.tiles_per_pair), (unsigned)npairs);
  if (kind == 1) {
    if (trial) k_ndt<1, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
    else k_ndt<1, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  } else if (kind == 2) {
    if (trial) k_ndt<2, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
    else k_ndt<2, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  } else {
    if (trial) k_ndt<0, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
    else k_ndt<0, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  }
}

}  // namespace pcm

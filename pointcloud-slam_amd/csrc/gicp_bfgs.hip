// gicp_bfgs.hip -- objective / gradient of pclomp's GICP-BFGS on the device, gfx950.
//
// Replaces OptimizationFunctorWithIndices::operator() / df / fdf of jueying_slam's GICP_OMP option
// (/root/reference/src/pointcloud_match/ndt_omp/include/pclomp/gicp_omp_impl.hpp:246-365): the BFGS evaluates the
// functor tens of times per outer iteration over the SAME correspondence set, so the set is packed once
// (k_bfgs_pack: p_src, p_tgt and the 3x3 block of the source point's Mahalanobis matrix -> 64 bytes in four float4 planes)
// and every evaluation is one streaming pass over the records: 64 B per correspondence, HBM-bound, 14 sums
// in double (f of the float residual path, f of the double residual path, 3 x g_t, 9 x R).
// Sums are added in a fixed order (lane stride, 16-lane groups through LDS, workgroups), so a result is
// reproducible; the reference's order depends on the OpenMP schedule.
#include "pcm_device.h"
#include "pcm_host.h"

namespace pcm {

namespace {

constexpr int kBfgsSums = 14;
constexpr int kBfgsMaxBlocks = kGicpBfgsMaxBlocks;

struct BfgsXf { float T[12]; float B[12]; };   // rows 0..2 of transformation_matrix and base_transformation_

// record (4 float4, one per plane): p.xyz, q.xyz, M row-major 3x3, pad
__global__ void __launch_bounds__(256) k_bfgs_pack(const char* __restrict__ src, const char* __restrict__ tgt, size_t stride, const int* __restrict__ idx_src,
                                                   const int* __restrict__ idx_tgt, const float* __restrict__ maha, uint32_t m, float4* __restrict__ rec) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m) return;
  const int is = idx_src[i], it = idx_tgt[i];
  const float* p = reinterpret_cast<const float*>(src + (size_t)is * stride);
  const float* q = reinterpret_cast<const float*>(tgt + (size_t)it * stride);
  const float* M = maha + (size_t)is * 16;   // column-major Matrix4f: M(a,b) = M[b * 4 + a]
  rec[i] = make_float4(p[0], p[1], p[2], q[0]);                       // four planes of m float4: every load of a wave is one contiguous KB
  rec[(size_t)m + i] = make_float4(q[1], q[2], M[0], M[4]);
  rec[2 * (size_t)m + i] = make_float4(M[8], M[1], M[5], M[9]);
  rec[3 * (size_t)m + i] = make_float4(M[2], M[6], M[10], 0.f);
}

struct BfgsRec { float4 r0, r1, r2, r3; };

__device__ inline void bfgs_accumulate(const BfgsRec& R, const BfgsXf& X, double (&s)[kBfgsSums]) {
  const float p[3] = {R.r0.x, R.r0.y, R.r0.z}, q[3] = {R.r0.w, R.r1.x, R.r1.y};
  const float M[9] = {R.r1.z, R.r1.w, R.r2.x, R.r2.y, R.r2.z, R.r2.w, R.r3.x, R.r3.y, R.r3.z};
  float res[3], pb[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float pp = ((X.T[a * 4 + 0] * p[0] + X.T[a * 4 + 1] * p[1]) + X.T[a * 4 + 2] * p[2]) + X.T[a * 4 + 3] * 1.f;
    res[a] = pp - q[a];
    pb[a] = ((X.B[a * 4 + 0] * p[0] + X.B[a * 4 + 1] * p[1]) + X.B[a * 4 + 2] * p[2]) + X.B[a * 4 + 3] * 1.f;
  }
  float Mr[3];
#pragma unroll
  for (int a = 0; a < 3; a++) Mr[a] = (M[a * 3 + 0] * res[0] + M[a * 3 + 1] * res[1]) + M[a * 3 + 2] * res[2];
  s[0] += (double)((res[0] * Mr[0] + res[1] * Mr[1]) + res[2] * Mr[2]);          // operator():  res . (M res), float
  const double rd[3] = {(double)res[0], (double)res[1], (double)res[2]};
  double t[3];
#pragma unroll
  for (int a = 0; a < 3; a++) t[a] = ((double)M[a * 3 + 0] * rd[0] + (double)M[a * 3 + 1] * rd[1]) + (double)M[a * 3 + 2] * rd[2];
  s[1] += (rd[0] * t[0] + rd[1] * t[1]) + rd[2] * t[2];                            // fdf():  res^T (M res), double
#pragma unroll
  for (int a = 0; a < 3; a++) {
    s[2 + a] += t[a];
#pragma unroll
    for (int b = 0; b < 3; b++) s[5 + a * 3 + b] += (double)pb[a] * t[b];
  }
}

__global__ void __launch_bounds__(256) k_bfgs_fdf(const float4* __restrict__ rec, uint32_t m, BfgsXf X, double* __restrict__ partials) {
  double s[kBfgsSums];
#pragma unroll
  for (int a = 0; a < kBfgsSums; a++) s[a] = 0.0;
  const uint64_t S = (uint64_t)gridDim.x * 256u;
  // four records per trip, all sixteen loads issued before the first use (the pass is a latency chain otherwise)
  for (uint64_t i0 = blockIdx.x * 256u + threadIdx.x; i0 < m; i0 += 4 * S) {
    BfgsRec R[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t i = i0 + u * S;
      const size_t ic = i < m ? (size_t)i : (size_t)i0;
      R[u].r0 = gload4(rec + ic); R[u].r1 = gload4(rec + (size_t)m + ic); R[u].r2 = gload4(rec + 2 * (size_t)m + ic); R[u].r3 = gload4(rec + 3 * (size_t)m + ic);
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i0 + u * S < m) bfgs_accumulate(R[u], X, s);
  }
  // workgroup sum in a fixed order through LDS: 256 lanes -> 16 groups of 16 -> 1
  __shared__ double s_lane[kBfgsSums][256];
  __shared__ double s_part[kBfgsSums][16];
#pragma unroll
  for (int a = 0; a < kBfgsSums; a++) s_lane[a][threadIdx.x] = s[a];
  __syncthreads();
  if (threadIdx.x < kBfgsSums * 16) {
    const int a = threadIdx.x >> 4, part = threadIdx.x & 15;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) v += s_lane[a][part * 16 + k];
    s_part[a][part] = v;
  }
  __syncthreads();
  if (threadIdx.x < kBfgsSums) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) v += s_part[threadIdx.x][k];
    gstore_d(partials + (size_t)blockIdx.x * kBfgsSums + threadIdx.x, v);
  }
}

// fixed-order sum of the workgroup rows: 64 row groups x 16 columns, then the 64 group sums in order
__global__ void __launch_bounds__(1024) k_bfgs_finish(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  __shared__ double s_grp[64][16];
  const int t = threadIdx.x & 15, r = threadIdx.x >> 4;
  double v = 0.0;
  if (t < kBfgsSums)
    for (int b = r; b < nblocks; b += 64) v += gload_d(partials + (size_t)b * kBfgsSums + t);
  s_grp[r][t] = v;
  __syncthreads();
  if (threadIdx.x < kBfgsSums) {
    double a = 0.0;
    for (int k = 0; k < 64; k++) a += s_grp[k][threadIdx.x];
    out[threadIdx.x] = a;
  }
}

}  // namespace

size_t gicp_bfgs_scratch_bytes(size_t m) { return 64 * m + 256 + sizeof(double) * kBfgsSums * (kBfgsMaxBlocks + 1); }

int gicp_bfgs_blocks(size_t m) {
  const size_t b = (m + 255) / 256;
  return (int)(b < (size_t)kBfgsMaxBlocks ? (b ? b : 1) : (size_t)kBfgsMaxBlocks);
}

int gicp_bfgs_pack_device(hipStream_t stream, const void* d_src, const void* d_tgt, size_t stride, const int* d_idx_src, const int* d_idx_tgt, const float* d_maha, size_t m,
                          void* d_records, std::string* err) {
  if (m == 0) return PCM_OK;
  k_bfgs_pack<<<(unsigned)((m + 255) / 256), 256, 0, stream>>>(static_cast<const char*>(d_src), static_cast<const char*>(d_tgt), stride, d_idx_src, d_idx_tgt, d_maha, (uint32_t)m,
                                                               static_cast<float4*>(d_records));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { if (err) *err = hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

// sums[14] (device): f32-path f, f64-path f, g_t[3], R[9] (row-major), all un-normalised
int gicp_bfgs_fdf_device(hipStream_t stream, const void* d_records, size_t m, const float T[16], const float base[16], double* d_partials, double* d_sums, std::string* err) {
  BfgsXf X;
  for (int a = 0; a < 12; a++) { X.T[a] = T[a]; X.B[a] = base[a]; }
  const int nb = gicp_bfgs_blocks(m);
  k_bfgs_fdf<<<nb, 256, 0, stream>>>(static_cast<const float4*>(d_records), (uint32_t)m, X, d_partials);
  k_bfgs_finish<<<1, 1024, 0, stream>>>(d_partials, nb, d_sums);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { if (err) *err = hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

}  // namespace pcm

// gicp_bfgs.hip -- objective / gradient of pclomp's GICP-BFGS on the device, gfx950.
//
// Replaces OptimizationFunctorWithIndices::operator() / df / fdf of jueying_slam's GICP_OMP option
// (/root/reference/src/pointcloud_match/ndt_omp/include/pclomp/gicp_omp_impl.hpp:246-365): the BFGS evaluates the
// functor tens of times per outer iteration over the SAME correspondence set, so the set is packed once
// (k_bfgs_pack: p_src, p_tgt and the 3x3 block of the source point's Mahalanobis matrix -> one 64-byte record)
// and every evaluation is one streaming pass over the records: 64 B per correspondence, HBM-bound, 14 sums
// in double (f of the float residual path, f of the double residual path, 3 x g_t, 9 x R).
// Sums are added in a fixed order (lane stride, wave shuffle tree, waves, workgroups), so a result is
// reproducible; the reference's order depends on the OpenMP schedule.
#include "pcm_device.h"
#include "pcm_host.h"

namespace pcm {

namespace {

constexpr int kBfgsSums = 14;
constexpr int kBfgsMaxBlocks = 1024;

struct BfgsXf { float T[12]; float B[12]; };   // rows 0..2 of transformation_matrix and base_transformation_

// record: p.xyz, q.xyz, M row-major 3x3, pad
__global__ void __launch_bounds__(256) k_bfgs_pack(const char* __restrict__ src, const char* __restrict__ tgt, size_t stride, const int* __restrict__ idx_src,
                                                   const int* __restrict__ idx_tgt, const float* __restrict__ maha, uint32_t m, float4* __restrict__ rec) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= m) return;
  const int is = idx_src[i], it = idx_tgt[i];
  const float* p = reinterpret_cast<const float*>(src + (size_t)is * stride);
  const float* q = reinterpret_cast<const float*>(tgt + (size_t)it * stride);
  const float* M = maha + (size_t)is * 16;   // column-major Matrix4f: M(a,b) = M[b * 4 + a]
  rec[4 * (size_t)i + 0] = make_float4(p[0], p[1], p[2], q[0]);
  rec[4 * (size_t)i + 1] = make_float4(q[1], q[2], M[0], M[4]);
  rec[4 * (size_t)i + 2] = make_float4(M[8], M[1], M[5], M[9]);
  rec[4 * (size_t)i + 3] = make_float4(M[2], M[6], M[10], 0.f);
}

__device__ inline double wsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(256) k_bfgs_fdf(const float4* __restrict__ rec, uint32_t m, BfgsXf X, double* __restrict__ partials) {
  double s[kBfgsSums];
#pragma unroll
  for (int a = 0; a < kBfgsSums; a++) s[a] = 0.0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < m; i += gridDim.x * 256u) {
    const float4 r0 = gload4(rec + 4 * (size_t)i), r1 = gload4(rec + 4 * (size_t)i + 1), r2 = gload4(rec + 4 * (size_t)i + 2), r3 = gload4(rec + 4 * (size_t)i + 3);
    const float p[3] = {r0.x, r0.y, r0.z}, q[3] = {r0.w, r1.x, r1.y};
    const float M[9] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z};
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
  __shared__ double s_w[4][kBfgsSums];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < kBfgsSums; a++) {
    const double v = wsum(s[a]);
    if (lane == 0) s_w[wave][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < kBfgsSums) gstore_d(partials + (size_t)blockIdx.x * kBfgsSums + threadIdx.x, ((s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + s_w[2][threadIdx.x]) + s_w[3][threadIdx.x]);
}

__global__ void __launch_bounds__(64) k_bfgs_finish(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  if (threadIdx.x >= kBfgsSums) return;
  double v = 0.0;
  for (int b = 0; b < nblocks; b++) v += gload_d(partials + (size_t)b * kBfgsSums + threadIdx.x);
  out[threadIdx.x] = v;
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
  k_bfgs_finish<<<1, 64, 0, stream>>>(d_partials, nb, d_sums);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { if (err) *err = hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

}  // namespace pcm

// preprocess.hip -- scan pre-processing on the device, gfx950.
//
// Replaces ImuProcess::UndistortPcl's backward propagation
// (/root/reference/src/jueying_lio/include/imu_processing.hpp:245-285): every scan point is compensated from
// its own sampling time into the frame-end pose.  The reference walks IMU poses and points backwards in one
// serial loop; a point's segment only depends on its time stamp (the last IMU pose before it), so here every
// point finds its segment by binary search and the scan is one launch, in place, so it never returns to the
// host between the driver callback and registration.  Double arithmetic, as in the reference.
// The reference's loop visits the FIRST scan point again in every earlier segment once it has reached it;
// lane 0 reproduces that chain.
#include "pcm_device.h"
#include "pcm_host.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace pcm {

namespace {

__device__ inline void qrot3(const double (&q)[4], const double (&v)[3], double (&r)[3]) {   // Eigen quaternion (x,y,z,w) * vector
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
#pragma unroll
  for (int a = 0; a < 3; a++) r[a] = v[a] + q[3] * uv[a] + c[a];
}

// one compensation step of point P (sampled at t) with the segment (head, tail)   :259-276
__device__ inline void compensate(const pcm_imu_pose& head, const pcm_imu_pose& tail, const LioStateD& s, double t, float (&P)[3]) {
  const double dt = t - head.offset_time;
  const double* w = tail.gyr;
  const double n = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
  double E[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
  if (n > 0.0000001) {   // Exp(ang_vel, dt)  so3_math.h:31-49
    const double ax[3] = {w[0] / n, w[1] / n, w[2] / n};
    const double K[9] = {0.0, -ax[2], ax[1], ax[2], 0.0, -ax[0], -ax[1], ax[0], 0.0};
    const double ang = n * dt, sn = sin(ang), c1 = 1.0 - cos(ang);
    double cK[9], KK[9];
#pragma unroll
    for (int i = 0; i < 9; i++) cK[i] = c1 * K[i];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int j = 0; j < 3; j++) KK[i * 3 + j] = (cK[i * 3 + 0] * K[0 * 3 + j] + cK[i * 3 + 1] * K[1 * 3 + j]) + cK[i * 3 + 2] * K[2 * 3 + j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) E[i] = (E[i] + sn * K[i]) + KK[i];
  }
  double Ri[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) Ri[i * 3 + j] = (head.rot[i * 3 + 0] * E[0 * 3 + j] + head.rot[i * 3 + 1] * E[1 * 3 + j]) + head.rot[i * 3 + 2] * E[2 * 3 + j];
  }
  const double Pi[3] = {(double)P[0], (double)P[1], (double)P[2]};
  const double rot_c[4] = {-s.rot[0], -s.rot[1], -s.rot[2], s.rot[3]}, off_c[4] = {-s.off_R[0], -s.off_R[1], -s.off_R[2], s.off_R[3]};
  const double offR[4] = {s.off_R[0], s.off_R[1], s.off_R[2], s.off_R[3]};
  double Tei[3], a[3], b[3], c[3], d[3], e[3];
#pragma unroll
  for (int k = 0; k < 3; k++) Tei[k] = ((head.pos[k] + head.vel[k] * dt) + ((0.5 * tail.acc[k]) * dt) * dt) - s.pos[k];
  qrot3(offR, Pi, a);
#pragma unroll
  for (int k = 0; k < 3; k++) a[k] += s.off_T[k];
#pragma unroll
  for (int k = 0; k < 3; k++) b[k] = ((Ri[k * 3 + 0] * a[0] + Ri[k * 3 + 1] * a[1]) + Ri[k * 3 + 2] * a[2]) + Tei[k];
  qrot3(rot_c, b, c);
#pragma unroll
  for (int k = 0; k < 3; k++) d[k] = c[k] - s.off_T[k];
  qrot3(off_c, d, e);
  P[0] = (float)e[0]; P[1] = (float)e[1]; P[2] = (float)e[2];
}

__global__ void __launch_bounds__(256) k_undistort(char* __restrict__ base, size_t stride, size_t time_off, uint32_t n, const pcm_imu_pose* __restrict__ poses, int npose, LioStateD s) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* rec = reinterpret_cast<float*>(base + (size_t)i * stride);
  const double t = (double)*reinterpret_cast<const float*>(base + (size_t)i * stride + time_off) / (double)1000;
  // head = the last pose (not counting the final one) whose offset_time is below t
  int lo = 0, hi = npose - 2, h = -1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (poses[mid].offset_time < t) { h = mid; lo = mid + 1; } else hi = mid - 1;
  }
  if (h < 0) return;   // at or before the first IMU pose: the reference leaves the point as it is
  float P[3] = {rec[0], rec[1], rec[2]};
  compensate(poses[h], poses[h + 1], s, t, P);
  if (i == 0) {        // the reference re-visits the first point in every earlier segment (:248,278-280)
    for (int k = h - 1; k >= 0; k--) compensate(poses[k], poses[k + 1], s, t, P);
  }
  rec[0] = P[0]; rec[1] = P[1]; rec[2] = P[2];
}

}  // namespace

int undistort_device(hipStream_t stream, void* d_points, size_t n, size_t stride, size_t time_off, const pcm_imu_pose* d_poses, int npose, const LioStateD& s, std::string* err) {
  if (n == 0 || npose < 2) return PCM_OK;
  k_undistort<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(static_cast<char*>(d_points), stride, time_off, (uint32_t)n, d_poses, npose, s);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { *err = std::string("k_undistort: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}


// ---------------------------------------------------------------------------
// pcl::VoxelGrid down-sampling of the scan (jueying_lio/src/laser_mapping.cc:323-328; pcl/filters/impl/voxel_grid.hpp):
// cell = floor(p * inverse_leaf_size) - min_b, linear index ijk . (1, dx, dx dy), one centroid per occupied cell in
// increasing index order, every float field of the record averaged.  PCL sorts (index, point) pairs with std::sort and
// sums in float; here: radix sort (stable), one wave per cell, double sums (64 interleaved partial sums + a fixed tree).
// ---------------------------------------------------------------------------
namespace {

__device__ inline unsigned int f2ord(float f) { const unsigned int u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }   // order-preserving
inline float ord2f(unsigned int o) { const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; std::memcpy(&f, &u, 4); return f; }

__global__ void k_vg_minmax(const char* __restrict__ base, size_t stride, uint32_t n, unsigned int* __restrict__ mm /* min xyz, max xyz (ordered ints), count */) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u}, cnt = 0u;
  if (i < n) {
    const float* p = reinterpret_cast<const float*>(base + (size_t)i * stride);
    if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
      for (int a = 0; a < 3; a++) { lo[a] = hi[a] = f2ord(p[a]); }
      cnt = 1u;
    }
  }
  for (int off = 32; off >= 1; off >>= 1) {   // one atomic per wave, not per point
    for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], (unsigned int)__shfl_xor((int)lo[a], off, 64)); hi[a] = max(hi[a], (unsigned int)__shfl_xor((int)hi[a], off, 64)); }
    cnt += (unsigned int)__shfl_xor((int)cnt, off, 64);
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    for (int a = 0; a < 3; a++) { atomicMin(&mm[a], lo[a]); atomicMax(&mm[3 + a], hi[a]); }
    atomicAdd(&mm[6], cnt);
  }
}

__global__ void k_vg_keys(const char* __restrict__ base, size_t stride, uint32_t n, float inv, int mb0, int mb1, int mb2, long long div0, long long div01, uint64_t* __restrict__ keys,
                          uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = reinterpret_cast<const float*>(base + (size_t)i * stride);
  uint64_t key = 1ull << 32;   // non-finite points sort behind every cell
  if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
    const long long i0 = (long long)(floorf(p[0] * inv) - (float)mb0), i1 = (long long)(floorf(p[1] * inv) - (float)mb1), i2 = (long long)(floorf(p[2] * inv) - (float)mb2);
    key = (uint64_t)(i0 + i1 * div0 + i2 * div01);
  }
  keys[i] = key;
  vals[i] = i;
}

__global__ void k_vg_heads(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ head) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  head[i] = (k < (1ull << 32) && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

// position of every cell's first element in the sorted order
__global__ void k_vg_head_pos(const uint32_t* __restrict__ head, const uint32_t* __restrict__ slot, uint32_t n, uint32_t* __restrict__ pos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i]) pos[slot[i]] = i;
}

// one wave per cell: lane l sums elements l, l + 64, ... of the cell's run in double, then a fixed shuffle tree
// (deterministic; a dense leaf near the sensor holds thousands of points, which one lane alone would walk serially)
__global__ void __launch_bounds__(256) k_vg_average(const char* __restrict__ base, size_t stride, int nfields, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pos,
                                                    uint32_t ncells, uint32_t nvalid, float* __restrict__ out) {
  const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cell >= ncells) return;
  const uint32_t b = pos[cell], e = cell + 1 < ncells ? pos[cell + 1] : nvalid;
  double acc[16];
  for (int f = 0; f < 16; f++) acc[f] = 0.0;
  for (uint32_t j = b + lane; j < e; j += 64) {
    const float* p = reinterpret_cast<const float*>(base + (size_t)vals[j] * stride);
    for (int f = 0; f < nfields; f++) acc[f] += (double)p[f];
  }
  for (int f = 0; f < nfields; f++) {
    double v = acc[f];
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) out[(size_t)cell * nfields + f] = (float)(v / (double)(e - b));
  }
}

}  // namespace

// scratch bytes voxel_downsample_device needs for n points (keys, values, flags, rocPRIM temporaries)
size_t voxel_downsample_scratch_bytes(size_t n) {
  size_t t1 = 0, t2 = 0;
  uint64_t* k = nullptr; uint32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t1, k, k, v, v, n, 0, 33, nullptr);
  (void)rocprim::exclusive_scan(nullptr, t2, v, v, 0u, n, rocprim::plus<uint32_t>(), nullptr);
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return up(64) + 2 * up(8 * n) + 4 * up(4 * n) + up(t1) + up(t2);
}

// d_in: n records of `stride` bytes (nfields = stride / 4 floats, x y z first); d_out: room for n records.  *n_out receives
// the count.  `scratch`: voxel_downsample_scratch_bytes(n) bytes of device memory owned by the caller (no allocation here).
int voxel_downsample_device(hipStream_t stream, const void* d_in, size_t n, size_t stride, float leaf, float* d_out, size_t* n_out, void* scratch, std::string* err) {
  *n_out = 0;
  if (n == 0) return PCM_OK;
  const int nfields = (int)(stride / 4);
  if (nfields < 3 || nfields > 16 || (stride % 4) != 0) { *err = "records must be 3..16 floats"; return PCM_ERR_INVALID_ARGUMENT; }
  if (!(leaf > 0.f)) { *err = "leaf size must be > 0"; return PCM_ERR_INVALID_ARGUMENT; }
  const float inv = 1.0f / leaf;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t tmp_bytes = 0, tmp2_bytes = 0;
  {
    uint64_t* k = nullptr; uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k, k, v, v, n, 0, 33, stream);
    (void)rocprim::exclusive_scan(nullptr, tmp2_bytes, v, v, 0u, n, rocprim::plus<uint32_t>(), stream);
  }
  char* cur = static_cast<char*>(scratch);
  unsigned int* d_mm = reinterpret_cast<unsigned int*>(cur); cur += up(64);
  uint64_t* keys = reinterpret_cast<uint64_t*>(cur); cur += up(8 * n);
  uint64_t* keys_s = reinterpret_cast<uint64_t*>(cur); cur += up(8 * n);
  uint32_t* vals = reinterpret_cast<uint32_t*>(cur); cur += up(4 * n);
  uint32_t* vals_s = reinterpret_cast<uint32_t*>(cur); cur += up(4 * n);
  uint32_t* head = reinterpret_cast<uint32_t*>(cur); cur += up(4 * n);
  uint32_t* slot = reinterpret_cast<uint32_t*>(cur); cur += up(4 * n);
  void* tmp = cur; cur += up(tmp_bytes);
  void* tmp2 = cur;
  int rc = PCM_OK;
  const unsigned nb = (unsigned)((n + 255) / 256);
  const char* base = static_cast<const char*>(d_in);
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); return PCM_ERR_HIP; } \
  } while (0)
  unsigned int h_mm[7] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u};
  CK(hipMemcpyAsync(d_mm, h_mm, sizeof(h_mm), hipMemcpyHostToDevice, stream));
  k_vg_minmax<<<nb, 256, 0, stream>>>(base, stride, (uint32_t)n, d_mm);
  CK(hipMemcpyAsync(h_mm, d_mm, sizeof(h_mm), hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  if (h_mm[6] == 0) return PCM_OK;   // no finite point
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = ord2f(h_mm[a]); mx[a] = ord2f(h_mm[3 + a]); }
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > 2147483647ll) { *err = "leaf size too small for the extent of the cloud (index overflow)"; return PCM_ERR_OUT_OF_RANGE; }
  int min_b[3], max_b[3];
  for (int a = 0; a < 3; a++) { min_b[a] = (int)floorf(mn[a] * inv); max_b[a] = (int)floorf(mx[a] * inv); }
  const long long div0 = (long long)max_b[0] - min_b[0] + 1, div1 = (long long)max_b[1] - min_b[1] + 1;
  k_vg_keys<<<nb, 256, 0, stream>>>(base, stride, (uint32_t)n, inv, min_b[0], min_b[1], min_b[2], div0, div0 * div1, keys, vals);
  CK(hipGetLastError());
  CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_s, vals, vals_s, n, 0, 33, stream));
  k_vg_heads<<<nb, 256, 0, stream>>>(keys_s, (uint32_t)n, head);
  CK(hipGetLastError());
  CK(rocprim::exclusive_scan(tmp2, tmp2_bytes, head, slot, 0u, n, rocprim::plus<uint32_t>(), stream));
  uint32_t last[2];
  CK(hipMemcpyAsync(&last[0], slot + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&last[1], head + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  const uint32_t ncells = last[0] + last[1];
  uint32_t* pos = vals;   // the unsorted value array is free after the sort
  k_vg_head_pos<<<nb, 256, 0, stream>>>(head, slot, (uint32_t)n, pos);
  k_vg_average<<<(ncells + 3) / 4, 256, 0, stream>>>(base, stride, nfields, vals_s, pos, ncells, h_mm[6], d_out);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(stream));
  *n_out = (size_t)ncells;
  return rc;
#undef CK
}

// ---------------------------------------------------------------------------
// PointCloudPreprocess::AviaHandler  (jueying_lio/src/pointcloud_preprocess.cc:44-88): livox CustomMsg points -> PointXYZINormal cloud.
// A raw point is 20 bytes {uint32 offset_time; float x, y, z; uint8 reflectivity, tag, line; pad}.  Point i >= 1 passes when its line and
// tag do and i % point_filter_num == 0 (:58-61), and is kept when it differs from cloud_full_[i - 1]: the previous raw point if THAT one
// passed, else the zero point resize() left (:50) -- the serial reading of the reference's par_unseq loop --, under the reference's own
// operator precedence |dx| > 1e-7 || |dy| > 1e-7 || (|dz| > 1e-7 && r^2 > blind^2)  (:69-74).  Flag pass, exclusive scan, write pass:
// the kept points leave in input order (:82-86), 48-byte records {x, y, z, 1, 0, 0, 0, 0, intensity, curvature, 0, 0}.
// ---------------------------------------------------------------------------
struct LivoxRaw { uint32_t offset_time; float x, y, z; uint8_t reflectivity, tag, line, pad; };
static_assert(sizeof(LivoxRaw) == 20, "livox CustomPoint layout");

__device__ inline bool livox_passes(const LivoxRaw& r, uint32_t i, int num_scans, uint32_t filter_num) {
  return (int)r.line < num_scans && ((r.tag & 0x30) == 0x10 || (r.tag & 0x30) == 0x00) && (i % filter_num) == 0;
}

__global__ void __launch_bounds__(256) k_livox_flags(const LivoxRaw* __restrict__ msg, uint32_t n, int num_scans, uint32_t filter_num, double blind2, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t keep = 0;
  if (i >= 1) {
    const LivoxRaw r = msg[i];
    if (livox_passes(r, i, num_scans, filter_num)) {
      float px = 0.f, py = 0.f, pz = 0.f;
      if (i >= 2) {
        const LivoxRaw rp = msg[i - 1];
        if (livox_passes(rp, i - 1, num_scans, filter_num)) { px = rp.x; py = rp.y; pz = rp.z; }
      }
      const float r2 = r.x * r.x + r.y * r.y + r.z * r.z;
      keep = ((double)fabsf(r.x - px) > 1e-7 || (double)fabsf(r.y - py) > 1e-7 || ((double)fabsf(r.z - pz) > 1e-7 && (double)r2 > blind2)) ? 1u : 0u;
    }
  }
  flag[i] = keep;
}

__global__ void __launch_bounds__(256) k_livox_write(const LivoxRaw* __restrict__ msg, uint32_t n, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const LivoxRaw r = msg[i];
  float4* o = out + 3 * (size_t)pos[i];
  o[0] = make_float4(r.x, r.y, r.z, 1.f);
  o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  o[2] = make_float4((float)r.reflectivity, __fdiv_rn((float)r.offset_time, (float)1000000), 0.f, 0.f);   // intensity; curvature = offset_time / float(1000000), ms
}

size_t livox_filter_scratch_bytes(size_t n) {
  size_t t = 0;
  uint32_t* v = nullptr;
  (void)rocprim::exclusive_scan(nullptr, t, v, v, 0u, n, rocprim::plus<uint32_t>(), nullptr);
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return 2 * up(4 * n) + up(t);
}

int livox_filter_device(hipStream_t stream, const void* d_msg, size_t n, int num_scans, int point_filter_num, double blind, void* d_out, size_t* n_out, void* scratch, std::string* err) {
  *n_out = 0;
  if (n < 2) return PCM_OK;
  if (point_filter_num < 1 || num_scans < 0) { *err = "point_filter_num must be >= 1"; return PCM_ERR_INVALID_ARGUMENT; }
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char* cur = static_cast<char*>(scratch);
  uint32_t* flag = reinterpret_cast<uint32_t*>(cur); cur += up(4 * n);
  uint32_t* pos = reinterpret_cast<uint32_t*>(cur); cur += up(4 * n);
  void* tmp = cur;
  size_t tmp_bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, tmp_bytes, flag, pos, 0u, n, rocprim::plus<uint32_t>(), stream);
  const unsigned nb = (unsigned)((n + 255) / 256);
  const LivoxRaw* msg = static_cast<const LivoxRaw*>(d_msg);
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); return PCM_ERR_HIP; } \
  } while (0)
  k_livox_flags<<<nb, 256, 0, stream>>>(msg, (uint32_t)n, num_scans, (uint32_t)point_filter_num, blind * blind, flag);
  CK(hipGetLastError());
  CK(rocprim::exclusive_scan(tmp, tmp_bytes, flag, pos, 0u, n, rocprim::plus<uint32_t>(), stream));
  uint32_t tails[2] = {0, 0};
  CK(hipMemcpyAsync(&tails[0], flag + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  CK(hipMemcpyAsync(&tails[1], pos + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  CK(hipStreamSynchronize(stream));
  const size_t m = (size_t)tails[0] + tails[1];
  if (m) {
    k_livox_write<<<nb, 256, 0, stream>>>(msg, (uint32_t)n, flag, pos, static_cast<float4*>(d_out));
    CK(hipGetLastError());
  }
  *n_out = m;
  return PCM_OK;
#undef CK
}

}  // namespace pcm

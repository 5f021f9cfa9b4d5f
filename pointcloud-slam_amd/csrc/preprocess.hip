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

}  // namespace pcm

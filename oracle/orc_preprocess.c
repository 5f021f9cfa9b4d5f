/*
 * oracle/orc_preprocess.c -- scan pre-processing of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Restates ImuProcess::UndistortPcl's backward propagation loop
 * (/root/reference/src/jueying_lio/include/imu_processing.hpp:245-285) literally: IMU poses from last to first, scan
 * points from last to first, each point compensated into the frame-end pose with
 *   R_i = R_imu * Exp(angvel, dt)                                   so3_math.h:31-49 (Rodrigues)
 *   T_ei = pos_imu + vel_imu dt + 0.5 acc_imu dt^2 - pos_end
 *   p = off_R^-1 * (rot_end^-1 * (R_i * (off_R * P + off_T) + T_ei) - off_T)        (quaternion rotations, double)
 * including its quirk: the FIRST scan point is visited again by every earlier IMU segment once it has been reached.
 * Points must be sorted by time as the reference sorts them (imu_processing.hpp:177-178).
 */
#include "orc_internal.h"

static void qrot(const double q[4], const double v[3], double r[3]) {   /* Eigen quaternion (x,y,z,w) * vector */
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int a = 0; a < 3; a++) r[a] = v[a] + q[3] * uv[a] + c[a];
}

static void so3_exp_dt(const double w[3], double dt, double R[9]) {   /* Exp(ang_vel, dt)  so3_math.h:31-49 */
  const double n = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
  for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (!(n > 0.0000001)) return;
  const double ax[3] = {w[0] / n, w[1] / n, w[2] / n};
  const double K[9] = {0.0, -ax[2], ax[1], ax[2], 0.0, -ax[0], -ax[1], ax[0], 0.0};
  const double ang = n * dt, s = sin(ang), c1 = 1.0 - cos(ang);
  double cK[9], KK[9];
  for (int i = 0; i < 9; i++) cK[i] = c1 * K[i];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) KK[i * 3 + j] = (cK[i * 3 + 0] * K[0 * 3 + j] + cK[i * 3 + 1] * K[1 * 3 + j]) + cK[i * 3 + 2] * K[2 * 3 + j];
  for (int i = 0; i < 9; i++) R[i] = (R[i] + s * K[i]) + KK[i];
}

void orc_undistort(float *pts, long n, long stride_floats, long time_index, const orc_imu_pose *poses, int npose, const orc_lio_state *st) {
  if (n <= 0 || npose < 2) return;
  const double rot_c[4] = {-st->rot[0], -st->rot[1], -st->rot[2], st->rot[3]};
  const double off_c[4] = {-st->off_R[0], -st->off_R[1], -st->off_R[2], st->off_R[3]};
  long it = n - 1;
  for (int kp = npose - 1; kp != 0; kp--) {
    const orc_imu_pose *head = &poses[kp - 1], *tail = &poses[kp];
    for (; (double)pts[it * stride_floats + time_index] / (double)1000 > head->offset_time; it--) {
      float *P = pts + it * stride_floats;
      const double dt = (double)P[time_index] / (double)1000 - head->offset_time;
      double E[9], Ri[9];
      so3_exp_dt(tail->gyr, dt, E);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ri[i * 3 + j] = (head->rot[i * 3 + 0] * E[0 * 3 + j] + head->rot[i * 3 + 1] * E[1 * 3 + j]) + head->rot[i * 3 + 2] * E[2 * 3 + j];
      const double Pi[3] = {P[0], P[1], P[2]};
      double Tei[3], a[3], b[3], c[3], d[3], e[3];
      for (int k = 0; k < 3; k++) Tei[k] = ((head->pos[k] + head->vel[k] * dt) + ((0.5 * tail->acc[k]) * dt) * dt) - st->pos[k];
      qrot(st->off_R, Pi, a);
      for (int k = 0; k < 3; k++) a[k] += st->off_T[k];
      for (int k = 0; k < 3; k++) b[k] = ((Ri[k * 3 + 0] * a[0] + Ri[k * 3 + 1] * a[1]) + Ri[k * 3 + 2] * a[2]) + Tei[k];
      qrot(rot_c, b, c);
      for (int k = 0; k < 3; k++) d[k] = c[k] - st->off_T[k];
      qrot(off_c, d, e);
      P[0] = (float)e[0]; P[1] = (float)e[1]; P[2] = (float)e[2];
      if (it == 0) break;
    }
  }
}

/*
 * pcl::VoxelGrid<PointT>::applyFilter as LaserMapping::Run uses it (jueying_lio/src/laser_mapping.cc:323-328; PCL is not
 * in the tree: restated from pcl/filters/impl/voxel_grid.hpp): bounding box of the finite points, cell of a point =
 * floor(p * inverse_leaf_size) - min_b, linear index ijk . (1, dx, dx dy), one output point per occupied cell in increasing
 * index order, every float field of the record averaged (downsample_all_data_).  PCL sums the fields in float in the order
 * std::sort happens to leave equal indices; here the sums are double in input order, cast to float.
 * Returns the number of output points, or -1 when the index would overflow int32 (PCL warns and returns the input's size 0).
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { int64_t idx; long cp; } orc_vg_item;
static int vg_cmp(const void *a, const void *b) {
  const orc_vg_item *x = (const orc_vg_item *)a, *y = (const orc_vg_item *)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->cp < y->cp ? -1 : (x->cp > y->cp ? 1 : 0);
}

long orc_voxel_downsample(const float *pts, long n, long stride_floats, float leaf, float *out) {
  const float inv = 1.0f / leaf;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  long nf = 0;
  for (long i = 0; i < n; i++) {
    const float *p = pts + i * stride_floats;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    nf++;
    for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
  }
  if (nf == 0) return 0;
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) return -1;
  int min_b[3], max_b[3];
  for (int a = 0; a < 3; a++) { min_b[a] = (int)floorf(mn[a] * inv); max_b[a] = (int)floorf(mx[a] * inv); }
  const int64_t div0 = max_b[0] - min_b[0] + 1, div1 = max_b[1] - min_b[1] + 1;
  orc_vg_item *it = (orc_vg_item *)malloc(sizeof(orc_vg_item) * (size_t)nf);
  long m = 0;
  for (long i = 0; i < n; i++) {
    const float *p = pts + i * stride_floats;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    const int64_t i0 = (int64_t)(floorf(p[0] * inv) - (float)min_b[0]), i1 = (int64_t)(floorf(p[1] * inv) - (float)min_b[1]), i2 = (int64_t)(floorf(p[2] * inv) - (float)min_b[2]);
    it[m].idx = i0 + i1 * div0 + i2 * div0 * div1;
    it[m].cp = i;
    m++;
  }
  qsort(it, (size_t)m, sizeof(orc_vg_item), vg_cmp);
  long nout = 0;
  double *acc = (double *)malloc(sizeof(double) * (size_t)stride_floats);
  for (long s = 0; s < m;) {
    long e = s;
    for (long f = 0; f < stride_floats; f++) acc[f] = 0.0;
    while (e < m && it[e].idx == it[s].idx) { for (long f = 0; f < stride_floats; f++) acc[f] += (double)pts[it[e].cp * stride_floats + f]; e++; }
    for (long f = 0; f < stride_floats; f++) out[nout * stride_floats + f] = (float)(acc[f] / (double)(e - s));
    nout++;
    s = e;
  }
  free(acc); free(it);
  return nout;
}

/* PointCloudPreprocess::AviaHandler  (jueying_lio/src/pointcloud_preprocess.cc:44-88): livox CustomMsg points -> the PointXYZINormal
 * cloud the matcher receives.  `msg` holds n records of 20 bytes {uint32 offset_time; float x, y, z; uint8 reflectivity, tag, line; pad}
 * (livox_ros_driver/msg/CustomPoint.msg as the generated C++ struct lays it out).  A point i >= 1 is copied into cloud_full_[i] when
 * its line and tag pass and i % point_filter_num == 0 (:58-67), and kept when it differs from cloud_full_[i - 1] -- which is the
 * previous raw point if THAT one was copied, else the zero point the resize() left (:50; the reference's loop is par_unseq: this
 * is its serial reading) -- under the reference's own precedence: |dx| > 1e-7 || |dy| > 1e-7 || (|dz| > 1e-7 && r^2 > blind^2)
 * (:69-74).  out: 12 floats per kept point {x, y, z, 1, 0, 0, 0, 0, intensity, curvature, 0, 0} (pcl::PointXYZINormal), input order. */
long orc_livox_filter(const unsigned char *msg, long n, int num_scans, int point_filter_num, double blind, float *out) {
  long m = 0;
  for (long i = 1; i < n; i++) {
    const unsigned char *r = msg + 20 * i, *rp = msg + 20 * (i - 1);
    unsigned int ot; float xyz[3], pxyz[3] = {0.f, 0.f, 0.f};
    memcpy(&ot, r, 4); memcpy(xyz, r + 4, 12);
    const unsigned char refl = r[16], tag = r[17], line = r[18];
    if (!((int)line < num_scans && ((tag & 0x30) == 0x10 || (tag & 0x30) == 0x00))) continue;
    if (i % point_filter_num != 0) continue;
    if (i - 1 >= 1) {   /* was point i - 1 copied into cloud_full_?  (index 0 never is: the loop starts at 1) */
      const unsigned char ptag = rp[17], pline = rp[18];
      if ((int)pline < num_scans && ((ptag & 0x30) == 0x10 || (ptag & 0x30) == 0x00) && (i - 1) % point_filter_num == 0) memcpy(pxyz, rp + 4, 12);
    }
    const float r2 = xyz[0] * xyz[0] + xyz[1] * xyz[1] + xyz[2] * xyz[2];
    const int keep = ((double)fabsf(xyz[0] - pxyz[0]) > 1e-7) || ((double)fabsf(xyz[1] - pxyz[1]) > 1e-7) ||
                     (((double)fabsf(xyz[2] - pxyz[2]) > 1e-7) && ((double)r2 > blind * blind));
    if (!keep) continue;
    float *o = out + 12 * m;
    o[0] = xyz[0]; o[1] = xyz[1]; o[2] = xyz[2]; o[3] = 1.f; o[4] = o[5] = o[6] = o[7] = 0.f;
    o[8] = (float)refl;                          /* intensity = reflectivity  :64 */
    o[9] = (float)ot / (float)1000000;           /* curvature = offset_time / float(1000000), ms  :65-67 */
    o[10] = o[11] = 0.f;
    m++;
  }
  return m;
}

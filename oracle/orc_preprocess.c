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

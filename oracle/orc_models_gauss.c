/*
 * oracle/orc_models_gauss.c -- GICP / VGICP / NDT residual models of the CPU
 * oracle (TEST INFRASTRUCTURE ONLY).  Filled in after the P2PLANE slice.
 */
#include "orc_internal.h"
#include <stdlib.h>

double orc_gauss_linearize(oracle *o, const double T[16], double *H, double *b) {
  (void)o; (void)T;
  if (H) memset(H, 0, 36 * sizeof(double));
  if (b) memset(b, 0, 6 * sizeof(double));
  return 0.0;
}
double orc_gauss_compute_error(oracle *o, const double T[16]) { (void)o; (void)T; return 0.0; }
void orc_gauss_prepare(oracle *o) { (void)o; }
void orc_gauss_invalidate(oracle *o, int target) { (void)o; (void)target; }
void orc_gauss_swap(oracle *o) { (void)o; }
void orc_gauss_free(oracle *o) { (void)o; }

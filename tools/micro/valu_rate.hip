// Micro-benchmark: VALU issue rate of a wave64 v_fma_f32 / v_add_u32 / v_cndmask stream on gfx950 as a function of waves per SIMD.
// Settles whether a wave64 VALU op occupies its SIMD for 2 or 4 cycles (the roofline of k_linearize depends on it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (KIND == 0) {
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
      } else if (KIND == 1) {
        a0 = a0 < a1 ? a0 : a1 + c; a1 = a1 < a2 ? a1 : a2 + c; a2 = a2 < a3 ? a2 : a3 + c; a3 = a3 < a4 ? a3 : a4 + c;
        a4 = a4 < a5 ? a4 : a5 + c; a5 = a5 < a6 ? a5 : a6 + c; a6 = a6 < a7 ? a6 : a7 + c; a7 = a7 < a0 ? a7 : a0 + c;
      } else {
        a0 = __builtin_amdgcn_fmed3f(a0, a1, a2); a1 = __builtin_amdgcn_fmed3f(a1, a2, a3); a2 = __builtin_amdgcn_fmed3f(a2, a3, a4); a3 = __builtin_amdgcn_fmed3f(a3, a4, a5);
        a4 = __builtin_amdgcn_fmed3f(a4, a5, a6); a5 = __builtin_amdgcn_fmed3f(a5, a6, a7); a6 = __builtin_amdgcn_fmed3f(a6, a7, a0); a7 = __builtin_amdgcn_fmed3f(a7, a0, a1);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 16 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int kind = 0; kind < 3; kind++)
    for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {
      const int grid = 256 * wgs_per_cu;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (kind == 0) k<0><<<grid, 256>>>(d, iters); else if (kind == 1) k<1><<<grid, 256>>>(d, iters); else k<2><<<grid, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_wave = (double)iters * 64 * (kind == 1 ? 3 : 1);   // kind 1: cmp + add + cndmask per statement (approx.)
      const double wave_instr_per_simd = instr_per_wave * wgs_per_cu;            // one wave of each workgroup per SIMD
      printf("kind %d  waves/SIMD %d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", kind, wgs_per_cu, ms,
             ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    }
  return 0;
}

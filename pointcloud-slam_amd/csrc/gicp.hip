// gicp.hip -- GICP and VGICP residual models on the brick voxel hash, gfx950.
//
// Replaces, for the MI355X path (paths relative to
// /root/reference/src/pointcloud_match/fast_gicp/include/fast_gicp/gicp):
//   FastGICP::calculate_covariances ....... impl/fast_gicp_impl.hpp:239-298
//   FastGICP::update_correspondences ...... impl/fast_gicp_impl.hpp:114-152
//   FastGICP::linearize / compute_error ... impl/fast_gicp_impl.hpp:154-237
//   GaussianVoxelMap::create_voxelmap ..... fast_vgicp_voxel.hpp:129-156 (ADDITIVE)
//   FastVGICP::update_correspondences ..... impl/fast_vgicp_impl.hpp:72-124
//   FastVGICP::linearize / compute_error .. impl/fast_vgicp_impl.hpp:126-204
// Shape (not a port): the reference walks a FLANN kd-tree per point.  Here the cloud is already
// grouped by 8x8x8-voxel brick (voxel_hash.hip), so an exact nearest-neighbour query is a scan of
// the voxel columns under the query box -- mask word, popcount, contiguous point runs, no tree -- with an
// exactness test (k-th distance inside the box radius) and one guaranteed-sufficient retry at the
// radius the first pass proved; queries the map cannot answer within 32 voxels fall back to a
// pruned sweep of the brick table, so the result is the exact Euclidean kNN in every case.
// All covariance / Mahalanobis / normal-equation arithmetic is double, as in the reference.
// Compiled with -ffp-contract=off (see kernels.hip).
#include "pcm_device.h"
#include "pcm_host.h"
#include "dev_linalg.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace pcm {

namespace {

__device__ inline uint64_t slot_key(const uint4& s) { return ((uint64_t)s.y << 32) | s.x; }

// Visit every map point whose voxel lies under the box q +- rb.  A point p with |p - q|_inf <= rb is
// visited: the voxel coordinate is monotone in the position.  Inside a brick the voxels of one (x, y)
// column are adjacent bits of one mask word and their points one contiguous run.
template <class F>
__device__ inline void scan_box(const TargetView& tg, int mode, const float (&q)[3], float rb, F&& visit) {
  const float lim = (float)(kCoordBias - 64) * tg.res;
  int lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = voxel_coord(fminf(fmaxf(q[a] - rb, -lim), lim), tg.res, tg.inv_res, mode);
    hi[a] = voxel_coord(fminf(fmaxf(q[a] + rb, -lim), lim), tg.res, tg.inv_res, mode);
  }
  for (int bx = lo[0] >> kBrickShift; bx <= (hi[0] >> kBrickShift); bx++) {
    const int x0 = (lo[0] > bx * 8 ? lo[0] : bx * 8) & 7, x1 = (hi[0] < bx * 8 + 7 ? hi[0] : bx * 8 + 7) & 7;
    for (int by = lo[1] >> kBrickShift; by <= (hi[1] >> kBrickShift); by++) {
      const int y0 = (lo[1] > by * 8 ? lo[1] : by * 8) & 7, y1 = (hi[1] < by * 8 + 7 ? hi[1] : by * 8 + 7) & 7;
      for (int bz = lo[2] >> kBrickShift; bz <= (hi[2] >> kBrickShift); bz++) {
        const int z0 = (lo[2] > bz * 8 ? lo[2] : bz * 8) & 7, z1 = (hi[2] < bz * 8 + 7 ? hi[2] : bz * 8 + 7) & 7;
        const uint64_t key = pack_brick(bx, by, bz);
        uint32_t h = hash_coord(bx, by, bz) & tg.mask;
        uint4 s;
        bool found = false;
        for (;;) {
          s = gload4u(&tg.bricks[h]);
          const uint64_t sk = slot_key(s);
          if (sk == key) { found = true; break; }
          if (sk == kEmptyKey) break;
          h = (h + 1) & tg.mask;
        }
        if (!found) continue;
        const uint32_t zbits = (1u << (z1 - z0 + 1)) - 1u;
        for (int x = x0; x <= x1; x++) {
          for (int y = y0; y <= y1; y++) {
            const uint32_t w = (uint32_t)(x * 2 + (y >> 2)), sh = (uint32_t)((y & 3) * 8 + z0);
            const uint32_t m = gload_u(&tg.bmask[(size_t)h * 16 + w]);
            const uint32_t sel = m & (zbits << sh);
            if (!sel) continue;
            const uint32_t vs = s.z + gload_u16(&tg.bpref[(size_t)h * 16 + w]) + (uint32_t)__popc(m & ((1u << sh) - 1u));
            const uint32_t ps = gload_u(&tg.vox_start[vs]), pe = gload_u(&tg.vox_start[vs + (uint32_t)__popc(sel)]);
            for (uint32_t p = ps; p < pe; p++) {
              const float4 c = gload4(tg.pts + p);
              const float ex = c.x - q[0], ey = c.y - q[1], ez = c.z - q[2];
              visit(p, ex * ex + ey * ey + ez * ez);
            }
          }
        }
      }
    }
  }
}

// Sweep of the whole brick table, skipping bricks whose box is provably farther than bound().
template <class B, class F>
__device__ inline void scan_all(const TargetView& tg, int mode, const float (&q)[3], B&& bound, F&& visit) {
  const float shift = mode == COORD_ROUND ? -0.5f : 0.5f;   // voxel c covers [(c + shift) res, (c + shift + 1) res]
  for (uint32_t h = 0; h <= tg.mask; h++) {
    const uint4 s = gload4u(&tg.bricks[h]);
    const uint64_t sk = slot_key(s);
    if (sk == kEmptyKey) continue;
    const int b[3] = {(int)(sk >> 36) - kBrickBias, (int)((sk >> 18) & 0x3ffff) - kBrickBias, (int)(sk & 0x3ffff) - kBrickBias};
    float lb = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float slack = 1e-3f * tg.res + 4e-6f * fabsf(q[a]);
      const float blo = ((float)(b[a] * 8) + shift) * tg.res - slack, bhi = ((float)(b[a] * 8 + 8) + shift) * tg.res + slack;
      const float d = fmaxf(fmaxf(blo - q[a], q[a] - bhi), 0.f);
      lb += d * d;
    }
    if (lb * 0.999f > bound()) continue;
    const uint4 s2 = gload4u(reinterpret_cast<const uint4*>(&tg.bricks[h]) + 1);   // pt_start, npts
    for (uint32_t p = s2.x; p < s2.x + s2.y; p++) {
      const float4 c = gload4(tg.pts + p);
      const float ex = c.x - q[0], ey = c.y - q[1], ez = c.z - q[2];
      visit(p, ex * ex + ey * ey + ez * ez);
    }
  }
}

__device__ inline bool finite3(const float (&q)[3]) { return isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]); }

// exact nearest map point of q: index or -1 when (double)d2 >= max_sq   (nearestKSearch(pt, 1) + the threshold of fast_gicp_impl.hpp:136)
__device__ inline int nearest1(const TargetView& tg, int mode, const float (&q)[3], double max_sq) {
  if (!finite3(q)) return -1;
  uint64_t best = ~0ull;
  auto visit = [&](uint32_t p, float d2) {
    const uint64_t key = ((uint64_t)__float_as_uint(d2) << 32) | p;
    if (d2 == d2 && key < best) best = key;
  };
  float r = tg.res;
  bool exact = false;
  for (;;) {
    best = ~0ull;
    scan_box(tg, mode, q, r * 1.0001f, visit);
    const float d2 = __uint_as_float((uint32_t)(best >> 32));
    if (best != ~0ull && d2 < r * r) { exact = true; break; }
    if ((double)r * (double)r >= max_sq) { exact = true; break; }   // nothing closer than the threshold exists
    const float rn = best != ~0ull ? sqrtf(d2) * 1.001f : 2.f * r;
    if (rn > 32.f * tg.res) break;
    r = rn;
  }
  if (!exact) {
    best = ~0ull;
    scan_all(tg, mode, q, [&]() { return best == ~0ull ? 3.0e38f : __uint_as_float((uint32_t)(best >> 32)); }, visit);
  }
  if (best == ~0ull) return -1;
  const float d2 = __uint_as_float((uint32_t)(best >> 32));
  return (double)d2 < max_sq ? (int)(uint32_t)best : -1;
}

__device__ inline void regularize_cov(int method, const double (&cov)[9], double (&out)[9]) {
  if (method == PCM_REG_NONE) {
#pragma unroll
    for (int a = 0; a < 9; a++) out[a] = cov[a];
    return;
  }
  if (method == PCM_REG_FROBENIUS) {   // fast_gicp_impl.hpp:266-271
    double C[9], Ci[9], N[9];
#pragma unroll
    for (int a = 0; a < 9; a++) C[a] = cov[a];
    C[0] += 1e-3; C[4] += 1e-3; C[8] += 1e-3;
    inv3<double>(C, Ci);
    // Matrix3d::norm(): fixed-size sum over the column-major coefficients: four packets of two by a tree, then the ninth (CORE-1)
    double sq[9];
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int b = 0; b < 3; b++) sq[b * 3 + a] = Ci[a * 3 + b] * Ci[a * 3 + b];
    }
    const double nrm = sqrt((((sq[0] + sq[2]) + (sq[4] + sq[6])) + ((sq[1] + sq[3]) + (sq[5] + sq[7]))) + sq[8]);
#pragma unroll
    for (int a = 0; a < 9; a++) N[a] = Ci[a] / nrm;
    inv3<double>(N, out);
    return;
  }
  // Eigen::JacobiSVD<Matrix3d>(cov, ComputeFullU | ComputeFullV)  :273 (two-sided Jacobi, dev_linalg.h)
  double U[9], S[3], V[9], val[3];
  jacobi_svd<3>(cov, U, S, V);
  if (method == PCM_REG_PLANE) { val[0] = 1.0; val[1] = 1.0; val[2] = 1e-3; }                                  // :280
  else if (method == PCM_REG_MIN_EIG) {
#pragma unroll
    for (int k = 0; k < 3; k++) val[k] = S[k] > 1e-3 ? S[k] : 1e-3;                                            // :283
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) { const double v = S[k] / S[0]; val[k] = v > 1e-3 ? v : 1e-3; }                // :286-287
  }
  // svd.matrixU() * values.asDiagonal() * svd.matrixV().transpose()  :292
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) out[a * 3 + b] = ((U[a * 3 + 0] * val[0]) * V[b * 3 + 0] + (U[a * 3 + 1] * val[1]) * V[b * 3 + 1]) + (U[a * 3 + 2] * val[2]) * V[b * 3 + 2];
  }
}

// covariance_regularization.cu:14-121 (float): PLANE / MIN_EIG through V diag V^-1 with the general inverse of the
// eigenvector matrix, FROBENIUS; the other methods are unimplemented there and leave the matrix as it is.
__device__ inline void regularize_cov_f(int method, float (&c)[9]) {
  if (method == PCM_REG_FROBENIUS) {
    float C[9], Ci[9], N[9];
#pragma unroll
    for (int a = 0; a < 9; a++) C[a] = c[a];
    C[0] += 1e-3f; C[4] += 1e-3f; C[8] += 1e-3f;
    inv3<float>(C, Ci);
    float nrm = 0.0f;
#pragma unroll
    for (int a = 0; a < 9; a++) nrm += Ci[a] * Ci[a];
    nrm = sqrtf(nrm);
#pragma unroll
    for (int a = 0; a < 9; a++) N[a] = Ci[a] / nrm;
    inv3<float>(N, c);
    return;
  }
  if (method != PCM_REG_PLANE && method != PCM_REG_MIN_EIG) return;
  // SelfAdjointEigenSolver<Matrix3f>::computeDirect  covariance_regularization.cu:57-58,84-85 (closed form, dev_linalg.h)
  float w[3], Vf[9], Vi[9], val[3], VD[9];
  selfadjoint3_direct(c, w, Vf);
  if (method == PCM_REG_PLANE) { val[0] = 1e-3f; val[1] = 1.0f; val[2] = 1.0f; }
  else {
#pragma unroll
    for (int k = 0; k < 3; k++) val[k] = fmaxf(1e-3f, w[k]);
  }
  inv3<float>(Vf, Vi);
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) VD[a * 3 + b] = Vf[a * 3 + b] * val[b];
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int b = 0; b < 3; b++) c[a * 3 + b] = (VD[a * 3 + 0] * Vi[0 * 3 + b] + VD[a * 3 + 1] * Vi[1 * 3 + b]) + VD[a * 3 + 2] * Vi[2 * 3 + b];
  }
}

// ---------------------------------------------------------------------------
// k_covariances: 64 CONSECUTIVE map points per WORKGROUP, one query per lane, the four waves of the workgroup sharing the SAME 64
// queries and splitting the candidates.  The map is brick-major, so the 64 queries sit in neighbouring voxels and want nearly the
// same candidates.  Per pass the workgroup takes the voxel box of its queries grown by r, spreads the box's voxel-column segments
// over its 256 lanes (one hash probe + mask word + two vox_start words per lane, all in flight together), prefix-sums the run
// lengths, stages the runs' points into LDS 1024 at a time, and every wave scans its quarter of the staged points (broadcast LDS
// reads) into the k best of its lanes -- per query four partial lists.  A query is finished when the smallest of its four k-th
// distances (an upper bound of the true k-th) lies inside its margin to the faces of the scanned box; the next r is the largest
// radius an unfinished query has proven (k points seen) or 1.25 r.  At the end wave 0 folds the other three lists into its own
// and computes the covariances.  (One wave per 64 queries, round 2: a 100 k-point scan is 1 564 waves on 1 024 SIMDs, every one
// of them a serial chain of probe rounds and scans -- mean 550 us, slowest 2 ms = the kernel; the chip was idle.)
// Queries the box passes cannot finish (r beyond 16 voxels, scattered workgroups) run the per-lane search in wave 0.
// dynamic LDS: 1024 staged points + 257 offsets + 256 run starts + exchange words; re-used for the final merge
// ---------------------------------------------------------------------------
constexpr int kStageCap = 1024;
constexpr int kCovLdsFixed = kStageCap * 16 + 264 * 4 + 256 * 4 + 8 * 4 + 2 * 4 * 64 * 4;   // staging, soff, sps, wave sums, k-th distances, list fills
__host__ __device__ constexpr size_t cov_lds_bytes(int kcap) {
  const size_t merge = (size_t)3 * kcap * 64 * 8;   // three waves' lists, (distance, index) per slot
  return (merge > (size_t)kStageCap * 16 ? merge : (size_t)kStageCap * 16) + (kCovLdsFixed - kStageCap * 16);
}

__device__ inline float wave_min_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ inline float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

#ifdef PCM_COV_STATS   // diagnostic build only (tools/r03_cov_stats.sh): where the passes of k_covariances go
__device__ unsigned long long g_cov_stats[32];
#define COV_T0() const unsigned long long t_ph0 = wall_clock64()
#define COV_T(i) do { if (threadIdx.x == 0) atomicAdd(&g_cov_stats[i], wall_clock64() - t_ph0); } while (0)
#define COV_STAT(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_cov_stats[i], (unsigned long long)(v)); } while (0)
#else
#define COV_STAT(i, v) do { } while (0)
#define COV_T0() do { } while (0)
#define COV_T(i) do { } while (0)
#endif

template <int KCAP>
__global__ void __launch_bounds__(256, KCAP <= 20 ? 3 : 1) k_covariances(TargetView tg, TargetView tf, const uint32_t* __restrict__ f_order, const uint32_t* __restrict__ c_inv, int mode, int k,
                                                                      int reg, double* __restrict__ out) {
  extern __shared__ uint64_t s_dyn[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  char* const lbase = reinterpret_cast<char*>(s_dyn);
  const size_t stage_bytes = cov_lds_bytes(KCAP) - (kCovLdsFixed - kStageCap * 16);
  float4* st = reinterpret_cast<float4*>(lbase);                          // [1024] staged points (the merge area later)
  uint32_t* soff = reinterpret_cast<uint32_t*>(lbase + stage_bytes);      // [257] exclusive offsets of the round's runs
  uint32_t* sps = soff + 264;                                             // [256] first point of each run
  uint32_t* s_wsum = sps + 256;                                           // [4] run-length totals of the waves
  float* s_dk = reinterpret_cast<float*>(s_wsum + 8);                     // [4][64] k-th distance of every wave's partial list
  int* s_fill = reinterpret_cast<int*>(s_dk + 256);                       // [4][64] points in every wave's partial list
  // With a fine index (tf) the queries are taken in ITS order -- 64 consecutive points of a 1/8-size grid are one small patch even
  // where a whole voxel of the map's own grid holds thousands of points in insertion order -- and the row goes to the query's place
  // in the map (c_inv[f_order[.]]: fine position -> input index -> map position).
  const bool has_fine = tf.num_points != 0u;
  const uint32_t i = blockIdx.x * 64 + lane;
  const uint32_t nq = has_fine ? tf.num_points : tg.num_points;
  const bool active = i < nq;
  const float4 pq = gload4((has_fine ? tf.pts : tg.pts) + (active ? i : nq - 1u));
  const uint32_t oi = !active ? 0u : has_fine ? gload_u(c_inv + gload_u(f_order + i)) : i;
  const float q[3] = {pq.x, pq.y, pq.z};
  // The k best of a lane live in registers, sorted ascending in the LAST k of KCAP slots (the slots in
  // front hold a -1 sentinel no distance undercuts): the k-th best is always bd[KCAP - 1], and an
  // insertion is a branch-free shift network -- no LDS round trip per shifted element.
  float bd[KCAP];
  uint32_t bi[KCAP];
  // `cap`: only candidates strictly nearer than sqrt(cap) are kept.  Without one every staged point enters some lane's list while
  // the lists fill, and the shift network below runs for the whole wave whenever ANY lane inserts: ~1 800 staged points per pass,
  // ~68 instructions each, for the ~170 that end up mattering.  A slot that holds no point yet carries the cap and bi = ~0u.
  auto reset = [&](float cap) {
#pragma unroll
    for (int j = 0; j < KCAP; j++) { bd[j] = j < KCAP - k ? -1.f : cap; bi[j] = ~0u; }
  };
  reset(3.0e38f);
  auto visit = [&](uint32_t p, float d2) {
    if (!(d2 < bd[KCAP - 1])) return;   // also rejects NaN; an exact tie with the k-th keeps the earlier point
    bool c_cur = true;                  // d2 < bd[j] (old value)
#pragma unroll
    for (int j = KCAP - 1; j >= 1; j--) {
      const bool c_prev = d2 < bd[j - 1];
      bd[j] = c_prev ? bd[j - 1] : (c_cur ? d2 : bd[j]);
      bi[j] = c_prev ? bi[j - 1] : (c_cur ? p : bi[j]);
      c_cur = c_prev;
    }
    if (c_cur) { bd[0] = d2; bi[0] = p; }
  };
  const float shift = mode == COORD_ROUND ? -0.5f : 0.5f;   // voxel c covers [(c + shift) res, (c + shift + 1) res]
  // Box passes of the queries `mine` over the grid `tv`.  Every decision below is taken from values all four waves hold alike (the
  // same queries, the same reductions), so the workgroup's control flow is uniform and the barriers are safe.  Returns, per lane,
  // whether the k best of its query -- spread over the four waves' lists -- are exact (true for lanes outside `mine`).
  auto box_passes = [&](const TargetView& tv, bool mine, float r, const float rmax, const float (&bmin)[3], const float (&bmax)[3], int stat0) -> bool {
    bool gexact = !mine;
    const float lim = (float)(kCoordBias - 64) * tv.res;
    int plo[3] = {0, 0, 0}, phi[3] = {-1, -1, -1};
    // the radius that holds k points, estimated from the queries' own patch: n queries on an area e1 x e2 (the two larger extents
    // of their box) -> surface density -> disc with k points.  It sizes the first box (1.3 x) and caps the first pass's lists (2 x);
    // a wrong estimate costs a pass, never the result: a list that did not fill below its cap proves nothing and the pass is redone.
    float capl = 3.0e38f;
    {
      const float e0 = bmax[0] - bmin[0], e1 = bmax[1] - bmin[1], e2 = bmax[2] - bmin[2];
      const float emax = fmaxf(e0, fmaxf(e1, e2)), emin = fminf(e0, fminf(e1, e2)), emid = (e0 + e1 + e2) - emax - emin;
      const float nm = (float)__popcll(__ballot(mine));
      const float r_est = sqrtf((float)k * emax * emid / (3.14159265f * fmaxf(nm, 1.f)));
      if (r_est > 0.f && r_est < rmax) {
        r = fmaxf(r, 1.3f * r_est);
        capl = 4.f * r_est * r_est;
      }
    }
    while (r <= rmax && __ballot(!gexact) != 0ull) {
      int lo[3], hi[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        lo[a] = __builtin_amdgcn_readfirstlane(voxel_coord(fminf(fmaxf(bmin[a] - r, -lim), lim), tv.res, tv.inv_res, mode));
        hi[a] = __builtin_amdgcn_readfirstlane(voxel_coord(fminf(fmaxf(bmax[a] + r, -lim), lim), tv.res, tv.inv_res, mode));
      }
      if (lo[0] == plo[0] && lo[1] == plo[1] && lo[2] == plo[2] && hi[0] == phi[0] && hi[1] == phi[1] && hi[2] == phi[2]) { r *= 1.5f; continue; }
#pragma unroll
      for (int a = 0; a < 3; a++) { plo[a] = lo[a]; phi[a] = hi[a]; }
      const int nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, bz_lo = lo[2] >> kBrickShift, nbz = (hi[2] >> kBrickShift) - bz_lo + 1;
      if (nx > 64 || ny > 64 || nbz > 8 || nx * ny * nbz > 4096) break;   // scattered queries: per-lane search
      const int nseg = nx * ny * nbz;
      if (!gexact) reset(capl);
      COV_STAT(stat0, 1); COV_STAT(stat0 + 1, (nseg + 255) / 256);
      for (int segbase = 0; segbase < nseg; segbase += 256) {
        // ---- one voxel-column segment (x, y, brick-z) per lane of the workgroup ----
        const int sg = segbase + (int)tid;
        uint32_t cnt = 0, ps = 0;
        COV_T0();
        if (sg < nseg) {
          const int ibz = sg % nbz, t2 = sg / nbz, iy = t2 % ny, ix = t2 / ny;
          const int x = lo[0] + ix, y = lo[1] + iy, bz = bz_lo + ibz;
          const int z0 = (lo[2] > bz * 8 ? lo[2] : bz * 8) & 7, z1 = (hi[2] < bz * 8 + 7 ? hi[2] : bz * 8 + 7) & 7;
          const int bx = x >> kBrickShift, by = y >> kBrickShift;
          const uint64_t key = pack_brick(bx, by, bz);
          uint32_t h = hash_coord(bx, by, bz) & tv.mask;
          for (;;) {
            const uint4 sl = gload4u(&tv.bricks[h]);
            const uint64_t sk = slot_key(sl);
            if (sk == key) {
              const uint32_t w = (uint32_t)((x & 7) * 2 + ((y & 7) >> 2)), sh = (uint32_t)(((y & 7) & 3) * 8 + z0);
              const uint32_t m = gload_u(&tv.bmask[(size_t)h * 16 + w]);
              const uint32_t sel = m & (((1u << (z1 - z0 + 1)) - 1u) << sh);
              if (sel) {
                const uint32_t vs = sl.z + gload_u16(&tv.bpref[(size_t)h * 16 + w]) + (uint32_t)__popc(m & ((1u << sh) - 1u));
                ps = gload_u(&tv.vox_start[vs]);
                cnt = gload_u(&tv.vox_start[vs + (uint32_t)__popc(sel)]) - ps;
              }
              break;
            }
            if (sk == kEmptyKey) break;
            h = (h + 1) & tv.mask;
          }
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t t = __shfl_up(incl, off, 64);
          if ((int)lane >= off) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();   // also: the previous round's staged points have been scanned by every wave
        COV_T(24);
        const uint32_t w0 = s_wsum[0], w1 = s_wsum[1], w2 = s_wsum[2], w3 = s_wsum[3];
        const uint32_t total = w0 + w1 + w2 + w3;
        if (total == 0u) { __syncthreads(); continue; }   // s_wsum is rewritten by the next round
        COV_STAT(stat0 + 2, total);
        const uint32_t wbase = wave == 0 ? 0u : wave == 1 ? w0 : wave == 2 ? w0 + w1 : w0 + w1 + w2;
        soff[tid] = wbase + incl - cnt;
        sps[tid] = ps;
        if (tid == 0) soff[256] = total;
        __syncthreads();
        // ---- stage the runs' points 1024 at a time; every wave scans its quarter for its unfinished lanes ----
        for (uint32_t r0 = 0; r0 < total; r0 += kStageCap) {
          const uint32_t nst = total - r0 < (uint32_t)kStageCap ? total - r0 : (uint32_t)kStageCap;
          COV_T0();
          for (uint32_t t = r0 + tid; t < r0 + nst; t += 256) {
            uint32_t a = 0;   // run of staged point t: the last one whose offset is <= t
#pragma unroll
            for (uint32_t step = 128; step >= 1; step >>= 1) { if (soff[a + step] <= t) a += step; }
            const uint32_t p = sps[a] + (t - soff[a]);
            float4 c = gload4(tv.pts + p);
            c.w = __uint_as_float(p);
            st[t - r0] = c;
          }
          __syncthreads();
          COV_T(25);
          if (!gexact) {
            // four staged points per trip, all four LDS reads under way before the first is used: one point per trip is one LDS
            // round trip per point (the scan was 120 of a workgroup's 210 us, ~500 cycles per point and wave).
            // (Queueing the candidates that pass a lane's threshold per lane and running the shift network on full queues only --
            // every lane inserting in the same turn -- was tried: 131 -> 161 us, the network is not what the scan waits for.)
            uint32_t j = wave;
            for (; j + 12 < nst; j += 16) {
              const float4 c0 = st[j], c1 = st[j + 4], c2 = st[j + 8], c3 = st[j + 12];
              { const float ex = c0.x - q[0], ey = c0.y - q[1], ez = c0.z - q[2]; visit(__float_as_uint(c0.w), ex * ex + ey * ey + ez * ez); }
              { const float ex = c1.x - q[0], ey = c1.y - q[1], ez = c1.z - q[2]; visit(__float_as_uint(c1.w), ex * ex + ey * ey + ez * ez); }
              { const float ex = c2.x - q[0], ey = c2.y - q[1], ez = c2.z - q[2]; visit(__float_as_uint(c2.w), ex * ex + ey * ey + ez * ez); }
              { const float ex = c3.x - q[0], ey = c3.y - q[1], ez = c3.z - q[2]; visit(__float_as_uint(c3.w), ex * ex + ey * ey + ez * ez); }
            }
            for (; j < nst; j += 4) {
              const float4 c = st[j];
              const float ex = c.x - q[0], ey = c.y - q[1], ez = c.z - q[2];
              visit(__float_as_uint(c.w), ex * ex + ey * ey + ez * ez);
            }
          }
          __syncthreads();
          COV_T(26);   // staging + scan
        }
      }
      // margin of this lane to the faces of the scanned voxel box: every unseen point is farther away
      float mg = 3.0e38f;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float slack = 1e-4f * tv.res + 4e-6f * fabsf(q[a]);
        const float wlo = ((float)lo[a] + shift) * tv.res, whi = ((float)hi[a] + shift + 1.f) * tv.res;
        mg = fminf(mg, fminf(q[a] - wlo, whi - q[a]) - slack);
      }
      // an upper bound of the query's true k-th distance: the smallest k-th of a partial list that is full, or the cap when the four
      // lists together hold k points (all of them nearer than the cap)
      int fill = 0;
#pragma unroll
      for (int j = 0; j < KCAP; j++) fill += (j >= KCAP - k && bi[j] != ~0u) ? 1 : 0;
      s_dk[wave * 64 + lane] = fill == k ? bd[KCAP - 1] : 3.0e38f;
      s_fill[wave * 64 + lane] = fill;
      __syncthreads();
      float d2k = fminf(fminf(s_dk[lane], s_dk[64 + lane]), fminf(s_dk[128 + lane], s_dk[192 + lane]));
      if (s_fill[lane] + s_fill[64 + lane] + s_fill[128 + lane] + s_fill[192 + lane] >= k) d2k = fminf(d2k, capl);
      __syncthreads();
      float need = 0.f;
      if (!gexact) {
        const bool seen = d2k < 3.0e38f;
        if (seen && mg > 0.f && d2k < mg * mg) gexact = true;
        else {
          need = seen ? sqrtf(d2k) * 1.001f : 2.f * r;   // k points seen: they lie inside this radius
          capl = seen ? need * need : 3.0e38f;            // ... and strictly inside the next pass's cap
        }
      }
      r = fmaxf(wave_max_f(need), 1.25f * r);
    }
    return gexact;
  };
  bool exact = !active;   // lanes past the end only help with the staging
  bool from_fine = false;
  COV_STAT(0, 1);
#ifdef PCM_COV_STATS
  const unsigned long long t_wave0 = wall_clock64();
#endif
  if (has_fine) {
    // ---- the fine grid first: worth it where the radius that holds k points is a fraction of the map's own voxel --------------
    float bmin[3], bmax[3], e[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { bmin[a] = wave_min_f(active ? q[a] : 3.0e38f); bmax[a] = wave_max_f(active ? q[a] : -3.0e38f); e[a] = bmax[a] - bmin[a]; }
    const bool finite = isfinite(bmin[0]) && isfinite(bmin[1]) && isfinite(bmin[2]) && isfinite(bmax[0]) && isfinite(bmax[1]) && isfinite(bmax[2]);
    const float emax = fmaxf(e[0], fmaxf(e[1], e[2])), emin = fminf(e[0], fminf(e[1], e[2]));
    const float emid = (e[0] + e[1] + e[2]) - emax - emin;
    const float nact = (float)__popcll(__ballot(active));
    const float r_est = 1.3f * sqrtf((float)k * emax * emid / (3.14159265f * nact));
    if (finite && emax <= 8.f * tf.res && r_est <= 2.f * tf.res) {
      COV_STAT(1, 1);
      const bool ok = box_passes(tf, active, 0.25f * tf.res, 4.f * tf.res, bmin, bmax, 2);
      if (active) {
        if (ok) { exact = true; from_fine = true; }
        else reset(3.0e38f);
      }
    }
  }
  // A workgroup whose queries are far apart in space (the brick order jumps) is served in groups of lanes
  // within 8 voxels of the first pending lane: each group gets its own box passes, the others only help staging.
  uint64_t pending = __ballot(active && !exact);
  for (int grp = 0; grp < 8 && pending != 0ull; grp++) {
    const int lead = __ffsll((unsigned long long)pending) - 1;
    bool mine = active && ((pending >> lane) & 1ull);
#pragma unroll
    for (int a = 0; a < 3; a++) mine = mine && fabsf(q[a] - __shfl(q[a], lead, 64)) <= 8.f * tg.res;
    pending &= ~__ballot(mine);
    float bmin[3], bmax[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { bmin[a] = wave_min_f(mine ? q[a] : 3.0e38f); bmax[a] = wave_max_f(mine ? q[a] : -3.0e38f); }
    const bool finite = isfinite(bmin[0]) && isfinite(bmin[1]) && isfinite(bmin[2]) && isfinite(bmax[0]) && isfinite(bmax[1]) && isfinite(bmax[2]);
    bool gexact = !mine;
    COV_STAT(6, 1);
    if (finite) gexact = box_passes(tg, mine, 0.25f * tg.res, 16.f * tg.res, bmin, bmax, 7);
    if (mine) exact = gexact;
  }
  // queries the box passes left open: the per-lane search below, by wave 0 alone (it rebuilds the whole list; the other waves'
  // partial lists of such a query are dropped)
  if (!exact && wave != 0) reset(3.0e38f);
  COV_STAT(10, __popcll(__ballot(!exact)));
#ifdef PCM_COV_STATS
  {   // search time of this wave (100 MHz ticks): sum, max, histogram < 50 / 200 / 800 / 3200 us / more; by path
    const unsigned long long dt = wall_clock64() - t_wave0;
    if (threadIdx.x == 0) {
      atomicAdd(&g_cov_stats[11], dt);
      atomicMax(&g_cov_stats[12], dt);
      const int b = dt < 5000 ? 0 : dt < 20000 ? 1 : dt < 80000 ? 2 : dt < 320000 ? 3 : 4;
      atomicAdd(&g_cov_stats[13 + b], 1ull);
      atomicAdd(&g_cov_stats[18 + b], dt);
    }
  }
#endif
  if (!exact && wave == 0) {   // per-lane search (sparse neighbourhoods, scattered workgroups)
    float r = bi[KCAP - 1] != ~0u ? sqrtf(bd[KCAP - 1]) * 1.001f : 2.f * tg.res;   // a full (partial) list: k points within its k-th distance
    for (;;) {
      if (r > 32.f * tg.res) break;
      reset(3.0e38f);
      scan_box(tg, mode, q, r * 1.0001f, visit);
      float rn = 2.f * r;
      if (bd[KCAP - 1] < 3.0e38f) {
        if (bd[KCAP - 1] < r * r) { exact = true; break; }
        rn = sqrtf(bd[KCAP - 1]) * 1.001f;
      }
      r = rn;
    }
    if (!exact) {
      reset(3.0e38f);
      scan_all(tg, mode, q, [&]() { return bd[KCAP - 1]; }, visit);
    }
  }
  // ---- wave 0 folds the other waves' partial lists into its own (the staging area is free now) ----
  {
    uint2* mg2 = reinterpret_cast<uint2*>(lbase);   // [3][KCAP][64]
    __syncthreads();
    if (wave != 0) {
#pragma unroll
      for (int j = 0; j < KCAP; j++) mg2[((wave - 1) * KCAP + j) * 64 + lane] = make_uint2(__float_as_uint(bd[j]), bi[j]);
    }
    __syncthreads();
    if (wave != 0) return;
    for (int w = 0; w < 3; w++) {
      for (int j = KCAP - k; j < KCAP; j++) {
        const uint2 e2 = mg2[(w * KCAP + j) * 64 + lane];
        if (e2.y != ~0u) visit(e2.y, __uint_as_float(e2.x));
      }
    }
  }
#ifdef PCM_COV_STATS
  if (threadIdx.x == 0) atomicAdd(&g_cov_stats[27], wall_clock64() - t_wave0);   // search + merge
#endif
  if (!active) return;
  const float4* nb = from_fine ? tf.pts : tg.pts;   // bi[] index the grid that answered
  if (reg >= 16) {
    // CUDA-core semantics (PCM_MODEL_VGICP_CUDA): float sums over the neighbours nearest first, cov = S / k - mean mean^T
    // (covariance_estimation.cu:16-42), float regularisation; the 9 floats of the (not exactly symmetric) result go into
    // the point's 48-byte slot
    float meanf[3] = {0.f, 0.f, 0.f}, cf[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KCAP; j++) {
      if (j >= KCAP - k && bi[j] != ~0u) {
        const float4 c = gload4(nb + bi[j]);
        const float x[3] = {c.x, c.y, c.z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
          meanf[a] += x[a];
#pragma unroll
          for (int b = 0; b < 3; b++) cf[a * 3 + b] += x[a] * x[b];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) meanf[a] /= (float)k;
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int b = 0; b < 3; b++) cf[a * 3 + b] = cf[a * 3 + b] / (float)k - meanf[a] * meanf[b];
    }
    regularize_cov_f(reg - 16, cf);
    float* of = reinterpret_cast<float*>(out + (size_t)oi * 6);
#pragma unroll
    for (int a = 0; a < 9; a++) gstore_f(of + a, cf[a]);
    return;
  }
  double mean[3] = {0.0, 0.0, 0.0}, cov[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (reg == PCM_REG_PCLOMP) {
    // pclomp computeCovariances (gicp_omp_impl.hpp:48-122): raw second moments, FLOAT products added to doubles nearest
    // first, cov = S / k - mean mean^T on the lower triangle (mirrored), singular values -> (1, 1, 0.001), largest first
#pragma unroll
    for (int j = 0; j < KCAP; j++) {
      if (j >= KCAP - k && bi[j] != ~0u) {
        const float4 c = gload4(nb + bi[j]);
        const float x[3] = {c.x, c.y, c.z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
          mean[a] += (double)x[a];
#pragma unroll
          for (int b = 0; b <= a; b++) cov[a * 3 + b] += (double)(x[a] * x[b]);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) mean[a] /= (double)k;
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int b = 0; b <= a; b++) {
        cov[a * 3 + b] /= (double)k;
        cov[a * 3 + b] -= mean[a] * mean[b];
        cov[b * 3 + a] = cov[a * 3 + b];
      }
    }
    double U[9], S[3], R[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    jacobi_svd<3, true, false>(cov, U, S, nullptr);   // JacobiSVD<Matrix3d>(cov, ComputeFullU)  gicp_omp_impl.hpp:110
    const double val[3] = {1.0, 1.0, 0.001};
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {        // cov += v * col * col.transpose()  :114-120
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) R[a * 3 + b] += (val[kk] * U[a * 3 + kk]) * U[b * 3 + kk];
      }
    }
    double* o = out + (size_t)oi * 6;
    gstore_d(o + 0, R[0]); gstore_d(o + 1, R[1]); gstore_d(o + 2, R[2]);
    gstore_d(o + 3, R[4]); gstore_d(o + 4, R[5]); gstore_d(o + 5, R[8]);
    return;
  }
  // neighbours (k columns; the mean and the covariance divide by k)  fast_gicp_impl.hpp:254-260
#pragma unroll
  for (int j = 0; j < KCAP; j++) {
    if (j >= KCAP - k && bi[j] != ~0u) {
      const float4 c = gload4(nb + bi[j]);
      mean[0] += (double)c.x; mean[1] += (double)c.y; mean[2] += (double)c.z;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) mean[a] /= (double)k;
#pragma unroll
  for (int j = 0; j < KCAP; j++) {
    if (j >= KCAP - k && bi[j] != ~0u) {
      const float4 c = gload4(nb + bi[j]);
      const double d[3] = {(double)c.x - mean[0], (double)c.y - mean[1], (double)c.z - mean[2]};
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) cov[a * 3 + b] += d[a] * d[b];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 9; a++) cov[a] /= (double)k;
  double R[9];
  regularize_cov(reg, cov, R);
  double* o = out + (size_t)oi * 6;
  gstore_d(o + 0, R[0]); gstore_d(o + 1, R[1]); gstore_d(o + 2, R[2]);
  gstore_d(o + 3, R[4]); gstore_d(o + 4, R[5]); gstore_d(o + 5, R[8]);
#ifdef PCM_COV_STATS
  if (threadIdx.x == 0) atomicAdd(&g_cov_stats[28], wall_clock64() - t_wave0);   // whole workgroup
#endif
}

// voxel distributions of the VGICP map: a voxel's points are one run in input order
// mode 0 / 1: AdditiveGaussianVoxel (fast_vgicp_voxel.hpp:104-122); mode 2: MultiplicativeGaussianVoxel (:79-102)
__global__ void __launch_bounds__(128) k_vgicp_voxels(const float4* __restrict__ pts, const uint32_t* __restrict__ vox_start, const double* __restrict__ covs,
                                                      uint32_t nvox, int mode, VgVoxel* __restrict__ out) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t p0 = vox_start[v], p1 = vox_start[v + 1];
  double mean[3] = {0.0, 0.0, 0.0}, cov[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (uint32_t p = p0; p < p1; p++) {   // append()
    const float4 c = pts[p];
    const double x[3] = {(double)c.x, (double)c.y, (double)c.z};
    const double* s6 = covs + (size_t)p * 6;
    const double C[9] = {s6[0], s6[1], s6[2], s6[1], s6[3], s6[4], s6[2], s6[4], s6[5]};
    if (mode == 2) {
      double Ci[9];
      inv3<double>(C, Ci);
#pragma unroll
      for (int a = 0; a < 9; a++) cov[a] += Ci[a];
#pragma unroll
      for (int a = 0; a < 3; a++) mean[a] += (Ci[a * 3 + 0] * x[0] + Ci[a * 3 + 1] * x[1]) + Ci[a * 3 + 2] * x[2];
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++) mean[a] += x[a];
#pragma unroll
      for (int a = 0; a < 9; a++) cov[a] += C[a];
    }
  }
  VgVoxel g;
  g.n = (int32_t)(p1 - p0);
  g.pad = 0;
  if (mode == 2) {   // finalize(): cov = cov^-1, mean = cov * mean
    double C[9];
    inv3<double>(cov, C);
#pragma unroll
    for (int a = 0; a < 3; a++) g.mean[a] = (C[a * 3 + 0] * mean[0] + C[a * 3 + 1] * mean[1]) + C[a * 3 + 2] * mean[2];
    g.cov[0] = C[0]; g.cov[1] = C[1]; g.cov[2] = C[2]; g.cov[3] = C[4]; g.cov[4] = C[5]; g.cov[5] = C[8];
  } else {
#pragma unroll
    for (int a = 0; a < 3; a++) g.mean[a] = mean[a] / g.n;
    g.cov[0] = cov[0] / g.n; g.cov[1] = cov[1] / g.n; g.cov[2] = cov[2] / g.n; g.cov[3] = cov[4] / g.n; g.cov[4] = cov[5] / g.n; g.cov[5] = cov[8] / g.n;
  }
  out[v] = g;
}

__constant__ int8_t c_vg_direct7[7][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {-1, 0, 0, 0}, {0, 1, 0, 0}, {0, -1, 0, 0}, {0, 0, 1, 0}, {0, 0, -1, 0}};

// voxel index of cell (vx,vy,vz) or -1
__device__ inline int vg_lookup(const TargetView& tg, int vx, int vy, int vz) {
  const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
  const uint64_t key = pack_brick(bx, by, bz);
  uint32_t h = hash_coord(bx, by, bz) & tg.mask;
  uint4 s;
  for (;;) {
    s = gload4u(&tg.bricks[h]);
    const uint64_t sk = slot_key(s);
    if (sk == key) break;
    if (sk == kEmptyKey) return -1;
    h = (h + 1) & tg.mask;
  }
  const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
  const uint32_t m = gload_u(&tg.bmask[(size_t)h * 16 + w]);
  if (!((m >> bit) & 1u)) return -1;
  return (int)(s.z + gload_u16(&tg.bpref[(size_t)h * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u)));
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline void load6(const double* p, double (&C)[9]) {
  const double a = gload_d(p), b = gload_d(p + 1), c = gload_d(p + 2), d = gload_d(p + 3), e = gload_d(p + 4), f = gload_d(p + 5);
  C[0] = a; C[1] = b; C[2] = c; C[3] = b; C[4] = d; C[5] = e; C[6] = c; C[7] = e; C[8] = f;
}

// ---------------------------------------------------------------------------
// k_gicp: TRIAL = false -> correspondences + Mahalanobis matrices at x0, H, b, cost (linearize)
//         TRIAL = true  -> cost of the remembered correspondences / matrices at xi (compute_error)
// grid = (blocks, npairs), block = 256
// ---------------------------------------------------------------------------
template <bool VG, bool TRIAL>
__global__ void __launch_bounds__(256) k_gicp(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp) {
  const int pair = PCM_PAIR_OF(kp, blockIdx.y);
  const int mode = states[pair].mode;
  if (mode != (TRIAL ? MODE_TRIAL : MODE_LINEARIZE)) return;
  const PairDesc d = descs[pair];
  const uint32_t per = (uint32_t)(TRIAL ? kp.points_per_block : kp.lin_points_per_block);
  const uint32_t begin = blockIdx.x * per;
  if (begin >= d.src.num_points) return;
  uint32_t end = begin + per;
  end = end < d.src.num_points ? end : d.src.num_points;
  const TargetView tg = d.tgt;
  const int nO = VG ? kp.num_neighbors : 1;
  double T[12];
  float Rf[9], tf[3];
  {
    const double* Ts = TRIAL ? states[pair].xi : states[pair].x0;
#pragma unroll
    for (int a = 0; a < 12; a++) T[a] = Ts[a];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int j = 0; j < 3; j++) Rf[i * 3 + j] = (float)T[i * 4 + j];   // trans.cast<float>()
      tf[i] = (float)T[i * 4 + 3];
    }
  }
  double acc[kNumSums];
#pragma unroll
  for (int j = 0; j < kNumSums; j++) acc[j] = 0.0;

  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    const float4 p = gload4(d.src.pts + i);
    const double pa[3] = {(double)p.x, (double)p.y, (double)p.z};
    double q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) q[a] = T[a * 4 + 0] * pa[0] + T[a * 4 + 1] * pa[1] + T[a * 4 + 2] * pa[2] + T[a * 4 + 3];
    int c[3] = {0, 0, 0};
    bool inrange = true;
    if (VG && !TRIAL) {   // voxel_coord(trans * mean_A)  fast_vgicp_impl.hpp:88
      const double res = (double)tg.res, lim = (double)(kCoordBias - 32);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const double f = floor(q[a] / res - 0.5);
        inrange = inrange && fabs(f) < lim;
        c[a] = inrange ? (int)f : 0;
      }
    }
    double CA[9];
    if (!TRIAL) load6(d.src_cov + (size_t)i * 6, CA);
    for (int k = 0; k < nO; k++) {
      int j;
      double* mp = d.maha + ((size_t)i * nO + k) * 6;
      double M[9];
      if (TRIAL) {
        j = *(const PCM_GLOBAL int32_t*)(d.corr + (size_t)i * nO + k);
        if (j < 0) continue;
        load6(mp, M);
      } else {
        if (VG) {
          int ox, oy, oz;
          if (nO == 27) { ox = k / 9 - 1; oy = (k / 3) % 3 - 1; oz = k % 3 - 1; }
          else { ox = c_vg_direct7[k][0]; oy = c_vg_direct7[k][1]; oz = c_vg_direct7[k][2]; }
          j = inrange ? vg_lookup(tg, c[0] + ox, c[1] + oy, c[2] + oz) : -1;
        } else {
          float qf[3];   // pt = trans_f * p   fast_gicp_impl.hpp:128-131
#pragma unroll
          for (int a = 0; a < 3; a++) qf[a] = (Rf[a * 3 + 0] * p.x + Rf[a * 3 + 1] * p.y) + Rf[a * 3 + 2] * p.z + tf[a];
          j = nearest1(tg, kp.coord_mode, qf, kp.max_corr_sq);
        }
        *(PCM_GLOBAL int32_t*)(d.corr + (size_t)i * nO + k) = j;
        if (j < 0) continue;
        // RCR = cov_B + T cov_A T^T ; M = RCR^-1   fast_gicp_impl.hpp:146-150
        double CB[9], RC[9], S[9];
        load6(VG ? d.vvox[j].cov : d.tgt_cov + (size_t)j * 6, CB);
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
          for (int b = 0; b < 3; b++) RC[a * 3 + b] = T[a * 4 + 0] * CA[0 * 3 + b] + T[a * 4 + 1] * CA[1 * 3 + b] + T[a * 4 + 2] * CA[2 * 3 + b];
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
          for (int b = 0; b < 3; b++) S[a * 3 + b] = CB[a * 3 + b] + (RC[a * 3 + 0] * T[b * 4 + 0] + RC[a * 3 + 1] * T[b * 4 + 1] + RC[a * 3 + 2] * T[b * 4 + 2]);
        }
        {   // the reference inverts the 4 x 4 matrix with (3,3) = 1: Eigen's pair-of-doubles 4x4 inverse (dev_linalg.h)
          double S4[16], I4[16];
#pragma unroll
          for (int a = 0; a < 16; a++) S4[a] = 0.0;
#pragma unroll
          for (int a = 0; a < 3; a++) {
#pragma unroll
            for (int b = 0; b < 3; b++) S4[a * 4 + b] = S[a * 3 + b];
          }
          S4[15] = 1.0;
          inv4d(S4, I4);
#pragma unroll
          for (int a = 0; a < 3; a++) {
#pragma unroll
            for (int b = 0; b < 3; b++) M[a * 3 + b] = I4[a * 4 + b];
          }
        }
        M[3] = M[1]; M[6] = M[2]; M[7] = M[5];   // stored symmetric (upper triangle; the two triangles differ by rounding only)
        gstore_d(mp + 0, M[0]); gstore_d(mp + 1, M[1]); gstore_d(mp + 2, M[2]);
        gstore_d(mp + 3, M[4]); gstore_d(mp + 4, M[5]); gstore_d(mp + 5, M[8]);
      }
      double mb[3], w = 1.0;
      if (VG) {
        const VgVoxel* v = d.vvox + j;
        mb[0] = gload_d(&v->mean[0]); mb[1] = gload_d(&v->mean[1]); mb[2] = gload_d(&v->mean[2]);
        w = sqrt((double)*(const PCM_GLOBAL int32_t*)&v->n);   // fast_vgicp_impl.hpp:149
      } else {
        const float4 m4 = gload4(tg.pts + j);
        mb[0] = (double)m4.x; mb[1] = (double)m4.y; mb[2] = (double)m4.z;
      }
      double e[3], Me[3];
#pragma unroll
      for (int a = 0; a < 3; a++) e[a] = mb[a] - q[a];
#pragma unroll
      for (int a = 0; a < 3; a++) Me[a] = M[a * 3 + 0] * e[0] + M[a * 3 + 1] * e[1] + M[a * 3 + 2] * e[2];
      acc[27] += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
      acc[28] += 1.0;
      if (TRIAL) continue;
      // J = [skew(T p), -I]; H += w J^T M J ; b += w J^T M e
      const double J[3][6] = {{0.0, -q[2], q[1], -1.0, 0.0, 0.0}, {q[2], 0.0, -q[0], 0.0, -1.0, 0.0}, {-q[1], q[0], 0.0, 0.0, 0.0, -1.0}};
      double JtM[6][3];
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) JtM[r][cc] = J[0][r] * M[0 * 3 + cc] + J[1][r] * M[1 * 3 + cc] + J[2][r] * M[2 * 3 + cc];
      }
      int tt = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int cc = r; cc < 6; cc++) { acc[tt] += w * (JtM[cc][0] * J[0][r] + JtM[cc][1] * J[1][r] + JtM[cc][2] * J[2][r]); tt++; }   // entry (cc, r): the lower triangle, which the reference's LDLT reads
      }
#pragma unroll
      for (int r = 0; r < 6; r++) acc[21 + r] += w * (JtM[r][0] * e[0] + JtM[r][1] * e[1] + JtM[r][2] * e[2]);
    }
  }

  __shared__ double s_part[4][kPartialStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < kNumSums; j++) {
    const double v = wave_sum(acc[j]);
    if (lane == 0) s_part[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    const double v = ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) + s_part[3][threadIdx.x];
    gstore_d(d.partials + (size_t)blockIdx.x * kPartialStride + threadIdx.x, v);
  }
}

TargetView view_of(const TargetMap& m) {
  TargetView v{};
  v.pts = m.pts; v.vox_start = m.vox_start; v.bricks = m.bricks; v.bmask = m.bmask; v.bpref = m.bpref; v.gvox = m.gvox;
  v.mask = m.cap - 1; v.num_points = m.num_points; v.inv_res = m.inv_res; v.res = m.res;
  return v;
}

}  // namespace

namespace {
__global__ void k_invert_order(const uint32_t* __restrict__ order, uint32_t n, uint32_t* __restrict__ inv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[order[i]] = i;
}
}  // namespace

// regularised kNN covariance of every point of `map` (map order), 6 doubles each.  `fine`: a second index of the SAME points on a
// finer grid, both built with keep_order -- the kNN search of dense neighbourhoods runs on it (see k_covariances).
int compute_covariances(hipStream_t stream, const TargetMap& map, int k, int regularization, double* d_out, std::string* err, const TargetMap* fine) {   // regularization + 16: CUDA-core float semantics
  if (k < 1 || k > 64) { *err = "k_correspondences must be in [1, 64]"; return PCM_ERR_INVALID_ARGUMENT; }
  const dim3 grid((map.num_points + 63) / 64);   // 64 queries per workgroup
  const dim3 grid256((map.num_points + 255) / 256);
  TargetView tf{};
  uint32_t* c_inv = nullptr;
  const uint32_t* f_order = nullptr;
  if (fine && fine->valid && fine->order && map.order && fine->num_points == map.num_points) {
    if (hipMallocAsync(reinterpret_cast<void**>(&c_inv), sizeof(uint32_t) * (size_t)map.num_points, stream) != hipSuccess) { *err = "hipMallocAsync(c_inv)"; return PCM_ERR_HIP; }
    k_invert_order<<<grid256, 256, 0, stream>>>(map.order, map.num_points, c_inv);
    tf = view_of(*fine);
    f_order = fine->order;
  }
  if (k <= 20) k_covariances<20><<<grid, 256, cov_lds_bytes(20), stream>>>(view_of(map), tf, f_order, c_inv, map.coord_mode, k, regularization, d_out);
  else if (k <= 32) k_covariances<32><<<grid, 256, cov_lds_bytes(32), stream>>>(view_of(map), tf, f_order, c_inv, map.coord_mode, k, regularization, d_out);
  else k_covariances<64><<<grid, 256, cov_lds_bytes(64), stream>>>(view_of(map), tf, f_order, c_inv, map.coord_mode, k, regularization, d_out);
  const hipError_t e = hipGetLastError();
#ifdef PCM_COV_STATS
  {
    unsigned long long h[32];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cov_stats), sizeof(h));
    fprintf(stderr, "cov_stats n %u: waves %llu | fine attempts %llu passes %llu seg-rounds %llu candidates %llu exact lanes %llu | coarse groups %llu passes %llu seg-rounds %llu candidates %llu | per-lane %llu\n",
            map.num_points, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
    fprintf(stderr, "cov_time n %u: search us per wave mean %.1f max %.1f | waves <50us %llu <200 %llu <800 %llu <3200 %llu more %llu | their us %.0f %.0f %.0f %.0f %.0f\n", map.num_points,
            h[0] ? 0.01 * h[11] / h[0] : 0.0, 0.01 * h[12], h[13], h[14], h[15], h[16], h[17], 0.01 * h[18], 0.01 * h[19], 0.01 * h[20], 0.01 * h[21], 0.01 * h[22]);
    fprintf(stderr, "cov_phase n %u: us per workgroup: probe rounds %.1f, staging %.1f, staging+scan %.1f, search+merge %.1f, whole %.1f\n", map.num_points, 0.01 * h[24] / h[0], 0.01 * h[25] / h[0],
            0.01 * h[26] / h[0], 0.01 * h[27] / h[0], 0.01 * h[28] / h[0]);
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cov_stats), z, sizeof(z));
  }
#endif
  if (c_inv) (void)hipFreeAsync(c_inv, stream);
  if (e != hipSuccess) { *err = std::string("k_covariances: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

// ---------------------------------------------------------------------------
// RBF-kernel covariances of the CUDA core: NearestNeighborMethod::GPU_RBF_KERNEL of FastVGICPCuda
// (/root/reference/src/pointcloud_match/fast_gicp/src/fast_gicp/cuda/covariance_estimation_rbf.cu:59-151, selected at
//  include/fast_gicp/gicp/impl/fast_vgicp_cuda_impl.hpp:107,136, constants fast_vgicp_cuda.cu:25-26,205-219).
// Every point x sums over ALL points p of its cloud the weights w = expf(-kernel_width * |x - p|^2) of those within max_dist:
// sum w, sum w p, sum (w p) p^T in float; cov = (sum (w p) p^T - mean (sum w p)^T) / sum w, then covariance_regularization.
// The reference walks the cloud in blocks of 512 points in INPUT order (the last block padded with points at the origin, which
// take part like any other point), one accumulator per (point, block), and adds the block accumulators in block order; that
// order is kept here because float sums are being compared.  What is not kept is the O(N^2) of it: a block of the input order is a
// compact piece of the sensor's sweep (or of a map tile), so a block whose bounding box lies farther than max_dist from every
// point of a workgroup is skipped -- its accumulators are exactly zero in the reference too (the `continue` at :77-79).
// No MFMA: the contraction would be sum_p W[x][p] * F[p][10] with W produced on the fly, in f32 (the 1e-3 eigenvalue floor of the
// regularisation does not survive bf16 moments), and f32 MFMA runs at the vector rate on gfx950 -- the cull is worth 10x, the
// matrix pipe nothing.
// ---------------------------------------------------------------------------
constexpr int kRbfBlock = 512;   // covariance_estimation_kernel::BLOCK_SIZE  :61

// bounding box of every block of the input order (padding points at the origin included for the last block)
__global__ void __launch_bounds__(64) k_rbf_block_bounds(const float4* __restrict__ in, uint32_t n, float* __restrict__ bounds /* [blocks][6] */) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (uint32_t j = lane; j < (uint32_t)kRbfBlock; j += 64) {
    const uint32_t i = b * kRbfBlock + j;
    const float4 p = i < n ? in[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
    hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64)); }
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { bounds[b * 6 + a] = lo[a]; bounds[b * 6 + 3 + a] = hi[a]; }
  }
}

// grid = ceil(M / 256): one lane per point of the MAP order (the covariances are stored in that order), the sums over the INPUT order
__global__ void __launch_bounds__(256) k_covariances_rbf(const float4* __restrict__ map_pts, uint32_t m, const float4* __restrict__ in, uint32_t n, const float* __restrict__ bounds,
                                                         uint32_t nblocks, float exp_factor, float max_dist, int reg, double* __restrict__ out) {
  __shared__ float4 s_p[kRbfBlock];
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  const bool live = p < m;
  const float4 xq = live ? map_pts[p] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float x[3] = {xq.x, xq.y, xq.z};
  const float max_dist_sq = max_dist * max_dist;   // :73
  float sw = 0.f, sm[3] = {0.f, 0.f, 0.f}, sc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // sum over the blocks, in block order (finalization_kernel :108-111)
  for (uint32_t b = 0; b < nblocks; b++) {
    // can any point of this block be within max_dist of this lane's point?  (conservative: a little slack for the float compare)
    float d2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float lo = bounds[b * 6 + a], hi = bounds[b * 6 + 3 + a];
      const float g = fmaxf(fmaxf(lo - x[a], x[a] - hi), 0.f);
      d2 += g * g;
    }
    const bool need = live && d2 <= max_dist_sq * 1.0001f + 1e-12f;
    if (!__syncthreads_or(need ? 1 : 0)) continue;   // the whole workgroup skips the block: its accumulators are zero
#pragma unroll
    for (int r = 0; r < kRbfBlock / 256; r++) {
      const uint32_t j = threadIdx.x + 256u * r, i = b * kRbfBlock + j;
      s_p[j] = i < n ? in[i] : make_float4(0.f, 0.f, 0.f, 0.f);   // padding  :126-129
    }
    __syncthreads();
    float w0 = 0.f, m0[3] = {0.f, 0.f, 0.f}, c0[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // NormalDistribution::zero()
    if (need) {
      for (int j = 0; j < kRbfBlock; j++) {
        const float4 q = s_p[j];
        const float dx = x[0] - q.x, dy = x[1] - q.y, dz = x[2] - q.z;
        const float sq_d = (dx * dx + dy * dy) + dz * dz;          // (x - points[i]).squaredNorm()  :76
        if (sq_d > max_dist_sq) continue;                          // :77-79
        const float w = expf(-exp_factor * sq_d);                  // :81
        const float pj[3] = {q.x, q.y, q.z};
        w0 += w;                                                   // accumulate  :41-45
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const float wp = w * pj[a];
          m0[a] += wp;
#pragma unroll
          for (int c = 0; c < 3; c++) c0[a * 3 + c] += wp * pj[c];
        }
      }
    }
    sw += w0;
#pragma unroll
    for (int a = 0; a < 3; a++) sm[a] += m0[a];
#pragma unroll
    for (int a = 0; a < 9; a++) sc[a] += c0[a];
    __syncthreads();   // the block's points are through before the next block overwrites them
  }
  if (!live) return;
  // finalize  :47-53
  float mean[3], cov[9];
#pragma unroll
  for (int a = 0; a < 3; a++) mean[a] = sm[a] / sw;
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int c = 0; c < 3; c++) cov[a * 3 + c] = (sc[a * 3 + c] - mean[a] * sm[c]) / sw;
  }
  regularize_cov_f(reg, cov);   // covariance_regularization  fast_vgicp_cuda.cu:210,218
  float* o = reinterpret_cast<float*>(out + (size_t)p * 6);   // the float 3x3 in the point's 6-double slot, like k_covariances with reg >= 16
#pragma unroll
  for (int a = 0; a < 9; a++) o[a] = cov[a];
}

int compute_covariances_rbf(hipStream_t stream, const TargetMap& map, const float4* d_input_order, uint32_t n, double kernel_width, double max_dist, int regularization, double* d_out,
                            std::string* err) {
  if (!(kernel_width > 0.0) || !(max_dist > 0.0)) { *err = "rbf_kernel_width and rbf_max_dist must be > 0"; return PCM_ERR_INVALID_ARGUMENT; }
  const uint32_t nblocks = (n + kRbfBlock - 1) / kRbfBlock;
  float* bounds = nullptr;
  if (hipMallocAsync(reinterpret_cast<void**>(&bounds), sizeof(float) * 6 * nblocks, stream) != hipSuccess) { *err = "hipMallocAsync(rbf block bounds)"; return PCM_ERR_HIP; }
  k_rbf_block_bounds<<<nblocks, 64, 0, stream>>>(d_input_order, n, bounds);
  // thrust::device_vector<float> constants: the doubles are narrowed once  :118-121
  k_covariances_rbf<<<(map.num_points + 255) / 256, 256, 0, stream>>>(map.pts, map.num_points, d_input_order, n, bounds, nblocks, (float)kernel_width, (float)max_dist, regularization, d_out);
  const hipError_t e = hipGetLastError();
  (void)hipFreeAsync(bounds, stream);
  if (e != hipSuccess) { *err = std::string("k_covariances_rbf: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

// GaussianVoxelMap::create_voxelmap(points, covariances) of the CUDA core: voxel mean = mean of its points, voxel covariance =
// mean of the point covariances (gaussian_voxelmap.cu:75-169); sums in double, input order (the reference: float atomics)
__global__ void __launch_bounds__(128) k_vgc_voxels(const float4* __restrict__ pts, const uint32_t* __restrict__ vox_start, const double* __restrict__ covs, uint32_t nvox,
                                                    VgcVoxel* __restrict__ out) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t p0 = vox_start[v], p1 = vox_start[v + 1];
  double sx[3] = {0.0, 0.0, 0.0}, sc[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (uint32_t p = p0; p < p1; p++) {
    const float4 c = pts[p];
    sx[0] += (double)c.x; sx[1] += (double)c.y; sx[2] += (double)c.z;
    const float* cf = reinterpret_cast<const float*>(covs + (size_t)p * 6);
#pragma unroll
    for (int a = 0; a < 9; a++) sc[a] += (double)cf[a];
  }
  VgcVoxel g;
  g.n = (int32_t)(p1 - p0);
#pragma unroll
  for (int a = 0; a < 3; a++) g.mean[a] = (float)(sx[a] / (double)g.n);
#pragma unroll
  for (int a = 0; a < 9; a++) g.cov[a] = (float)(sc[a] / (double)g.n);
  g.pad[0] = g.pad[1] = g.pad[2] = 0.f;
  out[v] = g;
}

int build_vgc_voxels(hipStream_t stream, const TargetMap& map, const double* d_cov, VgcVoxel* d_out, std::string* err) {
  k_vgc_voxels<<<(map.num_voxels + 127) / 128, 128, 0, stream>>>(map.pts, map.vox_start, d_cov, map.num_voxels, d_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { *err = std::string("k_vgc_voxels: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

int build_vgicp_voxels(hipStream_t stream, const TargetMap& map, const double* d_cov, int mode, VgVoxel* d_out, std::string* err) {
  k_vgicp_voxels<<<(map.num_voxels + 127) / 128, 128, 0, stream>>>(map.pts, map.vox_start, d_cov, map.num_voxels, mode, d_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { *err = std::string("k_vgicp_voxels: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

void launch_gicp(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool vgicp, bool trial) {
  dim3 grid((unsigned)(trial ? kp.blocks_per_pair : kp.tiles_per_pair), (unsigned)npairs);
  if (vgicp) {
    if (trial) k_gicp<true, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
    else k_gicp<true, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  } else {
    if (trial) k_gicp<false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
    else k_gicp<false, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  }
}

// ---------------------------------------------------------------------------
// k_fitness: pcl::Registration::getFitnessScore(max_range) on the device.  Every reference call site asks for it right
// after align() (jueying_slam/src/localization.cpp:325-326, mapOptmization.cpp:693,719, fast_gicp/src/align.cpp:63); PCL
// answers with a CPU kd-tree 1-NN of every source point in the target -- tens of milliseconds against a 130 us registration.
// Here: the source point transformed by the FLOAT final transformation (pcl::transformPointCloud), its exact nearest map point
// by the voxel-column box walk of the GICP correspondence search (nearest1), the squared distance compared with max_range
// itself (PCL's own quirk: pcl/registration/impl/registration.hpp getFitnessScore), sum and count in double.
// grid = ceil(n / 256), block = 256; one (sum, count) row per block, added by the host in block order.
// ---------------------------------------------------------------------------
struct FitnessPose { float m[12]; };

__global__ void __launch_bounds__(256) k_fitness(TargetView tg, int coord_mode, const float4* __restrict__ src, uint32_t n, FitnessPose P, double max_range, double* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  double d2sum = 0.0, cnt = 0.0;
  if (i < n) {
    const float4 p = gload4(src + i);
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) q[a] = P.m[a * 4 + 0] * p.x + (P.m[a * 4 + 1] * p.y + (P.m[a * 4 + 2] * p.z + P.m[a * 4 + 3]));
    const int j = nearest1(tg, coord_mode, q, 1.0e300);
    if (j >= 0) {
      const float4 mp = gload4(tg.pts + j);
      const float dx = mp.x - q[0], dy = mp.y - q[1], dz = mp.z - q[2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      if ((double)d2 <= max_range) { d2sum = (double)d2; cnt = 1.0; }
    }
  }
  __shared__ double s_sum[4], s_cnt[4];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { d2sum += __shfl_xor(d2sum, off, 64); cnt += __shfl_xor(cnt, off, 64); }
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = d2sum; s_cnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x + 0] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
    out[2 * blockIdx.x + 1] = ((s_cnt[0] + s_cnt[1]) + s_cnt[2]) + s_cnt[3];
  }
}

// setSourceCovariances / setTargetCovariances (fast_gicp_impl.hpp:93-100): the caller's matrices, input order -> map order
__global__ void __launch_bounds__(256) k_scatter_cov(const uint32_t* __restrict__ order, const double* __restrict__ in6, uint32_t n, double* __restrict__ out6) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  const double* s = in6 + (size_t)order[k] * 6;
#pragma unroll
  for (int a = 0; a < 6; a++) out6[(size_t)k * 6 + a] = s[a];
}
int upload_covariances(hipStream_t stream, const TargetMap& map, const double* h_cov6, double* d_cov, std::string* err) {
  const uint32_t n = map.num_points;
  if (n == 0) return PCM_OK;
  double* tmp = nullptr;
  hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&tmp), sizeof(double) * 6 * (size_t)n, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(tmp, h_cov6, sizeof(double) * 6 * (size_t)n, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) { k_scatter_cov<<<(n + 255u) / 256u, 256, 0, stream>>>(map.order, tmp, n, d_cov); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);   // h_cov6 is the caller's
  if (tmp) (void)hipFreeAsync(tmp, stream);
  if (e != hipSuccess) { *err = std::string("upload_covariances: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

// ---------------------------------------------------------------------------
// pclomp GICP-BFGS, correspondence step of computeTransformation (ndt_omp/include/pclomp/gicp_omp_impl.hpp:405-462): for every
// source point i -- output[i] = guess * input[i] (pcl::transformPointCloud, float), query = transformation_ * output[i] (float
// Matrix4f * Vector4f), exact nearest target point, kept when nn_dist < corr_dist_threshold^2 -- the Mahalanobis matrix
// M = (R C1 R^T + C2)^-1 in double (R = rotation of transformation_ * guess formed in double, :416-421), cast to float.  Written
// per SOURCE INDEX (the covariances live in map order; `order` maps back), then compacted in source order (the reference sorts
// its pairs by source index, :466-472) straight into the functor's four-plane records (gicp_bfgs.hip): no host copy of the set.
// ---------------------------------------------------------------------------
struct BfgsCorrXf { float G[12]; float T[12]; double R[9]; double max_sq; };

__global__ void __launch_bounds__(256) k_bfgs_correspond(TargetView tg, int coord_mode, const float4* __restrict__ src, const uint32_t* __restrict__ src_order, uint32_t n,
                                                         const double* __restrict__ src_cov, const double* __restrict__ tgt_cov, const uint32_t* __restrict__ tgt_order, BfgsCorrXf X,
                                                         uint32_t* __restrict__ flag, float4* __restrict__ rec /* 4 planes of n */, int32_t* __restrict__ tgt_idx) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = src_order[k];
  const float4 p = gload4(src + k);
  float o[3], qf[3];
#pragma unroll
  for (int a = 0; a < 3; a++) o[a] = X.G[a * 4 + 0] * p.x + (X.G[a * 4 + 1] * p.y + (X.G[a * 4 + 2] * p.z + X.G[a * 4 + 3]));      // pcl::transformPointCloud
#pragma unroll
  for (int a = 0; a < 3; a++) qf[a] = ((X.T[a * 4 + 0] * o[0] + X.T[a * 4 + 1] * o[1]) + X.T[a * 4 + 2] * o[2]) + X.T[a * 4 + 3] * 1.f;   // Matrix4f * Vector4f, column by column
  const int j = nearest1(tg, coord_mode, qf, X.max_sq);
  flag[i] = j >= 0 ? 1u : 0u;
  if (j < 0) return;
  const float4 q = gload4(tg.pts + j);
  const double* c1 = src_cov + (size_t)k * 6;
  const double* c2 = tgt_cov + (size_t)j * 6;
  const double C1[9] = {c1[0], c1[1], c1[2], c1[1], c1[3], c1[4], c1[2], c1[4], c1[5]};
  const double C2[9] = {c2[0], c2[1], c2[2], c2[1], c2[3], c2[4], c2[2], c2[4], c2[5]};
  double M[9], tmp[9], inv[9];
  // Matrix3d lazy products, coefficient (a, b) = sum of three terms by the fixed-size tree t0 + (t1 + t2)  (CORE-1)
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) M[a * 3 + b] = X.R[a * 3 + 0] * C1[0 * 3 + b] + (X.R[a * 3 + 1] * C1[1 * 3 + b] + X.R[a * 3 + 2] * C1[2 * 3 + b]);
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) tmp[a * 3 + b] = (M[a * 3 + 0] * X.R[b * 3 + 0] + (M[a * 3 + 1] * X.R[b * 3 + 1] + M[a * 3 + 2] * X.R[b * 3 + 2])) + C2[a * 3 + b];
  inv3<double>(tmp, inv);
  const float Mf[9] = {(float)inv[0], (float)inv[1], (float)inv[2], (float)inv[3], (float)inv[4], (float)inv[5], (float)inv[6], (float)inv[7], (float)inv[8]};
  rec[i] = make_float4(o[0], o[1], o[2], q.x);
  rec[(size_t)n + i] = make_float4(q.y, q.z, Mf[0], Mf[1]);
  rec[2 * (size_t)n + i] = make_float4(Mf[2], Mf[3], Mf[4], Mf[5]);
  rec[3 * (size_t)n + i] = make_float4(Mf[6], Mf[7], Mf[8], 0.f);
  tgt_idx[i] = (int32_t)tgt_order[j];
}

__global__ void __launch_bounds__(256) k_bfgs_compact(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, const float4* __restrict__ rec_in, const int32_t* __restrict__ tgt_idx,
                                                      uint32_t n, uint32_t m, float4* __restrict__ rec_out, int32_t* __restrict__ idx_src, int32_t* __restrict__ idx_tgt) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t d = pos[i];
#pragma unroll
  for (int pl = 0; pl < 4; pl++) rec_out[(size_t)pl * m + d] = rec_in[(size_t)pl * n + i];
  idx_src[d] = (int32_t)i;
  idx_tgt[d] = tgt_idx[i];
}

// returns the number of correspondences through *m_out; records + index lists land in the buffers the caller sized for n
int gicp_bfgs_correspond_device(hipStream_t stream, const TargetMap& tmap, int coord_mode, const TargetMap& smap, const double* src_cov, const double* tgt_cov,
                                const float* guess, const float* transformation, double max_corr_dist, float4* d_records, int32_t* d_idx_src, int32_t* d_idx_tgt, uint32_t* m_out,
                                std::string* err) {
  const uint32_t n = smap.num_points;
  *m_out = 0;
  if (n == 0) return PCM_OK;
  BfgsCorrXf X;
  for (int a = 0; a < 12; a++) { X.G[a] = guess[a]; X.T[a] = transformation[a]; }
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {   // transform_R(i,j) += double(transformation_(i,k)) * double(guess(k,j)), k = 0..3 from zero  (:416-419)
      double v = 0.0;
      for (int k = 0; k < 4; k++) v += (double)transformation[a * 4 + k] * (double)guess[k * 4 + b];
      X.R[a * 3 + b] = v;
    }
  X.max_sq = max_corr_dist * max_corr_dist;
  char* tmp = nullptr;
  void* scan_tmp = nullptr;
  size_t scan_bytes = 0;
  int rc = PCM_OK;
  uint32_t tails[2] = {0, 0};
  TargetView tg{};
  tg.pts = tmap.pts; tg.vox_start = tmap.vox_start; tg.bricks = tmap.bricks; tg.bmask = tmap.bmask; tg.bpref = tmap.bpref; tg.gvox = tmap.gvox;
  tg.mask = tmap.cap - 1; tg.num_points = tmap.num_points; tg.inv_res = tmap.inv_res; tg.res = tmap.res;
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); rc = PCM_ERR_HIP; goto done; } \
  } while (0)
  {
    const size_t b_flag = ((size_t)n * 4 + 255) & ~(size_t)255, b_rec = (size_t)n * 64, b_idx = b_flag;
    CK(hipMallocAsync(reinterpret_cast<void**>(&tmp), 2 * b_flag + b_rec + b_idx, stream));
    uint32_t* flag = reinterpret_cast<uint32_t*>(tmp);
    uint32_t* pos = reinterpret_cast<uint32_t*>(tmp + b_flag);
    float4* rec = reinterpret_cast<float4*>(tmp + 2 * b_flag);
    int32_t* tidx = reinterpret_cast<int32_t*>(tmp + 2 * b_flag + b_rec);
    k_bfgs_correspond<<<(n + 255u) / 256u, 256, 0, stream>>>(tg, coord_mode, smap.pts, smap.order, n, src_cov, tgt_cov, tmap.order, X, flag, rec, tidx);
    CK(hipGetLastError());
    CK(rocprim::exclusive_scan(nullptr, scan_bytes, flag, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    CK(hipMallocAsync(&scan_tmp, scan_bytes, stream));
    CK(rocprim::exclusive_scan(scan_tmp, scan_bytes, flag, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    CK(hipMemcpyAsync(&tails[0], flag + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(&tails[1], pos + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    const uint32_t m = tails[0] + tails[1];
    if (m) {
      k_bfgs_compact<<<(n + 255u) / 256u, 256, 0, stream>>>(flag, pos, rec, tidx, n, m, d_records, d_idx_src, d_idx_tgt);
      CK(hipGetLastError());
    }
    *m_out = m;
  }
done:
  if (tmp) (void)hipFreeAsync(tmp, stream);
  if (scan_tmp) (void)hipFreeAsync(scan_tmp, stream);
  return rc;
#undef CK
}

void launch_fitness(hipStream_t stream, const TargetView& tg, int coord_mode, const float4* src, uint32_t n, const float* T, double max_range, double* d_out) {
  FitnessPose P;
  for (int a = 0; a < 12; a++) P.m[a] = T[a];
  k_fitness<<<(n + 255u) / 256u, 256, 0, stream>>>(tg, coord_mode, src, n, P, max_range, d_out);
}

}  // namespace pcm

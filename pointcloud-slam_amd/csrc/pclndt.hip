// pclndt.hip -- pclomp::NormalDistributionsTransform on the brick voxel hash, gfx950.
//
// Replaces, for the MI355X path (paths relative to
// /root/reference/src/pointcloud_match/ndt_omp/include/pclomp):
//   VoxelGridCovariance::applyFilter (leaf statistics) ... voxel_grid_covariance_omp_impl.hpp:206-368
//   getNeighborhoodAtPoint{,7,1} .......................... voxel_grid_covariance_omp_impl.hpp:373-442
//   computeDerivatives + updateDerivatives ................ ndt_omp_impl.hpp:168-267, 451-495
//   computePointDerivatives (float and double) ............ ndt_omp_impl.hpp:369-448
//   computeHessian + updateHessian ........................ ndt_omp_impl.hpp:498-590
// Shape (not a port): the reference keeps a std::map<size_t, Leaf> over a dense linear index,
// builds it serially (seconds at 10 M points) and stores score, gradient and Hessian of ALL N
// points (N x 344 B) before summing them serially.  Here the leaves are a payload of the brick
// hash (built by one radix sort; a voxel's points are one run in input order, summed in double by
// one lane: deterministic), and a derivatives pass is one streaming kernel: <= 27 leaf lookups per
// point, the float inner products exactly as updateDerivatives writes them, 43 sums (36 Hessian,
// 6 gradient, score) accumulated per lane in double and reduced per workgroup, then by one
// small kernel in fixed order.  The Newton / More-Thuente control flow stays on the host
// (pclndt_host.h): it is a serial decision per evaluation, one 48-double read-back each.
// Compiled with -ffp-contract=off.
#include "pcm_device.h"
#include "pcm_host.h"
#include "dev_linalg.h"
#include "pclndt_host.h"

namespace pcm {

namespace {

constexpr int kNdtSums = 43;       // 36 H (row-major), 6 gradient, score
constexpr int kNdtStride = 48;

__device__ inline uint64_t slot_key3(const uint4& s) { return ((uint64_t)s.y << 32) | s.x; }

// leaf index of cell (vx,vy,vz) or -1
__device__ inline int leaf_lookup(const TargetView& tg, int vx, int vy, int vz) {
  const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
  const uint64_t key = pack_brick(bx, by, bz);
  uint32_t h = hash_coord(bx, by, bz) & tg.mask;
  uint4 s;
  for (;;) {
    s = gload4u(&tg.bricks[h]);
    const uint64_t sk = slot_key3(s);
    if (sk == key) break;
    if (sk == kEmptyKey) return -1;
    h = (h + 1) & tg.mask;
  }
  const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
  const uint32_t m = gload_u(&tg.bmask[(size_t)h * 16 + w]);
  if (!((m >> bit) & 1u)) return -1;
  return (int)(s.z + gload_u16(&tg.bpref[(size_t)h * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u)));
}

// second pass of applyFilter: one lane per voxel, its points in input order   :206-259, 262-366
__global__ void __launch_bounds__(128) k_pclndt_leaves(const float4* __restrict__ pts, const uint32_t* __restrict__ vox_start, uint32_t nvox, PclLeaf* __restrict__ out,
                                                       PclLeafF* __restrict__ out_f) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t p0 = vox_start[v], p1 = vox_start[v + 1];
  // Leaf(): cov_ starts as the IDENTITY and the first pass adds x x^T onto it (voxel_grid_covariance_omp.h:103-110, impl :236)
  double sum[3] = {0.0, 0.0, 0.0}, sxx[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
  float csum[3] = {0.f, 0.f, 0.f};
  for (uint32_t p = p0; p < p1; p++) {
    const float4 c = pts[p];
    csum[0] += c.x; csum[1] += c.y; csum[2] += c.z;   // leaf.centroid += pt (float)
    const double x[3] = {(double)c.x, (double)c.y, (double)c.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      sum[a] += x[a];
#pragma unroll
      for (int b = 0; b < 3; b++) sxx[a * 3 + b] += x[a] * x[b];
    }
  }
  PclLeaf L;
  int n = (int)(p1 - p0);
  L.in_centroids = n >= 6 ? 1 : 0;
  L.pad = 0.f;
#pragma unroll
  for (int a = 0; a < 3; a++) L.centroid[a] = csum[a] / (float)n;
#pragma unroll
  for (int a = 0; a < 3; a++) L.mean[a] = sum[a] / n;
#pragma unroll
  for (int a = 0; a < 9; a++) L.icov[a] = 0.0;
  if (n >= 6) {   // min_points_per_voxel_  voxel_grid_covariance_omp.h:210
    double cov[9];
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int b = 0; b < 3; b++) cov[a * 3 + b] = (sxx[a * 3 + b] - 2 * (sum[a] * L.mean[b])) / n + L.mean[a] * L.mean[b];   // :323
    }
#pragma unroll
    for (int a = 0; a < 9; a++) cov[a] *= (n - 1.0) / n;                                                                        // :324
    double sym[9], w[3], V[9];
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int b = 0; b < 3; b++) sym[a * 3 + b] = cov[(a > b ? a : b) * 3 + (a > b ? b : a)];   // self-adjoint view: lower triangle
    }
    selfadjoint3(sym, w, V);   // eigensolver.compute(leaf.cov_)  :327 (tridiagonalisation + implicit QR, dev_linalg.h)
    if (w[0] < 0 || w[1] < 0 || w[2] <= 0) {
      n = -1;                                                                                       // :331-335
    } else {
      const double min_ev = 0.01 * w[2];                                                            // min_covar_eigvalue_mult_  .h:211
      if (w[0] < min_ev) {                                                                          // :339-349
        w[0] = min_ev;
        if (w[1] < min_ev) w[1] = min_ev;
        double Vi[9], VW[9];
        inv3<double>(V, Vi);
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
          for (int b = 0; b < 3; b++) VW[a * 3 + b] = V[a * 3 + b] * w[b];
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
          for (int b = 0; b < 3; b++) cov[a * 3 + b] = VW[a * 3 + 0] * Vi[0 * 3 + b] + VW[a * 3 + 1] * Vi[1 * 3 + b] + VW[a * 3 + 2] * Vi[2 * 3 + b];
        }
      }
      inv3<double>(cov, L.icov);
      double mx = -1.7e308, mn = 1.7e308;
#pragma unroll
      for (int a = 0; a < 9; a++) { mx = L.icov[a] > mx ? L.icov[a] : mx; mn = L.icov[a] < mn ? L.icov[a] : mn; }
      if (isinf(mx) || isinf(mn)) n = -1;                                                           // :353-357
    }
  }
  L.n = n;
  out[v] = L;
  PclLeafF F;
#pragma unroll
  for (int a = 0; a < 3; a++) F.mean[a] = L.mean[a];
#pragma unroll
  for (int a = 0; a < 9; a++) F.ci[a] = (float)L.icov[a];
  F.n = n;
  out_f[v] = F;
}

__device__ inline void ndt_offset3(int nO, int k, int& ox, int& oy, int& oz) {
  if (nO == 27) { ox = k / 9 - 1; oy = (k / 3) % 3 - 1; oz = k % 3 - 1; return; }   // pcl::getAllNeighborCellIndices order
  ox = oy = oz = 0;                                                                  // getNeighborhoodAtPoint7  :414-428
  if (k == 1) ox = 1; else if (k == 2) ox = -1; else if (k == 3) oy = 1; else if (k == 4) oy = -1; else if (k == 5) oz = 1; else if (k == 6) oz = -1;
}

// k-th neighbour leaf of the transformed point xt whose cell is (cx,cy,cz), or -1.  num_neighbors 1 / 7 / 27: the DIRECT
// lookups (leaf must hold >= 6 points); 0: KDTREE = radiusSearch(point, resolution) over the centroid cloud
// (voxel_grid_covariance_omp.h:476-505): float squared distance to the leaf's float centroid strictly below radius^2.
__device__ inline int neighbour_leaf(const TargetView& tg, const PclLeaf* leaves, int nn, int k, int cx, int cy, int cz, const float (&xt)[3]) {
  int ox, oy, oz;
  ndt_offset3(nn == 0 ? 27 : nn, k, ox, oy, oz);
  const int v = leaf_lookup(tg, cx + ox, cy + oy, cz + oz);
  if (v < 0) return -1;
  const PclLeaf* L = leaves + v;
  if (nn == 0) {
    if (*(const PCM_GLOBAL int32_t*)&L->in_centroids == 0) return -1;
    const float r2 = (float)((double)tg.res * (double)tg.res);
    float d2 = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) { const float df = xt[a] - *(const PCM_GLOBAL float*)&L->centroid[a]; d2 += df * df; }
    return d2 < r2 ? v : -1;
  }
  return *(const PCM_GLOBAL int32_t*)&L->n >= 6 ? v : -1;   // nr_points >= min_points_per_voxel_  :396
}

// All neighbour leaves of one point at once.  neighbour_leaf() per cell is a chain of four dependent loads (brick slot, mask word,
// prefix, leaf header) and a pass that walks 7 cells x 4 points per lane one after the other is nothing but that chain (a lone
// workgroup needed 110 us for 1 024 points).  Here the brick of the point's own cell is probed once (6 of the 7 / 26 of the 27 cells
// share it unless the cell lies on a brick face), then every cell's mask word, prefix and leaf header are loaded by straight-line
// code -- independent loads, in flight together.  v[0 .. count) = the leaves that take part, in the reference's cell order.
constexpr int kNdtMaxCells = 27;
__device__ inline uint32_t brick_find_ndt(const TargetView& tg, int bx, int by, int bz, uint32_t& vox_base) {
  const uint64_t key = pack_brick(bx, by, bz);
  uint32_t h = hash_coord(bx, by, bz) & tg.mask;
  for (;;) {
    const uint4 s = gload4u(&tg.bricks[h]);
    const uint64_t sk = slot_key3(s);
    if (sk == key) { vox_base = s.z; return h; }
    if (sk == kEmptyKey) { vox_base = 0; return ~0u; }
    h = (h + 1) & tg.mask;
  }
}
__device__ inline int neighbour_leaves(const TargetView& tg, const PclLeaf* __restrict__ leaves, const PclLeafF* __restrict__ leaves_f, const TargetView& nl, int nn, int cx, int cy, int cz,
                                       const float (&xt)[3], int* __restrict__ v /* LDS, stride 256 */) {
  if (nl.pts) {
    // the grid's neighbour-leaf lists (built with a target that is registered against again: neighbour_lists.hip): one probe of the
    // list index, then one contiguous run of (centroid, leaf) entries in the order the cells would have been visited -- the 27-cell
    // searches no longer cost 27 mask words, prefixes and leaf headers per point
    int cnt = 0;
    uint32_t base = 0, np = 0;
    const uint32_t slot = brick_find_ndt(nl, cx >> kBrickShift, cy >> kBrickShift, cz >> kBrickShift, base);
    (void)np;
    if (slot == ~0u) return 0;
    const uint32_t li = local_index(cx, cy, cz), w = li >> 5, bit = li & 31;
    const uint32_t m = gload_u(&nl.bmask[(size_t)slot * 16 + w]);
    if (!((m >> bit) & 1u)) return 0;
    const uint32_t r = base + gload_u16(&nl.bpref[(size_t)slot * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u));
    const uint32_t s = gload_u(&nl.vox_start[r]), e = gload_u(&nl.vox_start[r + 1]);
    const float r2 = (float)((double)tg.res * (double)tg.res);
    for (uint32_t k = s; k < e; k += 4) {   // padded array: a load past the run is never used
      const float4 e0 = gload4(nl.pts + k), e1 = gload4(nl.pts + k + 1), e2 = gload4(nl.pts + k + 2), e3 = gload4(nl.pts + k + 3);
      const float4 en[4] = {e0, e1, e2, e3};
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (k + u >= e) break;
        bool ok = true;
        if (nn == 0) {   // KDTREE = radiusSearch(point, resolution) over the centroid cloud: float squared distance strictly below radius^2
          float d2 = 0.0f;
          const float df0 = xt[0] - en[u].x; d2 += df0 * df0;
          const float df1 = xt[1] - en[u].y; d2 += df1 * df1;
          const float df2 = xt[2] - en[u].z; d2 += df2 * df2;
          ok = d2 < r2;
        }
        if (ok) { v[cnt * 256] = __float_as_int(en[u].w); cnt++; }
      }
    }
    return cnt;
  }
  const int nk = nn == 0 ? 27 : nn;
  const int cbx = cx >> kBrickShift, cby = cy >> kBrickShift, cbz = cz >> kBrickShift;
  int ch = -1;
  uint32_t cbase = 0;
  {
    const uint64_t key = pack_brick(cbx, cby, cbz);
    uint32_t h = hash_coord(cbx, cby, cbz) & tg.mask;
    for (;;) {
      const uint4 s = gload4u(&tg.bricks[h]);
      const uint64_t sk = slot_key3(s);
      if (sk == key) { ch = (int)h; cbase = s.z; break; }
      if (sk == kEmptyKey) break;
      h = (h + 1) & tg.mask;
    }
  }
  const float r2 = (float)((double)tg.res * (double)tg.res);
  auto one = [&](int k) {
    int ox, oy, oz;
    ndt_offset3(nk, k, ox, oy, oz);
    const int vx = cx + ox, vy = cy + oy, vz = cz + oz;
    int h = ch;
    uint32_t base = cbase;
    if ((vx >> kBrickShift) != cbx || (vy >> kBrickShift) != cby || (vz >> kBrickShift) != cbz) {   // a cell across a brick face
      const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
      const uint64_t key = pack_brick(bx, by, bz);
      uint32_t hh = hash_coord(bx, by, bz) & tg.mask;
      h = -1;
      for (;;) {
        const uint4 s = gload4u(&tg.bricks[hh]);
        const uint64_t sk = slot_key3(s);
        if (sk == key) { h = (int)hh; base = s.z; break; }
        if (sk == kEmptyKey) break;
        hh = (hh + 1) & tg.mask;
      }
    }
    const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
    const size_t word = (size_t)(h < 0 ? 0 : h) * 16 + w;   // a valid address either way: the loads below carry no branch
    const uint32_t m = gload_u(&tg.bmask[word]);
    const uint32_t pref = gload_u16(&tg.bpref[word]);
    const bool present = h >= 0 && ((m >> bit) & 1u);
    const uint32_t leaf = present ? base + pref + (uint32_t)__popc(m & ((1u << bit) - 1u)) : 0u;
    const PclLeaf* L = leaves + leaf;
    bool ok;
    if (nn == 0) {   // KDTREE = radiusSearch(point, resolution) over the centroid cloud
      float d2 = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; a++) { const float df = xt[a] - *(const PCM_GLOBAL float*)&L->centroid[a]; d2 += df * df; }
      ok = *(const PCM_GLOBAL int32_t*)&L->in_centroids != 0 && d2 < r2;
    } else if (leaves_f) {
      ok = *(const PCM_GLOBAL int32_t*)&leaves_f[leaf].n >= 6;   // the line the float pass reads next
    } else {
      ok = *(const PCM_GLOBAL int32_t*)&L->n >= 6;   // nr_points >= min_points_per_voxel_  :396
    }
    return present && ok ? (int)leaf : -1;
  };
  int cnt = 0;
  if (nk == 7) {
    int r[7];
#pragma unroll
    for (int k = 0; k < 7; k++) r[k] = one(k);
#pragma unroll
    for (int k = 0; k < 7; k++) if (r[k] >= 0) { v[cnt * 256] = r[k]; cnt++; }
  } else {
    for (int k0 = 0; k0 < nk; k0 += 9) {
      int r[9];
#pragma unroll
      for (int k = 0; k < 9; k++) r[k] = k0 + k < nk ? one(k0 + k) : -1;
#pragma unroll
      for (int k = 0; k < 9; k++) if (r[k] >= 0) { v[cnt * 256] = r[k]; cnt++; }
    }
  }
  return cnt;
}

__device__ inline double wave_sum3(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int NS>
__device__ inline void block_reduce_store(double (&acc)[NS], double* dst) {
  __shared__ double s_part[4][kNdtStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const double v = wave_sum3(acc[j]);
    if (lane == 0) s_part[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS) gstore_d(dst + threadIdx.x, ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) + s_part[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------
// k_pclndt_derivatives: computeDerivatives with the float inner products of updateDerivatives.
// grid = blocks, block = 256, `per` points per workgroup
// ---------------------------------------------------------------------------
template <bool HESS>
__device__ inline void pclndt_derivatives_body(const TargetView& tg, const PclLeaf* __restrict__ leaves, const PclLeafF* __restrict__ leaves_f, const TargetView& nl, const float4* __restrict__ src, uint32_t n,
                                               uint32_t per, const NdtOmpParams& P,
                                               double* __restrict__ partials, int* __restrict__ s_leaf /* LDS [kNdtMaxCells][256]: the neighbour leaves of every lane's current point */) {
  const uint32_t begin = blockIdx.x * per;
  uint32_t end = begin + per;
  end = end < n ? end : n;
  double acc[kNdtSums];
#pragma unroll
  for (int j = 0; j < kNdtSums; j++) acc[j] = 0.0;
  const float gauss_d2 = (float)P.gauss_d2;
  float4 p_next = make_float4(0.f, 0.f, 0.f, 0.f);
  if (begin + threadIdx.x < end) p_next = gload4(src + begin + threadIdx.x);
  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    const float4 p = p_next;
    if (i + 256 < end) p_next = gload4(src + i + 256);   // the lane's next point is under way while this one is worked on
    // pcl::transformPointCloud with final_transformation_ (float)
    float xt[3];
#pragma unroll
    for (int a = 0; a < 3; a++) xt[a] = P.T[a * 4 + 0] * p.x + (P.T[a * 4 + 1] * p.y + (P.T[a * 4 + 2] * p.z + P.T[a * 4 + 3]));
    const float fx = floorf(xt[0] / tg.res), fy = floorf(xt[1] / tg.res), fz = floorf(xt[2] / tg.res);   // getNeighborhoodAtPoint  :379-381
    const float lim = (float)(kCoordBias - 32);
    if (!(fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim)) continue;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    // computePointDerivatives (float)  :369-412
    float xj[8], xh[15];
#pragma unroll
    for (int r = 0; r < 8; r++) xj[r] = ((P.j_ang[r][0] * p.x + P.j_ang[r][1] * p.y) + P.j_ang[r][2] * p.z) + P.j_ang[r][3] * 0.0f;
    if (HESS) {
#pragma unroll
      for (int r = 0; r < 15; r++) xh[r] = ((P.h_ang[r][0] * p.x + P.h_ang[r][1] * p.y) + P.h_ang[r][2] * p.z) + P.h_ang[r][3] * 0.0f;
    }
    // point_gradient4 (4 x 6): identity block + the 8 angular entries; row 3 is zero
    float pg[3][6];
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int j = 0; j < 6; j++) pg[a][j] = (a == j) ? 1.0f : 0.0f;
    }
    pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
    const int cnt = neighbour_leaves(tg, leaves, leaves_f, nl, P.num_neighbors, cx, cy, cz, xt, s_leaf + threadIdx.x);
    for (int k = 0; k < cnt; k++) {
      // the leaf's 64-byte line (mean in double, (float)icov): four 16-byte loads.  (Fetching the next leaf's line while this one is
      // worked on costs 16 more live registers in a body that is already capped at 255: 55 spilled dwords instead of 15, 2 148 ->
      // 1 755 registrations/s.)
      const uint4* Lp = reinterpret_cast<const uint4*>(leaves_f + s_leaf[k * 256 + threadIdx.x]);   // the lane's own words: no barrier
      const uint4 cu[4] = {gload4u(Lp), gload4u(Lp + 1), gload4u(Lp + 2), gload4u(Lp + 3)};
      const double mean[3] = {__hiloint2double((int)cu[0].y, (int)cu[0].x), __hiloint2double((int)cu[0].w, (int)cu[0].z), __hiloint2double((int)cu[1].y, (int)cu[1].x)};
      const float cif[9] = {__uint_as_float(cu[1].z), __uint_as_float(cu[1].w), __uint_as_float(cu[2].x), __uint_as_float(cu[2].y), __uint_as_float(cu[2].z),
                            __uint_as_float(cu[2].w), __uint_as_float(cu[3].x), __uint_as_float(cu[3].y), __uint_as_float(cu[3].z)};
      float xt4[3], ci[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++) xt4[a] = (float)((double)xt[a] - mean[a]);
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) ci[a][b] = cif[a * 3 + b];
      }
      // every 4-term float sum of the reference carries a fourth term that is an exact zero (x4[3] = 0, row/column 3 of c_inv4 = 0)
      float xc[3];   // x_trans4 * c_inv4
#pragma unroll
      for (int b = 0; b < 3; b++) xc[b] = ((xt4[0] * ci[0][b] + xt4[1] * ci[1][b]) + xt4[2] * ci[2][b]) + 0.0f;
      const float q = ((xt4[0] * xc[0] + xt4[1] * xc[1]) + xt4[2] * xc[2]) + 0.0f;
      float e = expf(-gauss_d2 * q * 0.5f);
      const float score_inc = (float)(-P.gauss_d1 * (double)e);
      e = gauss_d2 * e;
      if (e > 1 || e < 0 || e != e) continue;
      e = (float)((double)e * P.gauss_d1);
      acc[42] += (double)score_inc;
      float cg[3][6];   // c_inv4 * point_gradient4 (row 3 is zero)
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int j = 0; j < 6; j++) cg[a][j] = ((ci[a][0] * pg[0][j] + ci[a][1] * pg[1][j]) + ci[a][2] * pg[2][j]) + 0.0f;
      }
      float xg[6];
#pragma unroll
      for (int j = 0; j < 6; j++) xg[j] = ((xt4[0] * cg[0][j] + xt4[1] * cg[1][j]) + xt4[2] * cg[2][j]) + 0.0f;
#pragma unroll
      for (int j = 0; j < 6; j++) acc[36 + j] += (double)(e * xg[j]);
      if (!HESS) continue;
      float gg[6][6];   // point_gradient4^T * (c_inv4 * point_gradient4)
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int j = 0; j < 6; j++) gg[a][j] = ((pg[0][a] * cg[0][j] + pg[1][a] * cg[1][j]) + pg[2][a] * cg[2][j]) + 0.0f;
      }
      // point_hessian_ blocks (rows 4i..4i+3): only i = 3, 4, 5 and columns 3, 4, 5 are non-zero  :397-411
      //   i=3: a b c ; i=4: b d e ; i=5: c e f   with a = (0, xh0, xh1), b = (0, xh2, xh3), c = (0, xh4, xh5), d = (xh6..8), e = (xh9..11), f = (xh12..14)
      const float va[3] = {0.0f, xh[0], xh[1]}, vb[3] = {0.0f, xh[2], xh[3]}, vc[3] = {0.0f, xh[4], xh[5]}, vd[3] = {xh[6], xh[7], xh[8]}, ve[3] = {xh[9], xh[10], xh[11]},
                  vf[3] = {xh[12], xh[13], xh[14]};
#define PCM_XH(v) (((xc[0] * (v)[0] + xc[1] * (v)[1]) + xc[2] * (v)[2]) + 0.0f)
      const float ha = PCM_XH(va), hb = PCM_XH(vb), hc = PCM_XH(vc), hd = PCM_XH(vd), he = PCM_XH(ve), hf = PCM_XH(vf);
#undef PCM_XH
      // x_trans4_x_c_inv4 * block(i): zero for i < 3 and for columns < 3 (a float sum of four exact zeros)
      const float xh6[6][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, ha, hb, hc}, {0, 0, 0, hb, hd, he}, {0, 0, 0, hc, he, hf}};
#pragma unroll
      for (int i2 = 0; i2 < 6; i2++) {
#pragma unroll
        for (int j = 0; j < 6; j++) acc[i2 * 6 + j] += (double)(e * ((-gauss_d2 * xg[i2] * xg[j] + xh6[i2][j]) + gg[j][i2]));
      }
    }
  }
  block_reduce_store<kNdtSums>(acc, partials + (size_t)blockIdx.x * kNdtStride);
}

// ---------------------------------------------------------------------------
// k_pclndt_hessian: computeHessian / updateHessian in double  :498-590
// ---------------------------------------------------------------------------
__device__ inline void pclndt_hessian_body(const TargetView& tg, const PclLeaf* __restrict__ leaves, const PclLeafF* __restrict__ leaves_f, const TargetView& nl, const float4* __restrict__ src, uint32_t n,
                                           uint32_t per, const NdtOmpParams& P,
                                           double* __restrict__ partials, int* __restrict__ s_leaf) {
  const uint32_t begin = blockIdx.x * per;
  uint32_t end = begin + per;
  end = end < n ? end : n;
  double acc[36];
#pragma unroll
  for (int j = 0; j < 36; j++) acc[j] = 0.0;
  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    const float4 p = gload4(src + i);
    float xt[3];
#pragma unroll
    for (int a = 0; a < 3; a++) xt[a] = P.T[a * 4 + 0] * p.x + (P.T[a * 4 + 1] * p.y + (P.T[a * 4 + 2] * p.z + P.T[a * 4 + 3]));
    const float fx = floorf(xt[0] / tg.res), fy = floorf(xt[1] / tg.res), fz = floorf(xt[2] / tg.res);
    const float lim = (float)(kCoordBias - 32);
    if (!(fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim)) continue;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    const double x[3] = {(double)p.x, (double)p.y, (double)p.z};
    double pg[3][6];
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int j = 0; j < 6; j++) pg[a][j] = (a == j) ? 1.0 : 0.0;
    }
#define PCM_DOT3(v) ((x[0] * (v)[0] + x[1] * (v)[1]) + x[2] * (v)[2])
    pg[1][3] = PCM_DOT3(P.j_ang_d[0]); pg[2][3] = PCM_DOT3(P.j_ang_d[1]); pg[0][4] = PCM_DOT3(P.j_ang_d[2]); pg[1][4] = PCM_DOT3(P.j_ang_d[3]);
    pg[2][4] = PCM_DOT3(P.j_ang_d[4]); pg[0][5] = PCM_DOT3(P.j_ang_d[5]); pg[1][5] = PCM_DOT3(P.j_ang_d[6]); pg[2][5] = PCM_DOT3(P.j_ang_d[7]);
    const double va[3] = {0.0, PCM_DOT3(P.h_ang_d[0]), PCM_DOT3(P.h_ang_d[1])}, vb[3] = {0.0, PCM_DOT3(P.h_ang_d[2]), PCM_DOT3(P.h_ang_d[3])},
                 vc[3] = {0.0, PCM_DOT3(P.h_ang_d[4]), PCM_DOT3(P.h_ang_d[5])}, vd[3] = {PCM_DOT3(P.h_ang_d[6]), PCM_DOT3(P.h_ang_d[7]), PCM_DOT3(P.h_ang_d[8])},
                 ve[3] = {PCM_DOT3(P.h_ang_d[9]), PCM_DOT3(P.h_ang_d[10]), PCM_DOT3(P.h_ang_d[11])}, vf[3] = {PCM_DOT3(P.h_ang_d[12]), PCM_DOT3(P.h_ang_d[13]), PCM_DOT3(P.h_ang_d[14])};
#undef PCM_DOT3
    const int cnt = neighbour_leaves(tg, leaves, leaves_f, nl, P.num_neighbors, cx, cy, cz, xt, s_leaf + threadIdx.x);
    for (int k = 0; k < cnt; k++) {
      const PclLeaf* L = leaves + s_leaf[k * 256 + threadIdx.x];
      double xt3[3], ic[9], cxv[3];
#pragma unroll
      for (int a = 0; a < 3; a++) xt3[a] = (double)xt[a] - gload_d(&L->mean[a]);
#pragma unroll
      for (int a = 0; a < 9; a++) ic[a] = gload_d(&L->icov[a]);
#pragma unroll
      for (int a = 0; a < 3; a++) cxv[a] = (ic[a * 3 + 0] * xt3[0] + ic[a * 3 + 1] * xt3[1]) + ic[a * 3 + 2] * xt3[2];
      double e = P.gauss_d2 * exp(-P.gauss_d2 * ((xt3[0] * cxv[0] + xt3[1] * cxv[1]) + xt3[2] * cxv[2]) / 2);
      if (e > 1 || e < 0 || e != e) continue;
      e *= P.gauss_d1;
      double cg[3][6], xg[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
#pragma unroll
        for (int a = 0; a < 3; a++) cg[a][j] = (ic[a * 3 + 0] * pg[0][j] + ic[a * 3 + 1] * pg[1][j]) + ic[a * 3 + 2] * pg[2][j];
        xg[j] = (xt3[0] * cg[0][j] + xt3[1] * cg[1][j]) + xt3[2] * cg[2][j];
      }
      // x_trans . (c_inv * point_hessian_.block<3,1>(3i, j)): non-zero for i, j in 3..5 only
#define PCM_T2(v) ((xt3[0] * ((ic[0] * (v)[0] + ic[1] * (v)[1]) + ic[2] * (v)[2]) + xt3[1] * ((ic[3] * (v)[0] + ic[4] * (v)[1]) + ic[5] * (v)[2])) + xt3[2] * ((ic[6] * (v)[0] + ic[7] * (v)[1]) + ic[8] * (v)[2]))
      const double ta = PCM_T2(va), tb = PCM_T2(vb), tc = PCM_T2(vc), td = PCM_T2(vd), te = PCM_T2(ve), tf = PCM_T2(vf);
#undef PCM_T2
      const double t2[6][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, ta, tb, tc}, {0, 0, 0, tb, td, te}, {0, 0, 0, tc, te, tf}};
#pragma unroll
      for (int i2 = 0; i2 < 6; i2++) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
          const double t3 = (pg[0][j] * cg[0][i2] + pg[1][j] * cg[1][i2]) + pg[2][j] * cg[2][i2];
          acc[i2 * 6 + j] += e * ((-P.gauss_d2 * xg[i2] * xg[j] + t2[i2][j]) + t3);
        }
      }
    }
  }
  block_reduce_store<36>(acc, partials + (size_t)blockIdx.x * kNdtStride);
}

// one object, the parameters in the kernel arguments (pcm_ndt_derivatives, the host-driven solver)
template <bool HESS>
__global__ void __launch_bounds__(256, 2) k_pclndt_derivatives(TargetView tg, const PclLeaf* __restrict__ leaves, const PclLeafF* __restrict__ leaves_f, TargetView nl, const float4* __restrict__ src, uint32_t n, uint32_t per, NdtOmpParams P,
                                                            double* __restrict__ partials) {
  __shared__ int s_leaf[kNdtMaxCells * 256];
  pclndt_derivatives_body<HESS>(tg, leaves, leaves_f, nl, src, n, per, P, partials, s_leaf);
}
__global__ void __launch_bounds__(256) k_pclndt_hessian(TargetView tg, const PclLeaf* __restrict__ leaves, const PclLeafF* __restrict__ leaves_f, TargetView nl, const float4* __restrict__ src, uint32_t n, uint32_t per, NdtOmpParams P,
                                                        double* __restrict__ partials) {
  __shared__ int s_leaf[kNdtMaxCells * 256];
  pclndt_hessian_body(tg, leaves, leaves_f, nl, src, n, per, P, partials, s_leaf);
}

// ---------------------------------------------------------------------------
// Batched registration: every object's solver (ndtomp::NdtMachine, pclndt_host.h) lives in device memory.  One launch evaluates
// the pass each live object is waiting for (its pose and angle tables are in its machine), the step launch that follows sums the
// workgroup rows in the fixed order of k_pclndt_reduce and advances the machines -- Newton direction, More-Thuente trial, the
// convergence test -- so the host takes no decision between evaluations (it reads one status byte per object, a round behind).
// grid = (max workgroups of an object, objects)
// ---------------------------------------------------------------------------
// two workgroups per CU: the float-Hessian body wants 274 registers, which leaves ONE wave per SIMD; capped at 255 (19 spilled
// dwords) two waves hide each other's leaf loads: 1 593 -> 2 069 registrations/s at config 4 (three waves at 168: 1 287, the spills win)
__global__ void __launch_bounds__(256, 2) k_pclndt_batch_pass(const NdtObject* __restrict__ objs, const ndtomp::NdtMachine* __restrict__ ms) {
  const NdtObject ob = objs[blockIdx.y];
  const int req = ms[blockIdx.y].request;
  if (req < 0 || blockIdx.x >= (uint32_t)ob.nblocks) return;
  const NdtOmpParams& P = ms[blockIdx.y].P;
  __shared__ int s_leaf[kNdtMaxCells * 256];
  if (req == 0) pclndt_derivatives_body<true>(ob.tg, ob.leaves, ob.leaves_f, ob.nl, ob.src, ob.n, ob.per, P, ob.partials, s_leaf);
  else if (req == 1) pclndt_derivatives_body<false>(ob.tg, ob.leaves, ob.leaves_f, ob.nl, ob.src, ob.n, ob.per, P, ob.partials, s_leaf);
  else pclndt_hessian_body(ob.tg, ob.leaves, ob.leaves_f, ob.nl, ob.src, ob.n, ob.per, P, ob.partials, s_leaf);
}

// JacobiSVD<Matrix6d>(H, ComputeFullU | ComputeFullV).solve(b) by the whole workgroup: the arithmetic of pcm::svd_solve6 /
// jacobi_svd<6> (dev_linalg.h) element for element -- every rotation updates 6 independent entries of w (twice), U and V, which
// here are 18 lanes instead of 18 turns of one lane's loop; the 2x2 step and every decision are recomputed by all lanes from the
// same LDS values (uniform control flow, so the barriers are safe).  One lane alone needs ~160 us for a 6 x 6 matrix (a chain of
// dependent double operations), which was most of a solver round.
struct SvdScratch { double W[6][6], U[6][6], V[6][6]; };
__device__ void svd_solve6_block(const double* H, const double* b, double* x, SvdScratch& m) {
  const int t = threadIdx.x;
  const double precision = 2.0 * DBL_EPSILON, considerAsZero = DBL_MIN;
  double scale = 0.0;
  bool finite = true;
  for (int i = 0; i < 36; i++) {
    const double a = fabs(H[i]);
    if (!(a <= DBL_MAX)) finite = false;
    if (a > scale) scale = a;
  }
  if (!finite) {   // InvalidInput: singular values 0 -> rank 0 -> the zero vector
    if (t < 6) x[t] = 0.0;
    __syncthreads();
    return;
  }
  if (scale == 0.0) scale = 1.0;
  if (t < 36) {
    const int i = t / 6, j = t % 6;
    m.W[i][j] = H[t] / scale;
    m.U[i][j] = m.V[i][j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  double maxDiag = 0.0;
  for (int i = 0; i < 6; i++) if (fabs(m.W[i][i]) > maxDiag) maxDiag = fabs(m.W[i][i]);
  bool finished = false;
  while (!finished) {
    finished = true;
    for (int p = 1; p < 6; p++) {
      for (int q = 0; q < p; q++) {
        double thr = precision * maxDiag;
        if (considerAsZero > thr) thr = considerAsZero;
        if (fabs(m.W[p][q]) > thr || fabs(m.W[q][p]) > thr) {
          finished = false;
          double m00 = m.W[p][p], m01 = m.W[p][q], m10 = m.W[q][p], m11 = m.W[q][q];
          double r1c, r1s;
          const double tt = m00 + m11, d = m10 - m01;
          if (fabs(d) < DBL_MIN) { r1s = 0.0; r1c = 1.0; }
          else { const double u = tt / d; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
          plane_rot(m00, m10, r1c, r1s);
          plane_rot(m01, m11, r1c, r1s);
          double jrc, jrs;
          make_jacobi(m00, m01, m11, jrc, jrs);
          const double tc = jrc, ts = -jrs;
          const double jlc = r1c * tc - r1s * ts, jls = r1c * ts + r1s * tc;
          __syncthreads();   // every lane has read the four pivots (and the test above) before anyone rotates them
          if (t < 6) plane_rot(m.W[p][t], m.W[q][t], jlc, jls);
          else if (t < 12) plane_rot(m.U[t - 6][p], m.U[t - 6][q], jlc, jls);
          else if (t < 18) plane_rot(m.V[t - 12][p], m.V[t - 12][q], jrc, -jrs);
          __syncthreads();
          if (t < 6) plane_rot(m.W[t][p], m.W[t][q], jrc, -jrs);
          __syncthreads();
          const double mx = fabs(m.W[p][p]) > fabs(m.W[q][q]) ? fabs(m.W[p][p]) : fabs(m.W[q][q]);
          if (mx > maxDiag) maxDiag = mx;
        }
      }
    }
  }
  if (t == 0) {   // the serial tail of jacobi_svd<6> and svd_solve6: signs, scale, descending order, rank, solve
    double sv[6];
    for (int i = 0; i < 6; i++) {
      const double a = m.W[i][i];
      sv[i] = fabs(a);
      if (a < 0.0) for (int r = 0; r < 6; r++) m.U[r][i] = -m.U[r][i];
    }
    for (int i = 0; i < 6; i++) sv[i] *= scale;
    for (int i = 0; i < 6; i++) {
      int pos = i;
      double mxv = sv[i];
      for (int j = i + 1; j < 6; j++) if (sv[j] > mxv) { mxv = sv[j]; pos = j; }
      if (mxv == 0.0) break;
      if (pos != i) {
        const double tv = sv[i]; sv[i] = sv[pos]; sv[pos] = tv;
        for (int r = 0; r < 6; r++) {
          const double u = m.U[r][pos]; m.U[r][pos] = m.U[r][i]; m.U[r][i] = u;
          const double v = m.V[r][pos]; m.V[r][pos] = m.V[r][i]; m.V[r][i] = v;
        }
      }
    }
    double pre = sv[0] * (6.0 * DBL_EPSILON);
    if (DBL_MIN > pre) pre = DBL_MIN;
    int nz = 6;
    for (int i = 0; i < 6; i++) if (sv[i] == 0.0) { nz = i; break; }
    int rank = nz;
    while (rank > 0 && sv[rank - 1] < pre) rank--;
    double tmp[6];
    for (int j = 0; j < rank; j++) {
      const double p0 = m.U[0][j] * b[0], p1 = m.U[1][j] * b[1], p2 = m.U[2][j] * b[2], p3 = m.U[3][j] * b[3], p4 = m.U[4][j] * b[4], p5 = m.U[5][j] * b[5];
      tmp[j] = (p0 + (p2 + p4)) + (p1 + (p3 + p5));
    }
    for (int j = 0; j < rank; j++) tmp[j] = (1.0 / sv[j]) * tmp[j];
    for (int i = 0; i < 6; i++) {
      double sacc = 0.0;
      if (rank > 0) { sacc = m.V[i][0] * tmp[0]; for (int j = 1; j < rank; j++) sacc = sacc + m.V[i][j] * tmp[j]; }
      x[i] = sacc;
    }
  }
  __syncthreads();
}

// grid = objects, block = 64: ONE wave per object.  The Jacobi rotations of svd_solve6_block use 18 lanes and three barriers each, and a
// barrier that has to collect four waves costs more than the rotation between two of them (256 threads: 62-85 us per round)
__global__ void __launch_bounds__(64) k_pclndt_batch_step(const NdtObject* __restrict__ objs, ndtomp::NdtMachine* __restrict__ ms, unsigned char* __restrict__ flags_row) {
  __shared__ double s_grp[16][kNdtStride];
  __shared__ double s_row[kNdtStride];
  const int o = blockIdx.x;
  if (ms[o].request >= 0) {
    const NdtObject ob = objs[o];
    // 16 row groups x 64 columns (48 used), each group summed in row order, then the groups in order: as k_pclndt_reduce
    for (int idx = threadIdx.x; idx < 1024; idx += 64) {
      const int j = idx & 63, r = idx >> 6;
      if (j < kNdtStride) {
        double v = 0.0;
        for (int b = r; b < ob.nblocks; b += 16) v += gload_d(ob.partials + (size_t)b * kNdtStride + j);
        s_grp[r][j] = v;
      }
    }
    __syncthreads();
    // the Newton direction the step may ask for, by the whole workgroup: H and g of this evaluation (phase LS_HESS: the row holds
    // H only, g is the machine's).  Phase LS_ITER never reaches a Newton step in the same turn.
    __shared__ SvdScratch s_svd;
    __shared__ double s_mg[6], s_delta[6], s_curg[6];
    __shared__ int s_phase;
    if (threadIdx.x < kNdtStride) {
      double t = 0.0;
      for (int k = 0; k < 16; k++) t += s_grp[k][threadIdx.x];
      s_row[threadIdx.x] = t;
    }
    // the machine's phase and gradient are read ONCE, before the barrier: thread 0 rewrites both in ndt_machine_advance below, and a
    // wave that read them from memory after it (phase LS_ITER: no barrier between) would branch into svd_solve6_block's barriers
    // alone (round-2 advisor finding)
    if (threadIdx.x == 0) s_phase = ms[o].phase;
    if (threadIdx.x < 6) s_curg[threadIdx.x] = ms[o].cur.g[threadIdx.x];
    __syncthreads();
    const int phase = s_phase;
    const bool want = phase != ndtomp::NDT_PH_LS_ITER;
    if (want) {
      if (threadIdx.x < 6) s_mg[threadIdx.x] = -(phase == ndtomp::NDT_PH_LS_HESS ? s_curg[threadIdx.x] : s_row[36 + threadIdx.x]);
      __syncthreads();
      svd_solve6_block(s_row, s_mg, s_delta, s_svd);
    }
    if (threadIdx.x == 0) ndtomp::ndt_machine_advance(ms[o], s_row, want ? s_delta : nullptr);
  }
  if (threadIdx.x == 0) __hip_atomic_store(flags_row + o, (unsigned char)(ms[o].request >= 0 ? 1 : 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// k_pclndt_score: calculateScore  :835-880 (double): sum over the neighbour cells of (-d1 e - d3) / #cells
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pclndt_score(TargetView tg, const PclLeaf* __restrict__ leaves, const float4* __restrict__ src, uint32_t n, uint32_t per, NdtOmpParams P,
                                                      double gauss_d3, double* __restrict__ partials) {
  const uint32_t begin = blockIdx.x * per;
  uint32_t end = begin + per;
  end = end < n ? end : n;
  double acc[1] = {0.0};
  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    const float4 p = gload4(src + i);
    float xt[3];
#pragma unroll
    for (int a = 0; a < 3; a++) xt[a] = P.T[a * 4 + 0] * p.x + (P.T[a * 4 + 1] * p.y + (P.T[a * 4 + 2] * p.z + P.T[a * 4 + 3]));
    const float fx = floorf(xt[0] / tg.res), fy = floorf(xt[1] / tg.res), fz = floorf(xt[2] / tg.res);
    const float lim = (float)(kCoordBias - 32);
    if (!(fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim)) continue;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    const int nk = P.num_neighbors == 0 ? 27 : P.num_neighbors;
    int m = 0;
    for (int k = 0; k < nk; k++) m += neighbour_leaf(tg, leaves, P.num_neighbors, k, cx, cy, cz, xt) >= 0 ? 1 : 0;
    if (m == 0) continue;
    double pt = 0.0;
    for (int k = 0; k < nk; k++) {
      const int v = neighbour_leaf(tg, leaves, P.num_neighbors, k, cx, cy, cz, xt);
      if (v < 0) continue;
      const PclLeaf* L = leaves + v;
      double x[3], cxv[3];
#pragma unroll
      for (int a = 0; a < 3; a++) x[a] = (double)xt[a] - gload_d(&L->mean[a]);
#pragma unroll
      for (int a = 0; a < 3; a++) cxv[a] = (gload_d(&L->icov[a * 3 + 0]) * x[0] + gload_d(&L->icov[a * 3 + 1]) * x[1]) + gload_d(&L->icov[a * 3 + 2]) * x[2];
      const double e = exp(-P.gauss_d2 * ((x[0] * cxv[0] + x[1] * cxv[1]) + x[2] * cxv[2]) / 2);
      pt += (-P.gauss_d1 * e - gauss_d3) / m;
    }
    acc[0] += pt;
  }
  block_reduce_store<1>(acc, partials + (size_t)blockIdx.x * kNdtStride);
}

// fixed-order sum of the workgroup rows -> one row
__global__ void __launch_bounds__(1024) k_pclndt_reduce(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  __shared__ double s_grp[16][kNdtStride];
  const int j = threadIdx.x & 63, r = threadIdx.x >> 6;   // 16 row groups x 64 columns (48 used)
  double v = 0.0;
  if (j < kNdtStride) for (int b = r; b < nblocks; b += 16) v += partials[(size_t)b * kNdtStride + j];
  if (j < kNdtStride) s_grp[r][j] = v;
  __syncthreads();
  if (threadIdx.x < kNdtStride) {
    double t = 0.0;
    for (int k = 0; k < 16; k++) t += s_grp[k][threadIdx.x];
    out[threadIdx.x] = t;
  }
}

TargetView view_of2(const TargetMap& m) {
  TargetView v{};
  v.pts = m.pts; v.vox_start = m.vox_start; v.bricks = m.bricks; v.bmask = m.bmask; v.bpref = m.bpref; v.gvox = m.gvox;
  v.mask = m.cap - 1; v.num_points = m.num_points; v.inv_res = m.inv_res; v.res = m.res;
  return v;
}

}  // namespace

int build_pclndt_leaves(hipStream_t stream, const TargetMap& map, PclLeaf* d_out, PclLeafF* d_out_f, std::string* err) {
  k_pclndt_leaves<<<(map.num_voxels + 127) / 128, 128, 0, stream>>>(map.pts, map.vox_start, map.num_voxels, d_out, d_out_f);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { *err = std::string("k_pclndt_leaves: ") + hipGetErrorString(e); return PCM_ERR_HIP; }
  return PCM_OK;
}

void launch_pclndt_batch_round(hipStream_t stream, const NdtObject* d_objs, ndtomp::NdtMachine* d_ms, int nobj, int max_blocks, unsigned char* d_flags_row) {
  k_pclndt_batch_pass<<<dim3((unsigned)max_blocks, (unsigned)nobj), 256, 0, stream>>>(d_objs, d_ms);
  k_pclndt_batch_step<<<nobj, 64, 0, stream>>>(d_objs, d_ms, d_flags_row);
}

NdtObject make_ndt_object(const TargetMap& map, const PclLeaf* leaves, const PclLeafF* leaves_f, const TargetView& nl, const float4* src, uint32_t n, double* d_partials) {
  NdtObject ob{};
  ob.tg = view_of2(map); ob.leaves = leaves; ob.leaves_f = leaves_f; ob.nl = nl; ob.src = src; ob.n = n;
  ob.nblocks = pclndt_workgroups(n, &ob.per);
  ob.partials = d_partials;
  return ob;
}

int pclndt_workgroups(uint32_t n, uint32_t* per_out) {
  // ~100 workgroups per 100k-point scan, 4 points per lane: the 43-sum workgroup reduction is paid once per 1024 points (one point
  // per lane spent more time reducing than evaluating: 1243 -> 1600 registrations/s on 32 objects, profiles/r02_ndt_config4.json).
  // A function of the object's own size only, so a batch sums in the order its objects would alone.
  uint32_t per = ((n / 128 + 255) / 256) * 256;
  per = per < 256 ? 256 : (per > 2048 ? 2048 : per);
  *per_out = per;
  return (int)((n + per - 1) / per);
}

// one derivatives (or Hessian-only) pass; the 48-double result row lands in d_out
void launch_pclndt_pass(hipStream_t stream, const TargetMap& map, const PclLeaf* leaves, const PclLeafF* leaves_f, const TargetView& nl, const float4* src, uint32_t n, const NdtOmpParams& P, int pass, double* d_partials, double* d_out,
                        double gauss_d3) {
  uint32_t per = 0;
  const int nb = pclndt_workgroups(n, &per);
  if (pass == 0) k_pclndt_derivatives<true><<<nb, 256, 0, stream>>>(view_of2(map), leaves, leaves_f, nl, src, n, per, P, d_partials);
  else if (pass == 1) k_pclndt_derivatives<false><<<nb, 256, 0, stream>>>(view_of2(map), leaves, leaves_f, nl, src, n, per, P, d_partials);
  else if (pass == 2) k_pclndt_hessian<<<nb, 256, 0, stream>>>(view_of2(map), leaves, leaves_f, nl, src, n, per, P, d_partials);
  else k_pclndt_score<<<nb, 256, 0, stream>>>(view_of2(map), leaves, src, n, per, P, gauss_d3, d_partials);
  k_pclndt_reduce<<<1, 1024, 0, stream>>>(d_partials, nb, d_out);
}

}  // namespace pcm

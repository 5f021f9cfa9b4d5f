// linearize_flat.hip -- k_linearize_flat: the per-round kernel of the point-to-plane path (gfx950), third form.
//
// Same contract as k_linearize (kernels.hip): for every scan point the 5 nearest map points inside its 27 (1/7/19) neighbour
// voxels in the reference's visit order, the plane through them, the point-to-plane residual / Jacobian row and the 29
// normal-equation sums of the tile -- bit for bit the same neighbour lists, planes and sums.  Replaces, for the MI355X path
// (paths relative to /root/reference/src):
//   LaserMapping::ObsModel matcher loop            jueying_lio/src/laser_mapping.cc:606-637
//   IVox::GetClosestPoint / KNNPointByCondition    jueying_lio/include/ivox3d/ivox3d.h:132-204, ivox3d_node.hpp:140-205
//   common::esti_plane                             jueying_lio/include/common_lib.h:186-243
//   HTH = h_x^T h_x ("J^T J")                      jueying_lio/include/IKFoM_toolkit/esekfom/esekfom.hpp:1687
//
// What changed against k_linearize, and why (counters of the round-2 kernel: half of all wave-cycles were waits, the search
// loop ran 35-45 trips per wave for 19 candidates per lane because the 27 cells were walked one after the other in lock
// step -- sum over cells of the longest run among the lanes -- with one exposed LDS round trip per trip and per cell):
//   1. FLAT CANDIDATE LISTS.  Lanes of a wave sit in a few voxels (median 2 on the bench scans, the scan is Morton-ordered).
//      Every run of consecutive lanes with the same voxel gets ONE list of its candidates (LDS byte offsets of the staged
//      map points, reference order) built by the wave itself: 2 runs x 27 cells per step, a 32-lane prefix sum of the voxels'
//      point counts (the count of a voxel rides in its head point's tag since the map build), no workgroup barrier.
//      The search is then one loop over the list: four candidates per trip, their points fetched together, the next four
//      offsets fetched a trip ahead -- max over lanes of the list length (22 trips on average) instead of the sum over cells of
//      the per-cell maximum, no per-cell branches, no end-of-run test.
//   2. PLANE FITS ARE COMPACTED.  A lane whose five neighbours are the ones of the previous Gauss-Newton iteration, in the
//      same order, re-uses its plane (the plane is a function of the ordered neighbour tuple alone: common_lib.h:186-243 sees
//      only the points).  The other lanes queue their fits in LDS and the first lanes of the workgroup run them, so a tile
//      whose 256 lanes need 100 fits pays two waves of the QR, not four.  The 3- and 4-neighbour (double-precision) fits
//      queue from the other end of the same table.  The memo lives for ONE registration: the first linearize of an align
//      ignores and overwrites it.
//   3. Four workgroups per CU with 128 registers each instead of five with 96 and 8-11 spilled dwords per lane.
//   4. The Jacobian rows go to LDS as doubles (8 conversions per lane instead of 64 in the reduction loop).
// Tiles whose voxel box or staged points do not fit the LDS budget, and runs whose list cannot fit, search the global
// structures per lane exactly as before (knn_global).
//
// Compiled with -ffp-contract=off (see kernels.hip).
#include "pcm_device.h"
#include "pcm_host.h"
#include "plane_fit.h"
#include "linearize_common.h"

namespace pcm {

constexpr int kFCapCells = 2048;     // LDS voxel grid of a tile
constexpr int kFCapPts = 1536;       // staged map points per tile; the last slot holds the point at infinity the lists are padded with
constexpr int kFCapBricks = 64;
constexpr int kFListCap = 512;       // candidate entries per wave and pass
constexpr int kFListStride = kFListCap + 8;   // + the look-ahead read behind the last group
constexpr uint32_t kFPadOff = (uint32_t)(kFCapPts - 1) * 16u;
constexpr uint16_t kFNoCell = 0xffffu;
constexpr uint16_t kFOversize = 0xffffu;      // run length marker: the run's list cannot fit, its lanes search the global structures
// region A of the LDS: cell grid + per-wave lists and run tables while the tile is searched, the fit results afterwards
constexpr int kFOffCnt = kFCapCells * 2;                      // uint8  s_cnt[kFCapCells]
constexpr int kFOffList = kFOffCnt + kFCapCells;              // uint16 s_list[4][kFListStride]
constexpr int kFOffRuns = kFOffList + 4 * kFListStride * 2;   // uint16 s_rcell[4][64], s_rbase[4][64], s_rlen[4][64]
constexpr int kFRegionA = kFOffRuns + 3 * 4 * 64 * 2;
static_assert(kFOffList % 8 == 0 && (kFListStride * 2) % 8 == 0, "candidate groups are read as 8-byte words");
static_assert(256 * 16 <= kFRegionA, "the fit results alias region A");
static_assert(kFPadOff < 65536u, "list entries are 16-bit byte offsets");

// one staged point: a 16-byte read (ds_read_b128 is one LDS pass of 4 cycles per wave, the 12-byte form the compiler would
// pick when .w is unused takes 8)
__device__ inline float4 lds_point(const char* base, uint32_t off) {
  float4 v = *reinterpret_cast<const float4*>(base + off);
  asm volatile("" : "+v"(v.w));
  return v;
}

// TIMING (diagnostic build only): lane 0 of every tile stamps s_memtime at the phase boundaries and adds the differences to
// stats[8..14] (15: tiles); nothing is computed from them.
#define PCMF_STAMP(slot)                                                       \
  if (TIMING) {                                                                \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();             \
    if (threadIdx.x == 0) atomicAdd(&stats[8 + (slot)], t_now - t_prev);       \
    t_prev = t_now;                                                            \
  }

template <bool STATS, bool WRITE_PLANES, bool TIMING = false>
__global__ void __launch_bounds__(256, 4) k_linearize_flat(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp,
                                                           unsigned long long* __restrict__ stats) {
  // XCD-aware placement as in k_linearize: 8 x 8 transposition inside every run of 64 tiles
  uint32_t tile_x = blockIdx.x;
  {
    const uint32_t base = blockIdx.x & ~63u, w = blockIdx.x & 63u;
    if (base + 64u <= gridDim.x) tile_x = base + (w & 7u) * 8u + (w >> 3);
  }
  const int pair = PCM_PAIR_OF(kp, blockIdx.y);
  if (states[pair].mode != MODE_LINEARIZE) return;
  const PairDesc d = descs[pair];
  const uint32_t i = tile_x * 256u + threadIdx.x;
  if (tile_x * 256u >= d.src.num_points) return;
  const bool live = i < d.src.num_points;
  const PoseF P = load_pose(states[pair].x0);
  const TargetView tg = d.tgt;
  // the plane memo is valid from the second linearize of THIS align on (init_state zeroes the counter)
  const bool memo = kp.plane_cache != 0;
  const bool memo_valid = memo && states[pair].num_linearize > 0;

  __shared__ float4 s_pts[kFCapPts];                       // the bricks' map points; .w = index of the point in the map
  __shared__ __align__(16) unsigned char s_a[kFRegionA];
  __shared__ uint16_t s_fjob[256][6];                      // queued fits: 5 staged point indices + the neighbour count
  __shared__ int s_red[4][6];
  __shared__ int s_box[8];
  __shared__ int s_bbox[8];
  __shared__ short4 s_borg[kFCapBricks];
  __shared__ uint32_t s_bps[kFCapBricks];
  __shared__ uint32_t s_boff[kFCapBricks + 1];
  __shared__ int s_goff[32];
  __shared__ uint32_t s_ctr[4];                            // [0] float fits queued, [1] double fits queued, [2] a voxel holds more points than s_cnt can say
  uint16_t* const s_cell = reinterpret_cast<uint16_t*>(s_a);
  uint8_t* const s_cnt = s_a + kFOffCnt;
  float4* const s_fres = reinterpret_cast<float4*>(s_a);   // after the search

  uint32_t n_cand = 0, n_probe = 0;
  unsigned long long t_prev = 0;
  if (TIMING) t_prev = __builtin_amdgcn_s_memtime();
  float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
  float pn_body = 0.f;
  float q[3] = {0.f, 0.f, 0.f};
  int cx = 0, cy = 0, cz = 0;
  bool search = false;
  if (live) {
    p = gload4(d.src.pts + i);
    pn_body = pcm_sqrtf_rn(p.x * p.x + p.y * p.y + p.z * p.z);   // p_body.norm() of the 81 pd2^2 test (laser_mapping.cc:631)
    transform(P, p, q);
    const float fx = roundf(q[0] * tg.inv_res), fy = roundf(q[1] * tg.inv_res), fz = roundf(q[2] * tg.inv_res);  // Pos2Grid  ivox3d.h:283-286
    const float lim = (float)(kCoordBias - 32);
    search = fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim;  // also false for NaN
    if (search) { cx = (int)fx; cy = (int)fy; cz = (int)fz; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the memo of this point: issued here, read behind the search
  uint32_t memo_id[K] = {~0u, ~0u, ~0u, ~0u, ~0u};
  float4 memo_pl = make_float4(0.f, 0.f, 0.f, 0.f);
  if (memo_valid && live) {
#pragma unroll
    for (int j = 0; j < K; j++) memo_id[j] = gload_u(d.nn + (size_t)i * K + j);
    memo_pl = gload4(d.fitcache + i);
  }

  // ---- voxel bounding box of the tile ------------------------------------------------------------
  {
    const int big = 0x3fffffff;
    int mn[3] = {search ? cx : big, search ? cy : big, search ? cz : big};
    int mx[3] = {search ? cx : -big, search ? cy : -big, search ? cz : -big};
#pragma unroll
    for (int a = 0; a < 3; a++) { mn[a] = wave_min_i32(mn[a]); mx[a] = wave_max_i32(mx[a]); }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; a++) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
    }
  }
  if (threadIdx.x < 4) s_ctr[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int ncell = 1;
    bool ok = kp.use_lds != 0;
    for (int a = 0; a < 3; a++) {
      const int mn = min(min(s_red[0][a], s_red[1][a]), min(s_red[2][a], s_red[3][a]));
      const int mx = max(max(s_red[0][3 + a], s_red[1][3 + a]), max(s_red[2][3 + a], s_red[3][3 + a]));
      if (mx < mn) { ok = false; s_box[a] = 0; s_box[3 + a] = 0; continue; }  // no searchable lane in this tile
      const long long dim = (long long)mx - mn + 3;  // +-1 halo for the 27-cell neighbourhood
      s_box[a] = mn - 1;
      s_box[3 + a] = (int)(dim < 4096 ? dim : 4096);
      if (dim > kFCapCells) ok = false;
      ncell = ok ? ncell * (int)dim : ncell;
      if (ncell > kFCapCells) ok = false;
    }
    if (ok) {  // bricks under the box
      int nb = 1;
      for (int a = 0; a < 3; a++) {
        const int blo = s_box[a] >> kBrickShift, bhi = (s_box[a] + s_box[3 + a] - 1) >> kBrickShift;
        s_bbox[a] = blo;
        s_bbox[3 + a] = bhi - blo + 1;
        nb *= bhi - blo + 1;
      }
      s_bbox[6] = nb;
      if (nb > kFCapBricks) ok = false;
    }
    s_box[6] = ncell;
    s_box[7] = ok ? 1 : 0;
  }
  __syncthreads();
  bool use_lds = s_box[7] != 0;   // uniform over the workgroup
  const bool box_ok = use_lds;
  PCMF_STAMP(0)   // load + transform + tile box

  Best best;
  best_init(best, kp.max_range_sq);
  bool oversize = false;   // this lane's run has more candidates than a list holds

  if (use_lds) {
    const int ox0 = s_box[0], oy0 = s_box[1], oz0 = s_box[2];
    const int Dx = s_box[3], Dy = s_box[4], Dz = s_box[5];
    const int bx0 = s_bbox[0], by0 = s_bbox[1], bz0 = s_bbox[2], nby = s_bbox[4], nbz = s_bbox[5], nb = s_bbox[6];
    // ---- one probe per BRICK under the box (wave 0), exclusive scan of their point counts ---------
    if (wave == 0) {
      uint32_t npts = 0, ps = 0;
      if (lane < nb) {
        const int z = lane % nbz, xy = lane / nbz, y = xy % nby, x = xy / nby;
        const uint64_t key = pack_brick(bx0 + x, by0 + y, bz0 + z);
        uint32_t h = hash_coord(bx0 + x, by0 + y, bz0 + z) & tg.mask;
        for (;;) {   // both halves of the 32-byte slot in flight together
          const uint4 s0 = gload4u(&tg.bricks[h]);
          const uint4 s1 = gload4u(reinterpret_cast<const char*>(&tg.bricks[h]) + 16);
          if (STATS) n_probe++;
          const uint64_t sk = slot_key(s0);
          if (sk == key) { ps = s1.x; npts = s1.y; break; }
          if (sk == kEmptyKey) break;
          h = (h + 1) & tg.mask;
        }
        // voxel coordinates of the brick's corner relative to the tile box (|.| < 2048 + 8)
        s_borg[lane] = make_short4((short)(((bx0 + x) << kBrickShift) - ox0), (short)(((by0 + y) << kBrickShift) - oy0), (short)(((bz0 + z) << kBrickShift) - oz0), 0);
      }
      const uint32_t incl = scan_add_wave(npts);
      if (lane < nb) { s_bps[lane] = ps; s_boff[lane] = incl - npts; }
      if (lane == 63) s_boff[kFCapBricks] = incl;   // total
    } else if (wave == 1) {
      // cell offset of each neighbour cell, reference order (ivox3d.h:211-235)
      if (lane < 27) s_goff[lane] = ((int)c_nearby[lane][0] * Dy + (int)c_nearby[lane][1]) * Dz + (int)c_nearby[lane][2];
    }
    // meanwhile everybody clears the cell grid
#pragma unroll
    for (int j = 0; j < kFCapCells / 256; j++) s_cell[threadIdx.x + 256 * j] = kFNoCell;
    __syncthreads();
    const uint32_t total = s_boff[kFCapBricks];
    use_lds = total < (uint32_t)kFCapPts;   // still uniform; the last slot is the point at infinity
    PCMF_STAMP(1)   // brick probes
    if (use_lds) {
      // ---- stage the bricks' map points through LDS: flat, coalesced, all loads in flight ----------
      float4 v[kFCapPts / 256];
      int vb[kFCapPts / 256];
#pragma unroll
      for (int r = 0; r < kFCapPts / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        vb[r] = -1;
        if (k < total) {
          int b = 0;
          while (b + 1 < nb && s_boff[b + 1] <= k) b++;   // nb is small (typically 1..8)
          vb[r] = b;
          const uint32_t gi = s_bps[b] + (k - s_boff[b]);
          v[r] = gload4(tg.pts + gi);
        }
      }
      // a voxel head among the staged points (bit 31 of its tag; the tag also carries the voxel's point count) registers its
      // voxel in the cell grid while the points go to LDS; the tag's place is taken by the point's index in the map
#pragma unroll
      for (int r = 0; r < kFCapPts / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        if (k < total) {
          const int b = vb[r];
          const int tag = __float_as_int(v[r].w);
          const uint32_t gi = s_bps[b] + (k - s_boff[b]);
          s_pts[k] = make_float4(v[r].x, v[r].y, v[r].z, __uint_as_float(gi));
          if (tag < 0) {
            const short4 o = s_borg[b];
            const int li = tag & 511;
            const int x = o.x + (li >> 6), y = o.y + ((li >> 3) & 7), z = o.z + (li & 7);
            if (x >= 0 && x < Dx && y >= 0 && y < Dy && z >= 0 && z < Dz) {
              const uint32_t cnt = ((uint32_t)tag >> 9) & kMaxTagCount;
              const int c = (x * Dy + y) * Dz + z;
              s_cell[c] = (uint16_t)k;
              s_cnt[c] = (uint8_t)(cnt < 255u ? cnt : 255u);
              if (cnt > 255u) s_ctr[2] = 1u;
            }
          }
        }
      }
      if (threadIdx.x == 0) s_pts[kFCapPts - 1] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __uint_as_float(~0u));
      __syncthreads();
      use_lds = s_ctr[2] == 0u;   // uniform: a voxel with more than 255 points sends the tile to the global path
      PCMF_STAMP(2)   // stage map points + cell grid
    }
    if (use_lds) {
      // ---- per wave: runs of lanes with one voxel -> candidate lists -> flat search (no workgroup barrier in here) ----
      uint16_t* const w_list = reinterpret_cast<uint16_t*>(s_a + kFOffList) + wave * kFListStride;
      uint16_t* const w_rcell = reinterpret_cast<uint16_t*>(s_a + kFOffRuns) + wave * 64;
      uint16_t* const w_rbase = w_rcell + 4 * 64;
      uint16_t* const w_rlen = w_rbase + 4 * 64;
      const uint32_t mycell = search ? (uint32_t)(((cx - ox0) * Dy + (cy - oy0)) * Dz + (cz - oz0)) : 0xffffffffu;
      const uint32_t prevcell = __shfl_up(mycell, 1, 64);
      const bool leader = search && (lane == 0 || mycell != prevcell);
      const unsigned long long lmask = __ballot(leader);
      const int nruns = __popcll(lmask);
      // index of the last leader at or below this lane
      const int myrun = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(lmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lmask, 0u)) + (leader ? 1 : 0) - 1;
      if (leader) w_rcell[myrun] = (uint16_t)mycell;
      const int half = lane >> 5, g = lane & 31;
      const bool gact = g < kp.num_neighbors;
      const int goff = gact ? s_goff[g] : 0;
      const char* const pbase = reinterpret_cast<const char*>(s_pts);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int runs_done = 0;
      while (runs_done < nruns) {   // passes: as many runs as the wave's list holds (wave-uniform control flow)
        const int r_first = runs_done;
        uint32_t sbase = 0, smax = 0;
        int r = r_first;
        for (; r < nruns; r += 2) {
          // lanes 0..26 resolve the cells of run r, lanes 32..58 those of run r + 1
          const int rr = r + half;
          uint32_t start = 0, cnt = 0;
          if (gact && rr < nruns) {
            const int c = (int)w_rcell[rr] + goff;
            const uint32_t h = s_cell[c];
            if (h != (uint32_t)kFNoCell) { start = h; cnt = s_cnt[c]; }
          }
          const uint32_t incl = scan_add_half(cnt);
          const uint32_t tot0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31), tot1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          const bool has1 = r + 1 < nruns;
          const uint32_t p0 = (tot0 + 3u) & ~3u, p1 = (tot1 + 3u) & ~3u;   // groups of four
          const bool over0 = p0 > (uint32_t)kFListCap, over1 = has1 && p1 > (uint32_t)kFListCap;
          const uint32_t need0 = over0 ? 0u : p0, need1 = (has1 && !over1) ? p1 : 0u;
          if (sbase + need0 > (uint32_t)kFListCap) break;   // run r opens the next pass (a pass always takes its first run: sbase = 0 there)
          const bool take1 = sbase + need0 + need1 <= (uint32_t)kFListCap;
          const bool over = half ? over1 : over0;
          const bool mine = rr < nruns && !over && (half == 0 || take1);
          const uint32_t rbase = sbase + (half ? need0 : 0u);
          const uint32_t tot = half ? tot1 : tot0, padded = half ? p1 : p0;
          if (mine) {
            const uint32_t pos = rbase + (incl - cnt);
            for (uint32_t j = 0; j < cnt; j++) w_list[pos + j] = (uint16_t)((start + j) << 4);
            if ((uint32_t)g < padded - tot) w_list[rbase + tot + (uint32_t)g] = (uint16_t)kFPadOff;
            if (g == 31) { w_rbase[rr] = (uint16_t)rbase; w_rlen[rr] = (uint16_t)padded; }
          } else if (rr < nruns && over && g == 31) {
            w_rlen[rr] = kFOversize;
          }
          const uint32_t used1 = take1 ? need1 : 0u;
          smax = max(smax, max(need0, used1));
          sbase += need0 + used1;
          if (has1 && !take1) { r += 1; break; }   // run r + 1 opens the next pass
        }
        runs_done = r < nruns ? r : nruns;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        PCMF_STAMP(3)   // runs -> candidate lists
        // ---- flat search of the lanes whose run was listed in this pass ----------------------------
        uint32_t len = 0, lbase = 0;
        if (search && myrun >= r_first && myrun < runs_done) {
          const uint32_t l = w_rlen[myrun];
          if (l == (uint32_t)kFOversize) oversize = true;
          else { len = l; lbase = w_rbase[myrun]; }
        }
        const char* const lptr = reinterpret_cast<const char*>(w_list + lbase);
        uint2 ids = make_uint2(0u, 0u);
        if (len > 0u) ids = *reinterpret_cast<const uint2*>(lptr);
        for (uint32_t t = 0; t < smax; t += 4u) {
          if (t < len) {
            const uint32_t o0 = ids.x & 0xffffu, o1 = ids.x >> 16, o2 = ids.y & 0xffffu, o3 = ids.y >> 16;
            const float4 m0 = lds_point(pbase, o0), m1 = lds_point(pbase, o1), m2 = lds_point(pbase, o2), m3 = lds_point(pbase, o3);
            ids = *reinterpret_cast<const uint2*>(lptr + 2u * (t + 4u));   // the next group (behind the last one: within the stride)
            if (STATS) n_cand += (o0 != kFPadOff) + (o1 != kFPadOff) + (o2 != kFPadOff) + (o3 != kFPadOff);
            best_offer(best, m0, q, o0, kp.max_range_sq);
            best_offer(best, m1, q, o1, kp.max_range_sq);
            best_offer(best, m2, q, o2, kp.max_range_sq);
            best_offer(best, m3, q, o3, kp.max_range_sq);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the next pass rewrites the list
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        PCMF_STAMP(4)   // flat search
      }
#pragma unroll
      for (int j = 0; j < K; j++) best.i[j] >>= 4;
    }
  }
  // lanes without a list: the whole tile (voxel box or staged points beyond the LDS budget) or a run whose candidates cannot fit one
  if (search && (!use_lds || oversize)) {
    best_init(best, kp.max_range_sq);
    knn_global<STATS>(tg, q, cx, cy, cz, kp.num_neighbors, kp.max_range_sq, best, n_cand, n_probe);
    if (use_lds) {   // the fit stage reads staged points: the neighbours lie in the staged bricks, translate their map indices back
#pragma unroll
      for (int j = 0; j < K; j++) {
        uint32_t si = 0;
        if (best.d[j] < __builtin_inff()) {
          int b = 0;
          while (b + 1 < s_bbox[6] && !(best.i[j] >= s_bps[b] && best.i[j] - s_bps[b] < s_boff[b + 1] - s_boff[b])) b++;
          si = s_boff[b] + (best.i[j] - s_bps[b]);
        }
        best.i[j] = si;
      }
    }
  }
  best_finish(best);
  if (!use_lds) {   // tile on the global path: every lane stages its own neighbours, the fit queue below then serves all tiles alike
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (live && j < best.m) {
        float4 mp = gload4(tg.pts + best.i[j]);
        mp.w = __uint_as_float(best.i[j]);
        s_pts[threadIdx.x * K + j] = mp;
        best.i[j] = threadIdx.x * K + j;
      }
    }
  }

  // ---- plane fit on the <= 5 neighbours  (laser_mapping.cc:619-623) --------------------------------
  float4 pl = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
  uint32_t gid[K];
#pragma unroll
  for (int j = 0; j < K; j++) gid[j] = ~0u;
  const bool five = live && best.m == K;
  if (five && memo) {   // identities of the neighbours in the map
#pragma unroll
    for (int j = 0; j < K; j++) gid[j] = __float_as_uint(s_pts[best.i[j]].w);
  }
  bool hit = false;
  if (five && memo_valid) {
    hit = gid[0] == memo_id[0] && gid[1] == memo_id[1] && gid[2] == memo_id[2] && gid[3] == memo_id[3] && gid[4] == memo_id[4];
    if (hit) pl = memo_pl;   // the plane of this ordered tuple (x = NaN: esti_plane rejected it)
  }
  const bool job5 = five && !hit;   // a float fit (exactly five neighbours, common_lib.h:194-208)
  {
    const bool job34 = live && best.m >= KMIN && best.m < K;   // the double path (common_lib.h:210-226)
    const unsigned long long b5 = __ballot(job5), b34 = __ballot(job34);
    uint32_t base5 = 0, base34 = 0;
    if (lane == 0) {
      if (b5) base5 = atomicAdd(&s_ctr[0], (uint32_t)__popcll(b5));
      if (b34) base34 = atomicAdd(&s_ctr[1], (uint32_t)__popcll(b34));
    }
    base5 = (uint32_t)__builtin_amdgcn_readfirstlane((int)base5);
    base34 = (uint32_t)__builtin_amdgcn_readfirstlane((int)base34);
    uint32_t slot = ~0u;
    if (job5) slot = base5 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b5 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b5, 0u));
    if (job34) slot = 255u - (base34 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b34 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b34, 0u)));
    if (slot != ~0u) {   // float fits fill the table from the front, double fits from the back: at most one job per lane, 256 slots
#pragma unroll
      for (int j = 0; j < K; j++) s_fjob[slot][j] = (uint16_t)best.i[j];
      s_fjob[slot][5] = (uint16_t)best.m;
    }
    __syncthreads();   // every wave is through with the cell grid and the lists: region A becomes the result table
    const uint32_t n5 = s_ctr[0], n34 = s_ctr[1];
    if (threadIdx.x < n5 || threadIdx.x >= 256u - n34) {
      const uint32_t job = threadIdx.x;
      const int m = (int)s_fjob[job][5];
      float px[K], py[K], pz[K];
#pragma unroll
      for (int j = 0; j < K; j++) {
        float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < m) mp = s_pts[s_fjob[job][j]];
        px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
      }
      float4 fit;
      bool ok;
      if (threadIdx.x < n5) ok = esti_plane(px, py, pz, K, kp.plane_threshold, &fit);   // the float path, resolved at compile time
      else ok = esti_plane(px, py, pz, m, kp.plane_threshold, &fit);
      if (!ok) fit = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
      s_fres[job] = fit;
    }
    __syncthreads();   // also: nobody reads s_pts after this point (its memory is re-used below)
    if (slot != ~0u) pl = s_fres[slot];
  }
  const bool fitted = job5;   // this lane holds a freshly fitted plane of five neighbours: memoise it
  PCMF_STAMP(5)   // memo check + fit queue + fits
  if (memo && live) {
    if (fitted) {
#pragma unroll
      for (int j = 0; j < K; j++) *(PCM_GLOBAL uint32_t*)(d.nn + (size_t)i * K + j) = gid[j];
      gstore4(d.fitcache + i, pl);
    } else if (!memo_valid && !five) {
      *(PCM_GLOBAL uint32_t*)(d.nn + (size_t)i * K) = ~0u;   // first linearize of the align: nothing older may match
    }
  }

  // ---- residual / Jacobian of this lane's point -> one 8-double row in LDS ---------------------------
  double* const s_row = reinterpret_cast<double*>(s_pts);   // [256][8]: J0..J5, e, selected
  double* const s_grp = s_row + 256 * 8;                    // [8][32] group partials
  {
    float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
      bool sel = !(pl.x != pl.x);
      if (sel) {
        const float pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;  // laser_mapping.cc:627-629
        sel = pn_body > 81.f * pd2 * pd2;                                  // :631
        if (sel) {
          // left-perturbation Jacobian of e = n.(T p) + d :  [ (q x n)^T , n^T ]
          row[0] = q[1] * pl.z - q[2] * pl.y;
          row[1] = q[2] * pl.x - q[0] * pl.z;
          row[2] = q[0] * pl.y - q[1] * pl.x;
          row[3] = pl.x; row[4] = pl.y; row[5] = pl.z;
          row[6] = pd2;
          row[7] = 1.f;
        }
      }
      if (WRITE_PLANES) gstore4(d.planes + i, sel ? pl : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f));   // the selected set, for trial passes / parity hooks
    }
    double2* dst = reinterpret_cast<double2*>(s_row + threadIdx.x * 8);
#pragma unroll
    for (int a = 0; a < 4; a++) dst[a] = make_double2((double)row[2 * a], (double)row[2 * a + 1]);
  }
  __syncthreads();
  // ---- 29 sums over the tile's 256 rows: thread (group g, term j) adds 32 rows in double --------------
  {
    const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
    double v = 0.0;
    if (j < kNumSums) {
      const int ia = c_term_a[j], ib = c_term_b[j];
      const double* r0 = s_row + (g * 32) * 8;
#pragma unroll 8
      for (int k = 0; k < 32; k++) v = fma(r0[k * 8 + ia], r0[k * 8 + ib], v);
    }
    s_grp[g * kPartialStride + j] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    double v = 0.0;
#pragma unroll
    for (int g = 0; g < 8; g++) v += s_grp[g * kPartialStride + threadIdx.x];
    gstore_d(d.partials + (size_t)tile_x * kPartialStride + threadIdx.x, v);
  }
  PCMF_STAMP(6)   // memo store + residual + workgroup reduction
  if (TIMING && threadIdx.x == 0) atomicAdd(&stats[15], 1ull);
  if (STATS) {
    unsigned long long c = n_cand, pr = n_probe;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off, 64); pr += __shfl_xor(pr, off, 64); }
    if (lane == 0) {
      atomicAdd(&stats[0], c);
      atomicAdd(&stats[1], pr);
      if (wave == 0) { atomicAdd(&stats[2], use_lds ? 1ull : 0ull); atomicAdd(&stats[3], 1ull); atomicAdd(&stats[4], box_ok ? 1ull : 0ull); }
    }
  }
}

void launch_linearize_flat(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes,
                           unsigned long long* d_stats, bool timing) {
  dim3 grid((unsigned)kp.tiles_per_pair, (unsigned)npairs);
  if (d_stats && timing) {
    if (write_planes) k_linearize_flat<false, true, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
    else k_linearize_flat<false, false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  } else if (d_stats) {
    if (write_planes) k_linearize_flat<true, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
    else k_linearize_flat<true, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  } else {
    if (write_planes) k_linearize_flat<false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, nullptr);
    else k_linearize_flat<false, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, nullptr);
  }
}

}  // namespace pcm

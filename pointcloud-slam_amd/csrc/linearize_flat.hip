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

#ifndef PCM_FLAT_WG_PER_CU
#define PCM_FLAT_WG_PER_CU 4
#endif
constexpr int kFCapCells = 2048;     // LDS voxel grid of a tile: one word per cell = first staged point | point count << 16
#if PCM_FLAT_WG_PER_CU >= 5
constexpr int kFCapPts = 1216;       // staged map points per tile (5 workgroups per CU: 32 KB each)
#else
constexpr int kFCapPts = 1536;
#endif
constexpr int kFCapBricks = 64;
constexpr int kFBatch = 4;           // candidates of one cell fetched together
constexpr uint32_t kFMaxCount = 0xffffu;   // points per voxel the cell word can say; a tile that sees more searches the global structures

// one staged point: a 16-byte read (ds_read_b128 is one LDS pass of 4 cycles per wave; the 12-byte form the compiler would pick
// when .w is unused takes 8).  keep_w() names .w as an input of an empty asm AFTER the points have been consumed, so the
// wide form is kept without a wait in front of the first use.
__device__ inline float4 lds_point(const char* base, uint32_t off) { return *reinterpret_cast<const float4*>(base + off); }
__device__ inline void keep_w(const float4& a, const float4& b, const float4& c, const float4& d) { asm volatile("" ::"v"(a.w), "v"(b.w), "v"(c.w), "v"(d.w)); }

// TIMING (diagnostic build only): lane 0 of every tile stamps s_memtime at the phase boundaries and adds the differences to
// stats[8..14] (15: tiles); nothing is computed from them.
#define PCMF_STAMP(slot)                                                       \
  if (TIMING) {                                                                \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();             \
    if (threadIdx.x == 0) atomicAdd(&stats[8 + (slot)], t_now - t_prev);       \
    t_prev = t_now;                                                            \
  }

template <bool STATS, bool WRITE_PLANES, bool TIMING = false>
__global__ void __launch_bounds__(256, PCM_FLAT_WG_PER_CU) k_linearize_flat(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp,
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

  __shared__ float4 s_pts[kFCapPts + kFBatch];             // the bricks' map points; .w = index of the point in the map (+ the over-read of a batch)
  __shared__ __align__(16) uint32_t s_cell[kFCapCells];    // cell grid while the tile is searched, the fit results afterwards
  __shared__ uint16_t s_fjob[256][6];                      // queued fits: 5 staged point indices + the neighbour count
  __shared__ int s_red[4][6];
  __shared__ int s_box[8];
  __shared__ int s_bbox[8];
  __shared__ short4 s_borg[kFCapBricks];
  __shared__ uint32_t s_bps[kFCapBricks];
  __shared__ uint32_t s_boff[kFCapBricks + 1];
  __shared__ int s_goff[32];
  __shared__ uint32_t s_ctr[4];                            // [0] float fits queued, [1] double fits queued, [2] a voxel holds more points than a cell word can say
  float4* const s_fres = reinterpret_cast<float4*>(s_cell);   // after the search
  static_assert(256 * sizeof(float4) <= sizeof(uint32_t) * kFCapCells, "the fit results alias the cell grid");

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
    {   // count 0 = no voxel
      uint4* g4 = reinterpret_cast<uint4*>(s_cell);
#pragma unroll
      for (int j = 0; j < kFCapCells / 4 / 256; j++) g4[threadIdx.x + 256 * j] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const uint32_t total = s_boff[kFCapBricks];
    use_lds = total <= (uint32_t)kFCapPts;   // still uniform
    PCMF_STAMP(1)   // brick probes
    if (use_lds) {
      // ---- stage the bricks' map points through LDS: flat, coalesced, all loads in flight ----------
      float4 v[(kFCapPts + 255) / 256];
      int vb[(kFCapPts + 255) / 256];
      int b = 0;   // brick of staged point k: k grows with r, so the brick index only moves forward
#pragma unroll
      for (int r = 0; r < (kFCapPts + 255) / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        vb[r] = -1;
        if (k < total) {
          while (b + 1 < nb && s_boff[b + 1] <= k) b++;   // nb is small (typically 1..8)
          vb[r] = b;
          const uint32_t gi = s_bps[b] + (k - s_boff[b]);
          v[r] = gload4(tg.pts + gi);
        }
      }
      // a voxel head among the staged points (bit 31 of its tag; the tag also carries the voxel's point count) registers its
      // voxel in the cell grid while the points go to LDS; the tag's place is taken by the point's index in the map
#pragma unroll
      for (int r = 0; r < (kFCapPts + 255) / 256; r++) {
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
              s_cell[(x * Dy + y) * Dz + z] = k | ((cnt < kFMaxCount ? cnt : kFMaxCount) << 16);
              if (cnt > kFMaxCount) s_ctr[2] = 1u;
            }
          }
        }
      }
      __syncthreads();
      use_lds = s_ctr[2] == 0u;   // uniform: a voxel with more points than a cell word can say sends the tile to the global path
      PCMF_STAMP(2)   // stage map points + cell grid
    }
    // lane g of every wave holds the offset of neighbour cell g: a scalar per cell through v_readlane.  Read by ALL lanes, outside
    // the divergent search below: v_readlane takes the register of a lane whatever its exec bit, and a lane without a query point
    // (partial last tile, point outside the key range) would otherwise hand over a register it never wrote
    const int goff_l = (use_lds && (lane & 31) < 27) ? s_goff[lane & 31] : 0;
    if (use_lds && search) {
      // ---- per lane: the 27 (1 / 7 / 19) neighbour cells in the reference's order (ivox3d.h:211-235), nine cell words at a time;
      //      every occupied cell hands over up to kFBatch candidates per trip, fetched together (their addresses are start, start + 1, ...:
      //      no pointer chase, no end-of-run test); strict '<' keeps equal distances in visit order.
      const char* const pbase = reinterpret_cast<const char*>(s_pts);
      const int cell0 = ((cx - ox0) * Dy + (cy - oy0)) * Dz + (cz - oz0);
      const int nn = kp.num_neighbors;
      // One cell per trip, the loop NOT unrolled: the search is ~130 instructions that stay in the instruction cache.  (Unrolled
      // over the cells -- 27x in the round-2 kernel, 9x in this kernel's first counted form -- the same work ran 7-12x slower per
      // candidate than a compact loop: profiles/r03_flat_candidate_lists_experiment.txt, r03_search_code_size.txt.)
      uint32_t e_next = s_cell[cell0 + __builtin_amdgcn_readlane(goff_l, 0)];
#pragma unroll 1
      for (int g = 0; g < nn; g++) {
        const uint32_t e = e_next;
        if (g + 1 < nn) e_next = s_cell[cell0 + __builtin_amdgcn_readlane(goff_l, g + 1)];   // the next cell's word, in flight under this cell's candidates
        const uint32_t c = e >> 16;
        if (c) {
          const uint32_t o = (e & 0xffffu) << 4;
          const float4 m0 = lds_point(pbase, o), m1 = lds_point(pbase, o + 16u), m2 = lds_point(pbase, o + 32u), m3 = lds_point(pbase, o + 48u);
          if (STATS) n_cand += c;
          best_offer(best, m0, q, o, kp.max_range_sq);
          if (c > 1u) best_offer(best, m1, q, o + 16u, kp.max_range_sq);
          if (c > 2u) best_offer(best, m2, q, o + 32u, kp.max_range_sq);
          if (c > 3u) best_offer(best, m3, q, o + 48u, kp.max_range_sq);
          keep_w(m0, m1, m2, m3);
#pragma unroll 1
          for (uint32_t j = kFBatch; j < c; j++) best_offer(best, lds_point(pbase, o + 16u * j), q, o + 16u * j, kp.max_range_sq);
        }
      }
#pragma unroll
      for (int j = 0; j < K; j++) best.i[j] >>= 4;
    }
  }
  // tile on the global path (voxel box or staged points beyond the LDS budget)
  if (search && !use_lds) {
    knn_global<STATS>(tg, q, cx, cy, cz, kp.num_neighbors, kp.max_range_sq, best, n_cand, n_probe);
  }
  best_finish(best);
  PCMF_STAMP(4)   // search
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
    if (STATS) {   // plane memo: lanes with five neighbours / lanes that had to fit
      const unsigned long long bf = __ballot(five);
      if (lane == 0) { atomicAdd(&stats[5], (unsigned long long)__popcll(bf)); atomicAdd(&stats[6], (unsigned long long)__popcll(b5)); }
    }
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

  // ---- residual / Jacobian row of every lane, the 29 sums of the tile (the staged points' LDS is free: see the barrier above) ----
  static_assert(sizeof(float4) * kFCapPts >= (size_t)kReduceLdsBytes, "the reduction rows alias the staged points");
  residual_and_reduce<WRITE_PLANES>(d, i, tile_x, live, pl, q, pn_body, s_pts);
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

// linearize_counted.hip -- k_linearize_counted: a second form of the per-round kernel of the point-to-plane path (gfx950), selected
// by PCM_FLAG_COUNTED_SEARCH.  NOT the default: bit-identical results, 14 % fewer vector instructions, a third of the code -- and
// 12 % slower than k_linearize on the bench (9 450 against 10 700 registrations/s, profiles/r03_bench_ab_*.json).  Kept as the
// record of what was tried and as an A/B partner of the default kernel (tests/test_gpu_counted_search.py runs both).
//
// Same contract as k_linearize (kernels.hip): for
// every scan point the 5 nearest map points inside its 27 (1 / 7 / 19) neighbour voxels in the reference's visit order, the plane
// through them, the point-to-plane residual / Jacobian row and the 29 normal-equation sums of the tile -- bit for bit the same
// neighbour lists, planes and sums (tests/test_gpu_counted_search.py).  Replaces, for the MI355X path (paths relative to
// /root/reference/src):
//   LaserMapping::ObsModel matcher loop            jueying_lio/src/laser_mapping.cc:606-637
//   IVox::GetClosestPoint / KNNPointByCondition    jueying_lio/include/ivox3d/ivox3d.h:132-204, ivox3d_node.hpp:140-205
//   common::esti_plane                             jueying_lio/include/common_lib.h:186-243
//   HTH = h_x^T h_x ("J^T J")                      jueying_lio/include/IKFoM_toolkit/esekfom/esekfom.hpp:1687
//
// What differs from k_linearize (whose counters say: 80 % of all wave-cycles are waits, 2 360 vector instructions per wave,
// 8 spilled dwords per lane):
//   1. COUNTED CELLS.  The point count of a voxel rides in its head point's tag since the map build (voxel_hash.hip
//      k_gather_points), so the LDS cell grid holds (first staged point, count) per cell.  A lane then knows every candidate's
//      address up front: no pointer chase to the next staged point, no end-of-run test; the candidates of a cell are fetched
//      four at a time with independent 16-byte LDS reads, the next cell's word is in flight under the current cell's work.
//   2. The loop over the cells is NOT unrolled (27x in k_linearize): one cell per trip, ~130 instructions.
//   3. Wave reductions and prefix sums on the DPP path (linearize_common.h) instead of ds_bpermute shuffles; the brick index of
//      a staged point is carried from round to round of the staging loop instead of searched from zero.
//   4. Jacobian rows go to LDS as doubles (8 conversions per lane instead of 64 in the reduction loop).
//   5. Five workgroups per CU as before (96 registers, 32.1 KB of LDS: the staging tables double as the double-precision fit queue);
//      3 dwords per lane go to scratch on the hot path (k_linearize: 8).
// Why it loses although every single step looks cheaper is not settled: its LDS bank conflicts are three times k_linearize's
// (four 16-byte reads per occupied cell whatever the voxel holds, profiles/r03_pmc_*_B.json), and both kernels wait, they do not
// compute (87 % / 80 % of the wave-cycles are waits).
// Two further steps were built, measured and taken out again (profiles/r03_flat_candidate_lists_experiment.txt,
// profiles/r03_bench_ab_*.json): per-voxel candidate LISTS built by the wave (the search loop over a list is 12x cheaper than the
// cell walk, but building the lists in the kernel costs more than that and skews the waves), and a plane MEMO with a compacted
// fit queue (55 % of the fits answered from the previous Gauss-Newton pass, bit-identical, no gain in time: the kernel waits, it
// does not compute -- and the queue's 3 KB of LDS cost the fifth resident workgroup).
// Tiles whose voxel box or staged points do not fit the LDS budget search the global structures per lane (knn_global).
//
// Compiled with -ffp-contract=off (see kernels.hip).
#include "pcm_device.h"
#include "pcm_host.h"
#include "plane_fit.h"
#include "linearize_common.h"

namespace pcm {

constexpr int kCCapCells = 2048;     // LDS voxel grid of a tile: first staged point (uint16) and point count (uint8) per cell
constexpr int kCCapPts = 1536;       // staged map points per tile
constexpr int kCCapBricks = 64;
constexpr int kCCapJobs = 48;        // queued 3- and 4-neighbour (double-precision) plane fits per round of the queue
constexpr int kCBatch = 4;           // candidates of one cell fetched together
constexpr uint32_t kCMaxCount = 255u;   // points per voxel a cell can say; a tile that sees more searches the global structures
// staging tables (brick origin, first map point, offset in s_pts) until the points are staged; afterwards the same bytes queue the
// double-precision fits
constexpr int kCTblBorg = 0, kCTblBps = kCCapBricks * 8, kCTblBoff = kCTblBps + kCCapBricks * 4, kCTblBytes = kCTblBoff + (kCCapBricks + 1) * 4;
static_assert(kCCapJobs * 4 + kCCapJobs * 16 <= kCTblBytes, "the job queue aliases the staging tables");

// one staged point: a 16-byte read (ds_read_b128 is one LDS pass of 4 cycles per wave; the 12-byte form the compiler would pick
// when .w is unused takes 8).  keep_w() names .w as an input of an empty asm AFTER the points have been consumed, so the
// wide form is kept without a wait in front of the first use.
__device__ inline float4 lds_point(const char* base, uint32_t off) { return *reinterpret_cast<const float4*>(base + off); }
__device__ inline void keep_w(const float4& a, const float4& b, const float4& c, const float4& d) { asm volatile("" ::"v"(a.w), "v"(b.w), "v"(c.w), "v"(d.w)); }

// TIMING (diagnostic build only): lane 0 of every tile stamps s_memtime at the phase boundaries and adds the differences to
// stats[8..14] (15: tiles); nothing is computed from them.
#define PCMC_STAMP(slot)                                                       \
  if (TIMING) {                                                                \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();             \
    if (threadIdx.x == 0) atomicAdd(&stats[8 + (slot)], t_now - t_prev);       \
    t_prev = t_now;                                                            \
  }

template <bool STATS, bool WRITE_PLANES, bool TIMING = false>
__global__ void __launch_bounds__(256, 5) k_linearize_counted(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp,
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

  // 32.6 KB: five workgroups per CU
  __shared__ float4 s_pts[kCCapPts + kCBatch];             // the bricks' map points (+ the over-read of a batch)
  __shared__ uint16_t s_cstart[kCCapCells];                // first staged point of the voxel in that cell
  __shared__ uint8_t s_ccnt[kCCapCells];                   // its point count, 0 = no voxel
  __shared__ __align__(16) unsigned char s_tbl[kCTblBytes];
  __shared__ int s_red[4][6];
  __shared__ int s_box[8];
  __shared__ int s_bbox[8];
  __shared__ int s_goff[32];
  __shared__ uint32_t s_ctr[4];                            // [0] double-precision fits queued, [2] a voxel holds more points than a cell can say
  short4* const s_borg = reinterpret_cast<short4*>(s_tbl + kCTblBorg);       // voxel coordinates of each brick's corner relative to the box origin
  uint32_t* const s_bps = reinterpret_cast<uint32_t*>(s_tbl + kCTblBps);     // first map point of each brick under the box
  uint32_t* const s_boff = reinterpret_cast<uint32_t*>(s_tbl + kCTblBoff);   // its offset in s_pts (exclusive scan of the point counts)
  uint32_t* const s_job = reinterpret_cast<uint32_t*>(s_tbl);                // after the staging: owner tid | m << 16
  uint32_t (*const s_jobid)[4] = reinterpret_cast<uint32_t (*)[4]>(s_tbl + kCCapJobs * 4);   // the 3 or 4 neighbour ids of the job, then its plane

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
      if (dim > kCCapCells) ok = false;
      ncell = ok ? ncell * (int)dim : ncell;
      if (ncell > kCCapCells) ok = false;
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
      if (nb > kCCapBricks) ok = false;
    }
    s_box[6] = ncell;
    s_box[7] = ok ? 1 : 0;
  }
  __syncthreads();
  bool use_lds = s_box[7] != 0;   // uniform over the workgroup
  const bool box_ok = use_lds;
  PCMC_STAMP(0)   // load + transform + tile box

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
      if (lane == 63) s_boff[kCCapBricks] = incl;   // total
    } else if (wave == 1) {
      // cell offset of each neighbour cell, reference order (ivox3d.h:211-235)
      if (lane < 27) s_goff[lane] = ((int)c_nearby[lane][0] * Dy + (int)c_nearby[lane][1]) * Dz + (int)c_nearby[lane][2];
    }
    // meanwhile everybody clears the counts of the cell grid (count 0 = no voxel; the start words need no clearing)
    reinterpret_cast<uint2*>(s_ccnt)[threadIdx.x] = make_uint2(0u, 0u);
    static_assert(kCCapCells == 256 * 8, "one 8-byte store per thread clears the counts");
    __syncthreads();
    const uint32_t total = s_boff[kCCapBricks];
    use_lds = total <= (uint32_t)kCCapPts;   // still uniform
    PCMC_STAMP(1)   // brick probes
    if (use_lds) {
      // ---- stage the bricks' map points through LDS: flat, coalesced, all loads in flight ----------
      float4 v[kCCapPts / 256];
      int vb[kCCapPts / 256];
      int b = 0;   // brick of staged point k: k grows with r, so the brick index only moves forward
#pragma unroll
      for (int r = 0; r < kCCapPts / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        vb[r] = -1;
        if (k < total) {
          while (b + 1 < nb && s_boff[b + 1] <= k) b++;   // nb is small (typically 1..8)
          vb[r] = b;
          v[r] = gload4(tg.pts + s_bps[b] + (k - s_boff[b]));
        }
      }
      // a voxel head among the staged points (bit 31 of its tag; the tag also carries the voxel's point count) registers its
      // voxel in the cell grid while the points go to LDS: one barrier for both
#pragma unroll
      for (int r = 0; r < kCCapPts / 256; r++) {
        const uint32_t k = threadIdx.x + 256u * r;
        if (k < total) {
          s_pts[k] = v[r];
          const int tag = __float_as_int(v[r].w);
          if (tag < 0) {
            const short4 o = s_borg[vb[r]];
            const int li = tag & 511;
            const int x = o.x + (li >> 6), y = o.y + ((li >> 3) & 7), z = o.z + (li & 7);
            if (x >= 0 && x < Dx && y >= 0 && y < Dy && z >= 0 && z < Dz) {
              const uint32_t cnt = ((uint32_t)tag >> 9) & kMaxTagCount;
              const int c = (x * Dy + y) * Dz + z;
              s_cstart[c] = (uint16_t)k;
              s_ccnt[c] = (uint8_t)(cnt < kCMaxCount ? cnt : kCMaxCount);
              if (cnt > kCMaxCount) s_ctr[2] = 1u;
            }
          }
        }
      }
      __syncthreads();
      use_lds = s_ctr[2] == 0u;   // uniform: a voxel with more points than a cell can say sends the tile to the global path
      PCMC_STAMP(2)   // stage map points + cell grid
    }
    // lane g of every wave holds the offset of neighbour cell g: a scalar per cell through v_readlane.  Read by ALL lanes, outside
    // the divergent search below: v_readlane takes the register of a lane whatever its exec bit, and a lane without a query point
    // (partial last tile, point outside the key range) would otherwise hand over a register it never wrote
    const int goff_l = (use_lds && (lane & 31) < 27) ? s_goff[lane & 31] : 0;
    if (use_lds && search) {
      // ---- per lane: the 27 (1 / 7 / 19) neighbour cells in the reference's order (ivox3d.h:211-235), one cell per trip (the loop
      //      is not unrolled), up to kCBatch candidates of a cell fetched together; strict '<' keeps equal distances in visit order
      const char* const pbase = reinterpret_cast<const char*>(s_pts);
      const int cell0 = ((cx - ox0) * Dy + (cy - oy0)) * Dz + (cz - oz0);
      const int nn = kp.num_neighbors;
      int cell = cell0 + __builtin_amdgcn_readlane(goff_l, 0);
      uint32_t c_next = s_ccnt[cell], s_next = s_cstart[cell];
#pragma unroll 1
      for (int g = 0; g < nn; g++) {
        const uint32_t c = c_next, o = s_next << 4;
        if (g + 1 < nn) {   // the next cell's words, in flight under this cell's candidates
          cell = cell0 + __builtin_amdgcn_readlane(goff_l, g + 1);
          c_next = s_ccnt[cell]; s_next = s_cstart[cell];
        }
        if (c) {
          const float4 m0 = lds_point(pbase, o), m1 = lds_point(pbase, o + 16u), m2 = lds_point(pbase, o + 32u), m3 = lds_point(pbase, o + 48u);
          if (STATS) n_cand += c;
          best_offer(best, m0, q, o, kp.max_range_sq);
          if (c > 1u) best_offer(best, m1, q, o + 16u, kp.max_range_sq);
          if (c > 2u) best_offer(best, m2, q, o + 32u, kp.max_range_sq);
          if (c > 3u) best_offer(best, m3, q, o + 48u, kp.max_range_sq);
          keep_w(m0, m1, m2, m3);
#pragma unroll 1
          for (uint32_t j = kCBatch; j < c; j++) best_offer(best, lds_point(pbase, o + 16u * j), q, o + 16u * j, kp.max_range_sq);
        }
      }
#pragma unroll
      for (int j = 0; j < K; j++) best.i[j] >>= 4;
    }
  }
  // tile on the global path (voxel box or staged points beyond the LDS budget)
  if (search && !use_lds) knn_global<STATS>(tg, q, cx, cy, cz, kp.num_neighbors, kp.max_range_sq, best, n_cand, n_probe);
  best_finish(best);
  PCMC_STAMP(4)   // search

  // ---- plane fit on the <= 5 neighbours  (laser_mapping.cc:619-623) --------------------------------
  // 5 neighbours (the float path, almost every lane) is solved in place; the rare 3- and 4-neighbour cases (double path) are
  // queued and solved by the first lanes of the workgroup afterwards, so one straggler does not drag its whole wave through
  // the double-precision QR.
  float4 pl = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
  uint32_t my_job = ~0u;
  if (live) {
    if (best.m == K) {
      float px[K], py[K], pz[K];
#pragma unroll
      for (int j = 0; j < K; j++) {
        const float4 mp = use_lds ? s_pts[best.i[j]] : gload4(tg.pts + best.i[j]);
        px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
      }
      float4 fit;
      if (esti_plane(px, py, pz, K, kp.plane_threshold, &fit)) pl = fit;
    } else if (best.m >= KMIN) {
      my_job = atomicAdd(&s_ctr[0], 1u);
    }
  }
  PCMC_STAMP(5)   // float plane fit
  // the queue holds kCCapJobs fits; a tile with more of them (a scan over nearly empty map) goes round again
  for (uint32_t base = 0;; base += kCCapJobs) {
    if (my_job != ~0u && my_job >= base && my_job < base + kCCapJobs) {
      const uint32_t slot = my_job - base;
      s_job[slot] = threadIdx.x | ((uint32_t)best.m << 16);
#pragma unroll
      for (int j = 0; j < 4; j++) s_jobid[slot][j] = best.i[j];
    }
    __syncthreads();   // also: the float fits are through with s_pts
    const uint32_t njobs = s_ctr[0];   // nobody adds to it any more
    if (base >= njobs) break;
    const uint32_t here = min(njobs - base, (uint32_t)kCCapJobs);
    if (threadIdx.x < here) {
      const uint32_t job = threadIdx.x;
      const uint32_t m = s_job[job] >> 16;
      float px[K], py[K], pz[K];
#pragma unroll
      for (int j = 0; j < K; j++) {
        float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < (int)m && j < 4) mp = use_lds ? s_pts[s_jobid[job][j]] : gload4(tg.pts + s_jobid[job][j]);
        px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
      }
      float4 fit;
      if (!esti_plane(px, py, pz, (int)m, kp.plane_threshold, &fit)) fit.x = __builtin_nanf("");
      s_jobid[job][0] = __float_as_uint(fit.x); s_jobid[job][1] = __float_as_uint(fit.y);   // hand the plane back to its owner lane
      s_jobid[job][2] = __float_as_uint(fit.z); s_jobid[job][3] = __float_as_uint(fit.w);
    }
    __syncthreads();   // also: nobody reads s_pts after this point (its memory is re-used below)
    if (my_job != ~0u && my_job >= base && my_job < base + kCCapJobs) {
      const uint32_t slot = my_job - base;
      pl = make_float4(__uint_as_float(s_jobid[slot][0]), __uint_as_float(s_jobid[slot][1]), __uint_as_float(s_jobid[slot][2]), __uint_as_float(s_jobid[slot][3]));
    }
    if (base + kCCapJobs >= njobs) break;
    __syncthreads();   // the slots are free for the next round of the queue
  }

  // ---- residual / Jacobian row of every lane, the 29 sums of the tile (the staged points' LDS is free: see the barriers above) ----
  static_assert(sizeof(float4) * kCCapPts >= (size_t)kReduceLdsBytes, "the reduction rows alias the staged points");
  residual_and_reduce<WRITE_PLANES>(d, i, tile_x, live, pl, q, pn_body, s_pts);
  PCMC_STAMP(6)   // queued double-precision fits + residual + workgroup reduction
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

void launch_linearize_counted(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes,
                              unsigned long long* d_stats, bool timing) {
  dim3 grid((unsigned)kp.tiles_per_pair, (unsigned)npairs);
  if (d_stats && timing) {
    if (write_planes) k_linearize_counted<false, true, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
    else k_linearize_counted<false, false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  } else if (d_stats) {
    if (write_planes) k_linearize_counted<true, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
    else k_linearize_counted<true, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats);
  } else {
    if (write_planes) k_linearize_counted<false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, nullptr);
    else k_linearize_counted<false, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, nullptr);
  }
}

}  // namespace pcm

// kernels.hip -- the per-round kernels of the point-to-plane scan-to-submap path (gfx950):
//
//   k_linearize      one launch per GN / LM round: 5-NN of every scan point in the brick voxel hash of the
//                    submap, plane fit, point-to-plane residual / Jacobian and the 29 normal-equation sums
//                    of the tile (one partial row per 256-point workgroup); LIO variant: the 12-column IEKF
//                    measurement row and its 92 sums.  The tile kernel: a target's first registration, growing
//                    (sliding) targets and the LIO model; a static target that is registered against again
//                    runs k_linearize_lists (neighbour_lists.hip) on candidate lists built with its map
//   k_finish_round   fixed-order sum of the partial rows + the GN / LM state machine (lsq_step.h), one
//                    status byte per pair into mapped host memory
//   k_trial          LM trial cost on the planes of the last linearize
//
// Replaces, for the MI355X path (paths relative to /root/reference/src):
//   - LaserMapping::ObsModel matcher loop            jueying_lio/src/laser_mapping.cc:606-637
//   - IVox::GetClosestPoint / KNNPointByCondition    jueying_lio/include/ivox3d/ivox3d.h:132-204, ivox3d_node.hpp:140-205
//   - common::esti_plane                             jueying_lio/include/common_lib.h:186-243
//   - HTH = h_x^T h_x  ("J^T J")                      jueying_lio/include/IKFoM_toolkit/esekfom/esekfom.hpp:1687
//   - find_voxel_correspondences + compute_derivatives + transform_reduce of the
//     reference's CUDA path (materialised pair list, 43-float tuple reduction)
//                                                    pointcloud_match/fast_gicp/src/fast_gicp/cuda/{find_voxel_correspondences,compute_derivatives}.cu
//   - LsqRegistration step_gn / step_lm              pointcloud_match/fast_gicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp:105-172
//
// Shape (not a port).  The scan is ordered along a Morton curve once per scan, so the 256 points of a
// workgroup tile sit in a few neighbouring voxels.  The tile's voxel bounding box (+1 halo) is resolved
// against the brick hash ONCE per tile -- one probe per 8x8x8-voxel brick -- and the bricks' map points are
// staged through LDS with coalesced 16-byte loads; every voxel head registers itself in a dense LDS cell
// grid; each lane then runs its 27-cell / 5-NN search and the plane fit out of LDS and registers.  Tiles
// whose box does not fit fall back to per-lane probing of the global structures.  No correspondence list
// is written to HBM; the residual row goes through LDS into the tile's sums (float geometry, double
// accumulation, fixed order).  The GN / LM state stays on the device for the whole align().
// The kernel is VALU-bound: 2 463 vector instructions per wave, 1 672 of them in the
// search loop (which a wave executes for the union of its lanes' needs) and the plane fit behind it.
//
// Compiled with -ffp-contract=off: the float geometry that feeds discrete
// decisions (voxel key, kNN order, plane test) must round like the reference's
// plain x86 build; the double accumulations use explicit fma().
#include "pcm_device.h"
#include "pcm_host.h"
#include "plane_fit.h"

#include "linearize_common.h"

namespace pcm {

// ---------------------------------------------------------------------------
// End of a round, one 256-thread workgroup per pair (k_finish_round below): fixed-order sum of the round's partial rows
// (deterministic: 32 row groups of stride 32, each summed in row order, four groups per thread advancing together; then the
// group totals in group order) + the GN/LM state machine on an LDS copy of the pair's state, or the export of the sums for the
// parity hooks.  The pair's status byte for this round goes straight to mapped pinned host memory (a posted write), which is
// all the host polls.  (Doing this in the last-arriving workgroup of the residual kernel -- PCM_FLAG_FUSED_STEP -- is kept as an
// option and measured slower: every workgroup pays a store drain and a returned atomic.)
// grid = npairs, block = 256
// ---------------------------------------------------------------------------
__device__ inline void unpack_sums(const double* s, double* H, double* b, double* cost, int* inliers) {
  int t = 0;
  for (int a = 0; a < 6; a++) {
    for (int c = a; c < 6; c++) { H[a * 6 + c] = s[t]; H[c * 6 + a] = s[t]; t++; }
  }
  for (int a = 0; a < 6; a++) b[a] = s[21 + a];
  *cost = s[27];
  *inliers = (int)s[28];
}

// Write-through (sc1) accessors of the partial rows for the in-launch hand-off below: an agent-scope relaxed atomic store /
// load of the 8 bytes lowers to global_store_dwordx2 / global_load_dwordx2 with sc1, which writes through the XCD's L2 and
// reads past this CU's L1 (/opt/skills/guides/MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
// visibility": valid forms with `sc1` stores and loads on both sides).
__device__ inline void gstore_d_wt(double* p, double v) {
  __hip_atomic_store((PCM_GLOBAL unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double gload_d_wt(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const PCM_GLOBAL unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// The step of one pair done by the LAST workgroup of its k_linearize launch: the same sums in the same order as
// k_finish_round (its 32 row groups are walked four at a time by 256 threads), so a pair's result does not depend on
// which path ran its rounds.  The rows were stored write-through by their workgroups and are read with sc1 loads.
__device__ inline void finish_pair_in_place(const PairDesc& d, PairState* states, int pair, const LsqParams& lp, int nblocks, unsigned char* flags_row, double* s_grp /* 32 x 32 */,
                                            double* s_tot /* 32 */) {
  // thread (g8, j) owns row groups r = g8, g8 + 8, g8 + 16, g8 + 24 of column j; the rows of a group (b = r, r + 32, ...) are
  // added in that order exactly as k_finish_round does, but the write-through loads pay a fabric round trip each, so 16 rows
  // of two groups are fetched at once (32 loads in flight) before the adds
  const int j = threadIdx.x & 31, g8 = threadIdx.x >> 5;
  const int jj = j < kNumSums ? j : 0;
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
    const int r0 = g8 + 16 * h, r1 = r0 + 8;
    double v0 = 0.0, v1 = 0.0;
#pragma unroll 1
    for (int base = 0; base < nblocks; base += 32 * 16) {
      double x0[16], x1[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int b0 = base + r0 + 32 * k, b1 = base + r1 + 32 * k;
        x0[k] = b0 < nblocks ? gload_d_wt(d.partials + (size_t)b0 * kPartialStride + jj) : 0.0;
        x1[k] = b1 < nblocks ? gload_d_wt(d.partials + (size_t)b1 * kPartialStride + jj) : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if (base + r0 + 32 * k < nblocks) v0 += x0[k];
        if (base + r1 + 32 * k < nblocks) v1 += x1[k];
      }
    }
    s_grp[r0 * kPartialStride + j] = j < kNumSums ? v0 : 0.0;
    s_grp[r1 * kPartialStride + j] = j < kNumSums ? v1 : 0.0;
  }
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    double t = 0.0;
    for (int k = 0; k < 32; k++) t += s_grp[k * kPartialStride + threadIdx.x];
    s_tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    PairState& st = states[pair];
    double H[36], b[6], cost;
    int inl;
    unpack_sums(s_tot, H, b, &cost, &inl);
    after_linearize(st, lp, H, b, cost, inl);
    __hip_atomic_store(flags_row + pair, (unsigned char)(st.mode != MODE_DONE ? 1 : 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void __launch_bounds__(256) k_finish_round(const PairDesc* __restrict__ descs, PairState* __restrict__ states, KernelParams kp, LsqParams lp, int trial_round,
                                                      int write_flags, unsigned char* __restrict__ flags_row, double* __restrict__ sums_out, unsigned int* __restrict__ queue,
                                                      int npairs) {
  // 256 threads, 4.2 KB of LDS: with several batches in flight this launch has to find room on CUs that another stream's search
  // kernel fills (4 x 39.6 KB of LDS, 448 of 512 VGPRs per SIMD).  The 1024-thread / 8.5 KB form of round 1 needed three of the four
  // resident search workgroups of one CU to retire at once and averaged 41 us under the two-slot schedule (13 us alone); this one
  // fits as soon as ONE of them retires.  Same sums in the same order: 32 row groups of stride 32, each summed in row order, then
  // the group totals added in group order.
  const int pair = PCM_PAIR_OF(kp, blockIdx.x);
  const int mode = states[pair].mode;
  // batch window: a pair that was handed a slot (PENDING) starts with the NEXT round; only its own workgroup changes its mode here
  if (mode == MODE_PENDING && !trial_round && threadIdx.x == 0) states[pair].mode = MODE_LINEARIZE;
  if (mode == (trial_round ? MODE_TRIAL : MODE_LINEARIZE)) {
    __shared__ double s_grp[16 * kPartialStride];
    __shared__ double s_tot[kPartialStride];
    const PairDesc d = descs[pair];
    const uint32_t per = (uint32_t)(trial_round ? kp.points_per_block : kp.lin_points_per_block);
    const int nblocks = (int)((d.src.num_points + per - 1u) / per);
    const int j = threadIdx.x & 31, g8 = threadIdx.x >> 5;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    // canonical partition: group r (0..31) = rows r, r + 32, r + 64, ...; thread (g8, j) sums groups r = g8 + 8q, q = 0..3, each in
    // row order.  The four groups advance together, 32 independent loads in flight per trip (a 100k-point scan has 391 rows: 2
    // trips to memory instead of 16); a row past the end is skipped, not added as zero (-0.0 + 0.0 would lose the sign)
    if (j < kNumSums) {
      constexpr int kRowsPerTrip = 8;   // rows of each of the thread's four groups fetched together: 32 loads in flight
      for (int b0 = 0; b0 < nblocks; b0 += 32 * kRowsPerTrip) {
        double x[4][kRowsPerTrip];
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
          for (int k = 0; k < kRowsPerTrip; k++) {
            const int b = b0 + g8 + 8 * q + 32 * k;
            x[q][k] = b < nblocks ? gload_d(d.partials + (size_t)b * kPartialStride + j) : 0.0;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
          for (int k = 0; k < kRowsPerTrip; k++) {
            const int b = b0 + g8 + 8 * q + 32 * k;
            if (b < nblocks) v[q] += x[q][k];
          }
        }
      }
    }
    // group totals added in group order, 16 groups at a time through 4 KB of LDS
    double t = 0.0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      s_grp[(g8) * kPartialStride + j] = v[2 * half];            // groups 16 * half + g8
      s_grp[(g8 + 8) * kPartialStride + j] = v[2 * half + 1];    // groups 16 * half + 8 + g8
      __syncthreads();
      if (threadIdx.x < kNumSums) {
        for (int k = 0; k < 16; k++) t += s_grp[k * kPartialStride + threadIdx.x];
      }
      __syncthreads();
    }
    if (threadIdx.x < kNumSums) s_tot[threadIdx.x] = t;
    __syncthreads();
    if (kp.do_step) {
      // The solver step works on a copy of the pair's state in LDS: its ~200 loads and stores (several of them behind calls, i.e.
      // through memory and dependent on each other) then cost an LDS access each instead of a trip to L2 -- the copy in and the
      // copy out are one cooperative trip each.  Only this workgroup writes an ACTIVE pair's state (the hand-off below touches
      // pairs that are waiting), so the copy cannot go stale.
      __shared__ PairState s_state;
      static_assert(sizeof(PairState) % sizeof(double) == 0, "PairState is copied in doubles");
      constexpr int kStateDoubles = (int)(sizeof(PairState) / sizeof(double));
      double* ls = reinterpret_cast<double*>(&s_state);
      double* gs = reinterpret_cast<double*>(&states[pair]);
      for (int k = threadIdx.x; k < kStateDoubles; k += 256) ls[k] = gload_d(gs + k);
      __syncthreads();
      if (threadIdx.x == 0) {
        if (mode == MODE_LINEARIZE) {
          double H[36], b[6], cost;
          int inl;
          unpack_sums(s_tot, H, b, &cost, &inl);
          after_linearize(s_state, lp, H, b, cost, inl);
        } else {
          after_trial(s_state, lp, s_tot[27]);
        }
        // batch window: a pair that just finished hands its slot to the next queued pair.  The queued pair's own workgroup of THIS
        // launch may already have read its mode (WAIT), so it is only marked PENDING here and promoted by its own workgroup in a
        // later step launch -- activating it in place raced with that read (stale partial rows summed; round-1 advisor finding)
        if (s_state.mode == MODE_DONE && queue) {
          const unsigned int next = atomicAdd(queue, 1u);
          if (next < (unsigned int)npairs) states[next].mode = MODE_PENDING;
        }
        if (write_flags) __hip_atomic_store(flags_row + pair, (unsigned char)(s_state.mode != MODE_DONE ? 1 : 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
      for (int k = threadIdx.x; k < kStateDoubles; k += 256) gstore_d(gs + k, ls[k]);
      return;
    }
    if (threadIdx.x == 0) for (int k = 0; k < kNumSums; k++) sums_out[pair * kPartialStride + k] = s_tot[k];
  }
  // status of every pair this launch did not step, after the last step launch of the round: 1 = still active, 2 = done
  if (write_flags && threadIdx.x == 0)
    __hip_atomic_store(flags_row + pair, (unsigned char)(states[pair].mode != MODE_DONE ? 1 : 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_finish_round(hipStream_t stream, const PairDesc* d_descs, PairState* d_states, const KernelParams& kp, const LsqParams& lp, int npairs, bool trial_round,
                         bool write_flags, unsigned char* d_flags_row, double* d_sums, unsigned int* d_queue, int total_pairs) {
  k_finish_round<<<npairs, 256, 0, stream>>>(d_descs, d_states, kp, lp, trial_round ? 1 : 0, write_flags ? 1 : 0, d_flags_row, d_sums, d_queue, total_pairs > 0 ? total_pairs : npairs);
}

// ---------------------------------------------------------------------------
// k_linearize: grid = (tiles_per_pair, pairs launched), block = 256, one scan point per lane
// ---------------------------------------------------------------------------
constexpr int kCapCells = 2048;   // LDS voxel grid of a tile (one uint16 per cell)
constexpr int kCapPts = 1536;     // map points staged per tile (float4 each); keeps the workgroup under 32 KB of LDS (5 per CU)
constexpr int kCapJobs = 64;      // queued 3- and 4-neighbour plane fits per tile; beyond that the lane solves its own
constexpr int kCapBricks = 64;    // bricks under a tile box
constexpr uint16_t kNoCell = 0xffffu;

// TIMING (diagnostic build only): lane 0 of every tile stamps s_memtime at the phase
// boundaries and adds the differences to stats[8..15]; nothing is computed from them.
#define PCM_STAMP(slot)                                                        \
  if (TIMING) {                                                                \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();             \
    if (threadIdx.x == 0) atomicAdd(&stats[8 + (slot)], t_now - t_prev);       \
    t_prev = t_now;                                                            \
  }

template <bool STATS, bool TIMING, bool WRITE_PLANES, bool LIO, bool FUSED = false>
__global__ void __launch_bounds__(256, 5) k_linearize(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp,
                                                      unsigned long long* __restrict__ stats, LsqParams lp = LsqParams{}, unsigned char* __restrict__ flags_row = nullptr) {
  // XCD-aware placement, balanced: workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Inside every run of 64
  // consecutive tiles of a pair the ids are transposed (8 x 8), so that 8 neighbouring tiles -- neighbours on the scan's Morton curve,
  // readers of the same map bricks -- share an XCD while every XCD still takes an equal share of every pair.
  // (Handing every XCD one contiguous run of the whole grid instead lost 3.5 %: profiles/r02_xcd_aware_mapping_experiment.txt.)
  uint32_t tile_x = blockIdx.x;
  {
    const uint32_t base = blockIdx.x & ~63u, w = blockIdx.x & 63u;
    if (base + 64u <= gridDim.x) tile_x = base + (w & 7u) * 8u + (w >> 3);   // the last, partial run keeps its order
  }
  const uint32_t slot_y = blockIdx.y;
  const int pair = PCM_PAIR_OF(kp, slot_y);
  if (states[pair].mode != MODE_LINEARIZE) {
    if constexpr (FUSED) {   // a pair that finished since the host last looked: its status byte of this round still has to land
      if (tile_x == 0 && threadIdx.x == 0)
        __hip_atomic_store(flags_row + pair, (unsigned char)(states[pair].mode != MODE_DONE ? 1 : 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const PairDesc d = descs[pair];
  const uint32_t i = tile_x * 256u + threadIdx.x;
  if (tile_x * 256u >= d.src.num_points) return;
  const bool live = i < d.src.num_points;
  const PoseF P = load_pose(states[pair].x0);
  const TargetView tg = d.tgt;
  const bool do_search = !LIO || kp.lio_rematch != 0;   // LIO with converge == false re-uses the stored planes (laser_mapping.cc:616)

  __shared__ int s_red[4][6];
  __shared__ int s_box[8];                 // origin xyz, dims xyz, ncell, dense flag
  __shared__ int s_bbox[8];                // first brick xyz, brick dims xyz, nbricks
  __shared__ int4 s_borg[kCapBricks];      // voxel coordinates of each brick's corner relative to the box origin
  __shared__ uint32_t s_bps[kCapBricks];   // first map point of each brick under the box
  __shared__ uint32_t s_boff[kCapBricks + 1];  // its offset in s_pts (exclusive scan of the point counts)
  __shared__ uint32_t s_njobs;
  __shared__ uint16_t s_cell[kCapCells];   // first staged point of the voxel in that cell, kNoCell when empty
  __shared__ float4 s_pts[kCapPts];        // the bricks' map points, .w = voxel tag
  __shared__ uint32_t s_job[kCapJobs];     // owner tid | m << 16
  __shared__ uint32_t s_jobid[kCapJobs][4];   // the 3 or 4 neighbour ids of the job

  uint32_t n_cand = 0, n_probe = 0;
  unsigned long long t_prev = 0;
  if (TIMING) t_prev = __builtin_amdgcn_s_memtime();
  float4 p = make_float4(0.f, 0.f, 0.f, 1.f);
  float pn_body = 0.f;
  float q[3] = {0.f, 0.f, 0.f};
  int cx = 0, cy = 0, cz = 0;
  bool search = false;  // lanes whose query voxel lies inside the key range of the table
  if (live) {
    p = gload4(d.src.pts + i);
    pn_body = pcm_sqrtf_rn(p.x * p.x + p.y * p.y + p.z * p.z);   // p_body.norm() of the 81 pd2^2 test (:631), kept instead of the point
    if (LIO) {   // p_w = R_wl * p_body + t_wl with R_wl a float quaternion (Eigen _transformVector)  laser_mapping.cc:602-612
      const float qx = d.lio.q_wl[0], qy = d.lio.q_wl[1], qz = d.lio.q_wl[2], qw = d.lio.q_wl[3];
      float uv[3] = {qy * p.z - qz * p.y, qz * p.x - qx * p.z, qx * p.y - qy * p.x};
      uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
      const float c3[3] = {qy * uv[2] - qz * uv[1], qz * uv[0] - qx * uv[2], qx * uv[1] - qy * uv[0]};
      q[0] = (p.x + qw * uv[0] + c3[0]) + d.lio.t_wl[0];
      q[1] = (p.y + qw * uv[1] + c3[1]) + d.lio.t_wl[1];
      q[2] = (p.z + qw * uv[2] + c3[2]) + d.lio.t_wl[2];
    } else {
      transform(P, p, q);
    }
    const float fx = roundf(q[0] * tg.inv_res), fy = roundf(q[1] * tg.inv_res), fz = roundf(q[2] * tg.inv_res);  // Pos2Grid  ivox3d.h:283-286
    const float lim = (float)(kCoordBias - 32);
    search = fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim;  // also false for NaN
    if (search) { cx = (int)fx; cy = (int)fy; cz = (int)fz; }
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 pl = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
  bool use_lds = false, box_ok = false;
  bool ref_fit = false, ref_ok = false;   // LIO reference semantics: esti_plane ran for this point (pl = what it wrote) / what it returned
  if (do_search) {
    // ---- voxel bounding box of the tile ---------------------------------------------------------
    {
      const int big = 0x3fffffff;
      int mn[3] = {search ? cx : big, search ? cy : big, search ? cz : big};
      int mx[3] = {search ? cx : -big, search ? cy : -big, search ? cz : -big};
      // wave reductions on the DPP path (linearize_common.h): 36 ds_bpermute round trips less per tile
  #pragma unroll
      for (int a = 0; a < 3; a++) { mn[a] = wave_min_i32(mn[a]); mx[a] = wave_max_i32(mx[a]); }
      if (lane == 0) {
  #pragma unroll
        for (int a = 0; a < 3; a++) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
      }
    }
    if (threadIdx.x == 0) s_njobs = 0;
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
        if (dim > kCapCells) ok = false;
        ncell = ok ? ncell * (int)dim : ncell;
        if (ncell > kCapCells) ok = false;
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
        if (nb > kCapBricks) ok = false;
      }
      s_box[6] = ncell;
      s_box[7] = ok ? 1 : 0;
    }
    __syncthreads();
    use_lds = s_box[7] != 0;   // uniform over the workgroup
    if (STATS) box_ok = use_lds;   // counters pass: the tile's voxel box and bricks fitted (the staged points may still not)
    PCM_STAMP(0)   // load + transform + tile box

    Best best;
    best_init(best, kp.max_range_sq);

    if (use_lds) {
      const int ox0 = s_box[0], oy0 = s_box[1], oz0 = s_box[2];
      const int Dx = s_box[3], Dy = s_box[4], Dz = s_box[5];
      const int bx0 = s_bbox[0], by0 = s_bbox[1], bz0 = s_bbox[2], nby = s_bbox[4], nbz = s_bbox[5], nb = s_bbox[6];
      // ---- one probe per BRICK under the box (wave 0), exclusive scan of their point counts -------
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
          // voxel coordinates of the brick's corner relative to the tile box
          s_borg[lane] = make_int4(((bx0 + x) << kBrickShift) - ox0, ((by0 + y) << kBrickShift) - oy0, ((bz0 + z) << kBrickShift) - oz0, 0);
        }
        const uint32_t incl = scan_add_wave(npts);
        if (lane < nb) { s_bps[lane] = ps; s_boff[lane] = incl - npts; }
        if (lane == 63) s_boff[kCapBricks] = incl;   // total
      }
      // meanwhile everybody clears the cell grid
  #pragma unroll
      for (int j = 0; j < kCapCells / 256; j++) s_cell[threadIdx.x + 256 * j] = kNoCell;
      __syncthreads();
      const uint32_t total = s_boff[kCapBricks];
      use_lds = total < (uint32_t)kCapPts;   // still uniform; one slot is kept for the end-of-run sentinel
      PCM_STAMP(1)   // brick probes
      if (use_lds) {
        // ---- stage the bricks' map points through LDS: flat, coalesced, all loads in flight --------
        float4 v[kCapPts / 256];
        int vb[kCapPts / 256];
        int b = 0;   // brick of staged point k: k grows with r, so the brick index only moves forward
  #pragma unroll
        for (int r = 0; r < kCapPts / 256; r++) {
          const uint32_t k = threadIdx.x + 256u * r;
          vb[r] = -1;
          if (k < total) {
            while (b + 1 < nb && s_boff[b + 1] <= k) b++;   // nb is small (typically 1..8)
            vb[r] = b;
            v[r] = gload4(tg.pts + s_bps[b] + (k - s_boff[b]));
          }
        }
        // every voxel head among the staged points (bit 31 of its tag) registers itself in the cell grid while the
        // points go to LDS: one barrier for both
  #pragma unroll
        for (int r = 0; r < kCapPts / 256; r++) {
          const uint32_t k = threadIdx.x + 256u * r;
          if (k < total) {
            s_pts[k] = v[r];
            const int tag = __float_as_int(v[r].w);
            if (tag < 0) {
              const int4 o = s_borg[vb[r]];
              const int li = tag & 511;
              const int x = o.x + (li >> 6), y = o.y + ((li >> 3) & 7), z = o.z + (li & 7);
              if (x >= 0 && x < Dx && y >= 0 && y < Dy && z >= 0 && z < Dz) s_cell[(x * Dy + y) * Dz + z] = (uint16_t)k;
            }
          }
        }
        if (threadIdx.x == 0) s_pts[total] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));   // end-of-run sentinel (a "head")
        __syncthreads();
        PCM_STAMP(2)   // stage map points
        PCM_STAMP(3)   // cell grid
        // ---- per-lane 27-cell / 5-NN search out of LDS (reference cell order) ----------------------
        if (search) {
          // all 27 cell heads first (independent LDS reads, one wait), then the occupied cells in
          // reference order; inside a voxel the next staged point is fetched while the current one
          // is offered, so a candidate costs one overlapped LDS read instead of two serial ones
          const int DyDz = Dy * Dz;
          const int cell0 = ((cx - ox0) * Dy + (cy - oy0)) * Dz + (cz - oz0);
          uint16_t kh[27];
          if (kp.num_neighbors == 27) {   // the common case without 27 scalar compare-and-branch pairs
  #pragma unroll
            for (int g = 0; g < 27; g++) kh[g] = s_cell[cell0 + kNearby[g][0] * DyDz + kNearby[g][1] * Dz + kNearby[g][2]];
          } else {
  #pragma unroll
            for (int g = 0; g < 27; g++) {
              kh[g] = kNoCell;
              if (g < kp.num_neighbors) kh[g] = s_cell[cell0 + kNearby[g][0] * DyDz + kNearby[g][1] * Dz + kNearby[g][2]];
            }
          }
          // a voxel's points are one run of s_pts that ends where the next voxel head (tag bit 31; the sentinel
          // s_pts[total] is one) begins.  The list carries byte offsets (k * 16) until the search is over.
          const char* pbase = reinterpret_cast<const char*>(s_pts);
  #pragma unroll
          for (int g = 0; g < 27; g++) {
            if (kh[g] != kNoCell) {
              uint32_t off = (uint32_t)kh[g] << 4;
              float4 mp = *reinterpret_cast<const float4*>(pbase + off);
              for (;;) {
                const float4 nx = *reinterpret_cast<const float4*>(pbase + off + 16);
                if (STATS) n_cand++;
                best_offer(best, mp, q, off, kp.max_range_sq);
                // advance first, then leave: written the other way round the compiler keeps `mp` and `off` of the lanes that
                // leave with four selects and a second compare per trip, although nobody reads them after the loop
                mp = nx;
                off += 16;
                if (__float_as_int(nx.w) < 0) break;   // the staged point just fetched opens another voxel (or is the sentinel)
              }
            }
          }
  #pragma unroll
          for (int j = 0; j < K; j++) best.i[j] >>= 4;
        }
      }
    }
    if (!use_lds && search) knn_global<STATS>(tg, q, cx, cy, cz, kp.num_neighbors, kp.max_range_sq, best, n_cand, n_probe);
    best_finish(best);
    if constexpr (LIO) {   // nearest_points_[i] for MapIncremental: indices into the map's point array, nearest first
      if (live) {
#pragma unroll
        for (int j = 0; j < K; j++) {
          uint32_t gi = ~0u;
          if (j < best.m) {
            gi = best.i[j];
            if (use_lds) {
              int b = 0;
              while (b + 1 < s_bbox[6] && s_boff[b + 1] <= gi) b++;
              gi = s_bps[b] + (gi - s_boff[b]);
            }
          }
          *(PCM_GLOBAL uint32_t*)(d.nn + (size_t)i * K + j) = gi;
        }
      }
    }
    PCM_STAMP(4)   // 27-cell / 5-NN search

    // ---- plane fit on the <= 5 neighbours  (laser_mapping.cc:619-623) -----------------------------
    // 5 neighbours (the float path, almost every lane) is solved in place; the rare
    // 3- and 4-neighbour cases (double path) are queued and solved by the first lanes
    // of the workgroup afterwards, so one straggler does not drag its whole wave
    // through the double-precision QR.
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
        const bool ok = esti_plane(px, py, pz, K, kp.plane_threshold, &fit);
        if (LIO && kp.lio_ref) { pl = fit; ref_fit = true; ref_ok = ok; }   // plane_coef_[i] is written whatever the verdict (common_lib.h:229-233)
        else if (ok) pl = fit;
      } else if (best.m >= KMIN) {
        my_job = atomicAdd(&s_njobs, 1u);
      }
    }
    PCM_STAMP(5)   // float plane fit
    // the queue holds kCapJobs fits; a tile with more of them (a scan over nearly empty map) goes round again
    for (uint32_t base = 0;; base += kCapJobs) {
      if (my_job != ~0u && my_job >= base && my_job < base + kCapJobs) {
        const uint32_t slot = my_job - base;
        s_job[slot] = threadIdx.x | ((uint32_t)best.m << 16);
  #pragma unroll
        for (int j = 0; j < 4; j++) s_jobid[slot][j] = best.i[j];
      }
      __syncthreads();   // also: the float fits are through with s_pts
      const uint32_t njobs = s_njobs;   // nobody adds to it any more
      if (base >= njobs) break;
      const uint32_t here = min(njobs - base, (uint32_t)kCapJobs);
      if (threadIdx.x < here) {
        const uint32_t job = threadIdx.x;
        const uint32_t m = s_job[job] >> 16;
        float px[K], py[K], pz[K];
  #pragma unroll
        for (int j = 0; j < K; j++) {
          float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < (int)m) mp = use_lds ? s_pts[s_jobid[job][j]] : gload4(tg.pts + s_jobid[job][j]);
          px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
        }
        float4 fit;
        const bool ok = esti_plane(px, py, pz, (int)m, kp.plane_threshold, &fit);
        if (LIO && kp.lio_ref) s_job[job] = ok ? 1u : 0u;
        else if (!ok) fit.x = __builtin_nanf("");
        s_jobid[job][0] = __float_as_uint(fit.x); s_jobid[job][1] = __float_as_uint(fit.y);   // hand the plane back to its owner lane
        s_jobid[job][2] = __float_as_uint(fit.z); s_jobid[job][3] = __float_as_uint(fit.w);
      }
      __syncthreads();   // also: nobody reads s_pts after this point (its memory is re-used below)
      if (my_job != ~0u && my_job >= base && my_job < base + kCapJobs) {
        const uint32_t slot = my_job - base;
        pl = make_float4(__uint_as_float(s_jobid[slot][0]), __uint_as_float(s_jobid[slot][1]), __uint_as_float(s_jobid[slot][2]), __uint_as_float(s_jobid[slot][3]));
        if (LIO && kp.lio_ref) { ref_fit = true; ref_ok = s_job[slot] != 0u; }
      }
      if (base + kCapJobs >= njobs) break;
      __syncthreads();   // the slots are free for the next round of the queue
    }
  } else {
    if (live) pl = gload4(d.planes + ((LIO && kp.lio_ref == 2) ? __float_as_uint(p.w) : i));   // plane of the previous ObsModel call (clean semantics: NaN = none)
    __syncthreads();
  }

  if constexpr (LIO) {
    // ---- jueying_lio measurement row (laser_mapping.cc:674-698) -> 16-float row in LDS; the IEKF's
    //      HTH = h_x^T h_x and h_x^T h (esekfom.hpp:1687,1706) are reduced here: 92 sums per tile
    float* s_row = reinterpret_cast<float*>(s_pts);                     // [256][16]: 12 columns, h, selected
    double* s_grp = reinterpret_cast<double*>(s_row + 256 * 16);        // [8][96] group partials
    {
      float row[16];
#pragma unroll
      for (int a = 0; a < 16; a++) row[a] = 0.f;
      if (live) {
        bool sel;
        float res = 0.f;
        if (kp.lio_ref) {
          // the members of LaserMapping as they are: the flag and the plane of a point change only when a matching call
          // reaches them, and a selected point that fails the 81 pd2^2 test keeps its flag and the residual stored for its
          // index by an earlier call -- of this frame or of an older one (laser_mapping.cc:335-339, 616-636, 646-650)
          const uint32_t oi = kp.lio_ref == 2 ? __float_as_uint(p.w) : i;
          PCM_GLOBAL float* aux = (PCM_GLOBAL float*)d.lio_aux + 2 * (size_t)oi;
          res = aux[0];
          sel = aux[1] != 0.f;
          if (do_search) {
            sel = ref_fit && ref_ok;
            if (ref_fit) gstore4(d.planes + oi, pl);
          }
          if (sel) {
            const float pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;
            const float pn = pn_body;
            if (pn > 81.f * pd2 * pd2) res = pd2;
          }
          aux[0] = res;
          aux[1] = sel ? 1.f : 0.f;
        } else {
          sel = !(pl.x != pl.x);
          if (do_search) gstore4(d.planes + i, pl);                        // plane_coef_[i]; the residual test is re-evaluated every call
          if (sel) {
            res = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;
            const float pn = pn_body;
            sel = pn > 81.f * res * res;
          }
        }
        {
          const float pd2 = res;
          if (sel) {
            const float* oR = d.lio.off_R;
            const float* Rt = d.lio.Rt;
            float pt[3], C[3];
#pragma unroll
            for (int a = 0; a < 3; a++) pt[a] = (oR[a * 3 + 0] * p.x + oR[a * 3 + 1] * p.y) + oR[a * 3 + 2] * p.z + d.lio.off_t[a];   // point_this
#pragma unroll
            for (int a = 0; a < 3; a++) C[a] = (Rt[a * 3 + 0] * pl.x + Rt[a * 3 + 1] * pl.y) + Rt[a * 3 + 2] * pl.z;               // C = R^T n
            row[0] = pl.x; row[1] = pl.y; row[2] = pl.z;
            row[3] = (0.f * C[0] + -pt[2] * C[1]) + pt[1] * C[2];                                                                 // A = skew(point_this) C
            row[4] = (pt[2] * C[0] + 0.f * C[1]) + -pt[0] * C[2];
            row[5] = (-pt[1] * C[0] + pt[0] * C[1]) + 0.f * C[2];
            if (kp.lio_extrinsic) {   // B = (skew(p_body) off_R^T) C ; then C
              const float S[9] = {0.f, -p.z, p.y, p.z, 0.f, -p.x, -p.y, p.x, 0.f};
#pragma unroll
              for (int a = 0; a < 3; a++) {
                float sm[3];
#pragma unroll
                for (int b = 0; b < 3; b++) sm[b] = (S[a * 3 + 0] * oR[b * 3 + 0] + S[a * 3 + 1] * oR[b * 3 + 1]) + S[a * 3 + 2] * oR[b * 3 + 2];
                row[6 + a] = (sm[0] * C[0] + sm[1] * C[1]) + sm[2] * C[2];
                row[9 + a] = C[a];
              }
            }
            row[12] = -pd2;    // ekfom_data.h(i) = -residual
            row[13] = 1.f;
          }
        }
      }
      float4* dst = reinterpret_cast<float4*>(s_row + threadIdx.x * 16);
#pragma unroll
      for (int a = 0; a < 4; a++) dst[a] = make_float4(row[4 * a], row[4 * a + 1], row[4 * a + 2], row[4 * a + 3]);
    }
    __syncthreads();
    {
      const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
      const float* r0 = s_row + (g * 32) * 16;
#pragma unroll
      for (int tt = 0; tt < 3; tt++) {
        const int term = j + 32 * tt;
        if (term < kLioSums) {
          const int ia = c_lio_a[term], ib = c_lio_b[term];
          double v = 0.0;
#pragma unroll 8
          for (int k = 0; k < 32; k++) v = fma((double)r0[k * 16 + ia], (double)r0[k * 16 + ib], v);
          s_grp[g * kLioStride + term] = v;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < kLioSums) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < 8; g++) v += s_grp[g * kLioStride + threadIdx.x];
      gstore_d(d.partials + (size_t)tile_x * kLioStride + threadIdx.x, v);
    }
  } else if constexpr (!FUSED) {
    // residual / Jacobian rows as doubles in LDS, 29 sums of the tile (linearize_common.h): the same products and the same order of
    // additions as the float rows this kernel wrote before, with 8 conversions per lane instead of 64 in the reduction loop
    static_assert(sizeof(float4) * kCapPts >= (size_t)kReduceLdsBytes, "the reduction rows alias the staged points");
    residual_and_reduce<WRITE_PLANES>(d, i, tile_x, live, pl, q, pn_body, s_pts);
  } else {
    // ---- residual / Jacobian of this lane's point -> one 8-float row in LDS ---------------------------
    // (the LDS of s_pts is free now: nobody reads map points after the barrier above)
    float* s_row = reinterpret_cast<float*>(s_pts);                       // [256][8]: J0..J5, e, selected
    double* s_grp = reinterpret_cast<double*>(s_row + 256 * 8);           // [8][32] group partials
    {
      float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (live) {
        bool sel = !(pl.x != pl.x);
        if (sel) {
          const float pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;  // laser_mapping.cc:627-629
          const float pn = pn_body;
          sel = pn > 81.f * pd2 * pd2;                                       // :631
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
      float4* dst = reinterpret_cast<float4*>(s_row + threadIdx.x * 8);
      dst[0] = make_float4(row[0], row[1], row[2], row[3]);
      dst[1] = make_float4(row[4], row[5], row[6], row[7]);
    }
    __syncthreads();
    // ---- 29 sums over the tile's 256 rows: thread (group g, term j) adds 32 rows in double ------------
    // term j = row[ia] * row[ib]: 21 x H upper triangle, 6 x b = J e, cost = e e, count = sel sel
    {
      const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
      double v = 0.0;
      if (j < kNumSums) {
        const int ia = c_term_a[j], ib = c_term_b[j];
        const float* r0 = s_row + (g * 32) * 8;
  #pragma unroll 8
        for (int k = 0; k < 32; k++) v = fma((double)r0[k * 8 + ia], (double)r0[k * 8 + ib], v);
      }
      s_grp[g * kPartialStride + j] = v;
    }
    __syncthreads();
    if constexpr (!FUSED) {
      if (threadIdx.x < kNumSums) {
        double v = 0.0;
  #pragma unroll
        for (int g = 0; g < 8; g++) v += s_grp[g * kPartialStride + threadIdx.x];
        gstore_d(d.partials + (size_t)tile_x * kPartialStride + threadIdx.x, v);
      }
    } else {
      // In-launch hand-off of the partial rows to the workgroup that completes the pair's round (no second launch, no
      // agent-scope release fence -- that fence writes back the whole L2 and cost more than the launch it saved):
      //   producer  the 29 sums are stored write-through (sc1) by lanes of wave 0; the wave drains its stores
      //             (s_waitcnt vmcnt(0)) and only then lane 0 takes the arrival ticket (agent-scope relaxed add);
      //   consumer  the workgroup whose ticket is the last one: lane 0 runs ONE agent-scope acquire (buffer_inv sc1:
      //             this CU's L1 / non-local L2 lines of earlier rounds) behind its returned add, the workgroup barrier
      //             releases the other waves, and every row is read with sc1 loads.
      // The ticket counter is reset by the last arriver; the next round is a later launch on the same stream.
      __shared__ unsigned int s_last;
      if (threadIdx.x < 64) {   // wave 0
        if (threadIdx.x < kNumSums) {
          double v = 0.0;
  #pragma unroll
          for (int g = 0; g < 8; g++) v += s_grp[g * kPartialStride + threadIdx.x];
          gstore_d_wt(d.partials + (size_t)tile_x * kPartialStride + threadIdx.x, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
          const int nblocks = (int)((d.src.num_points + 255u) / 256u);
          const unsigned int t = __hip_atomic_fetch_add((PCM_GLOBAL unsigned int*)d.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned int last = (t == (unsigned int)(nblocks - 1)) ? 1u : 0u;
          if (last) {
            __hip_atomic_store((PCM_GLOBAL unsigned int*)d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next round's launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          s_last = last;
        }
      }
      __syncthreads();
      if (s_last) {
        const int nblocks = (int)((d.src.num_points + 255u) / 256u);
        double* f_grp = reinterpret_cast<double*>(s_pts);          // 32 x 32 doubles; s_pts is free by now
        double* f_tot = f_grp + 32 * kPartialStride;
        finish_pair_in_place(d, const_cast<PairState*>(states), pair, lp, nblocks, flags_row, f_grp, f_tot);
      }
    }
  
  }
  PCM_STAMP(6)   // queued double-precision fits + residual + workgroup reduction
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

// GN rounds with very few live pairs: the search kernel's last workgroup per pair also takes the step (no k_finish_round launch)
void launch_linearize_fused(hipStream_t stream, const PairDesc* d_descs, PairState* d_states, const KernelParams& kp, const LsqParams& lp, int npairs, unsigned char* d_flags_row) {
  dim3 grid((unsigned)kp.tiles_per_pair, (unsigned)npairs);
  k_linearize<false, false, false, false, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, nullptr, lp, d_flags_row);
}

void launch_linearize(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes,
                      unsigned long long* d_stats, bool timing) {
  dim3 grid((unsigned)kp.tiles_per_pair, (unsigned)npairs);
#define PCM_LAUNCH(S, T, W) k_linearize<S, T, W, false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, d_stats)
  if (d_stats && timing) { if (write_planes) PCM_LAUNCH(false, true, true); else PCM_LAUNCH(false, true, false); }
  else if (d_stats) { if (write_planes) PCM_LAUNCH(true, false, true); else PCM_LAUNCH(true, false, false); }
  else { if (write_planes) PCM_LAUNCH(false, false, true); else PCM_LAUNCH(false, false, false); }
#undef PCM_LAUNCH
}

// jueying_lio measurement model: same search kernel, 12-column rows, IEKF reduction
void launch_lio_obs(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp) {
  dim3 grid((unsigned)kp.tiles_per_pair, 1u);
  k_linearize<false, false, true, true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp, nullptr);
}

// fixed-order sum of the LIO partial rows -> 96 doubles (78 HTH, 12 HTh, sum h^2, count)
__global__ void __launch_bounds__(1024) k_lio_finish(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  __shared__ double s_grp[10 * kLioStride];
  const int t = threadIdx.x % kLioStride, r = threadIdx.x / kLioStride;   // 10 row groups x 96 columns (960 threads)
  if (r < 10) {
    double v = 0.0;
    if (t < kLioSums) for (int b = r; b < nblocks; b += 10) v += gload_d(partials + (size_t)b * kLioStride + t);
    s_grp[r * kLioStride + t] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLioSums) {
    double v = 0.0;
    for (int k = 0; k < 10; k++) v += s_grp[k * kLioStride + threadIdx.x];
    out[threadIdx.x] = v;
  }
}

void launch_lio_finish(hipStream_t stream, const double* d_partials, int nblocks, double* d_out) {
  k_lio_finish<<<1, 1024, 0, stream>>>(d_partials, nblocks, d_out);
}

// residuals_.resize(n, 0); point_selected_surf_.resize(n, true): the appended entries  (laser_mapping.cc:337-338)
__global__ void k_lio_members_init(float2* __restrict__ aux, uint32_t first, uint32_t last) {
  const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < last) aux[i] = make_float2(0.f, 1.f);
}
void launch_lio_members_init(hipStream_t stream, float2* aux, uint32_t first, uint32_t last) {
  if (last > first) k_lio_members_init<<<(last - first + 255u) / 256u, 256, 0, stream>>>(aux, first, last);
}

// ---------------------------------------------------------------------------
// k_trial: cost of the correspondences (planes) of the last linearize at the LM
// trial pose -- the compute_error contract (fast_gicp_impl.hpp:213-237).
// grid = (blocks_per_pair, npairs), block = 256
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_trial(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp) {
  const int pair = PCM_PAIR_OF(kp, blockIdx.y);
  if (states[pair].mode != MODE_TRIAL) return;
  const PairDesc d = descs[pair];
  const uint32_t begin = blockIdx.x * (uint32_t)kp.points_per_block;
  if (begin >= d.src.num_points) return;
  const PoseF P = load_pose(states[pair].xi);
  uint32_t end = begin + (uint32_t)kp.points_per_block;
  end = end < d.src.num_points ? end : d.src.num_points;
  double cost = 0.0, cnt = 0.0;
  for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
    const float4 pl = gload4(d.planes + i);
    if (pl.x != pl.x) continue;  // not selected by the last linearize
    const float4 p = gload4(d.src.pts + i);
    float q[3];
    transform(P, p, q);
    const float pd2 = pl.x * q[0] + pl.y * q[1] + pl.z * q[2] + pl.w;
    const double e = (double)pd2;
    cost = fma(e, e, cost);
    cnt += 1.0;
  }
  __shared__ double s_red64[4 * kPartialStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  cost = wave_sum(cost);
  cnt = wave_sum(cnt);
  if (threadIdx.x < 4 * kPartialStride) s_red64[threadIdx.x] = 0.0;
  __syncthreads();
  if (lane == 0) { s_red64[wave * kPartialStride + 27] = cost; s_red64[wave * kPartialStride + 28] = cnt; }
  __syncthreads();
  if (threadIdx.x < kNumSums) {
    const double v = ((s_red64[threadIdx.x] + s_red64[kPartialStride + threadIdx.x]) + s_red64[2 * kPartialStride + threadIdx.x]) + s_red64[3 * kPartialStride + threadIdx.x];
    gstore_d(d.partials + (size_t)blockIdx.x * kPartialStride + threadIdx.x, v);
  }
}

void launch_trial(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs) {
  dim3 grid((unsigned)kp.blocks_per_pair, (unsigned)npairs);
  k_trial<<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
}

__global__ void k_init_states(PairState* __restrict__ states, const float* __restrict__ guesses, int npairs, int max_iterations, int window, unsigned int* __restrict__ queue) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair == 0 && queue) *queue = (unsigned int)window;   // next pair to activate
  if (pair >= npairs) return;
  PairState s;
  init_state(s, guesses + pair * 16);
  if (max_iterations <= 0) s.mode = MODE_DONE;
  else if (pair >= window) s.mode = MODE_WAIT;
  states[pair] = s;
}

void launch_init_states(hipStream_t stream, PairState* d_states, const float* d_guesses, int npairs, int max_iterations, int window, unsigned int* d_queue) {
  k_init_states<<<(npairs + 63) / 64, 64, 0, stream>>>(d_states, d_guesses, npairs, max_iterations, window, d_queue);
}

__global__ void k_pack_results(const PairState* __restrict__ states, pcm_result* __restrict__ out, int npairs) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= npairs) return;
  const PairState& s = states[pair];
  pcm_result r;
  for (int i = 0; i < 16; i++) { r.T[i] = (float)s.x0[i]; r.T64[i] = s.x0[i]; }  // final_transformation_ = x0.cast<float>()
  for (int i = 0; i < 36; i++) r.H[i] = s.final_hessian[i];
  r.cost = s.last_cost;
  r.iterations = s.iter;
  r.converged = s.converged;
  r.num_linearize = s.num_linearize;
  r.num_compute_error = s.num_compute_error;
  r.num_inliers = s.num_inliers;
  r.status = s.mode == MODE_DONE ? s.status : PCM_ERR_INTERNAL;   // never launched / still iterating when the host loop ended: not a result
  out[pair] = r;
}

void launch_pack_results(hipStream_t stream, const PairState* d_states, pcm_result* d_results, int npairs) {
  k_pack_results<<<(npairs + 63) / 64, 64, 0, stream>>>(d_states, d_results, npairs);
}

}  // namespace pcm

// neighbour_lists.hip -- per-voxel candidate lists of a static submap and k_linearize_lists, the point-to-plane linearize pass that
// reads them (PCM_FLAG_NEIGHBOUR_LISTS).
//
// IVox::GetClosestPoint (/root/reference/src/jueying_lio/include/ivox3d/ivox3d.h:132-204) visits, for a query in voxel u, the
// voxels u + nearby_grids_[g] in list order and every point of each in insertion order.  That candidate sequence depends on u only.
// The tile kernel (kernels.hip) re-derives it per pass -- tile box, brick probes by one wave, the bricks' points staged through LDS,
// a cell grid, 27 per-cell walks in lock-step, five barriers in front of the plane fit -- and its waves spend four fifths of their
// life waiting (the search over a ready-made flat list is 12x cheaper than the cell walk).  For a submap that is
// registered against many times (the reference's own protocol: fast_gicp/src/align.cpp:51-104 reuses the target) the sequences are
// built ONCE, with the map: for every voxel of the occupied set dilated by the neighbourhood, the points of its <= 27 neighbour
// voxels in the reference's visit order, contiguous in HBM (27 x 16 B per map point).  The pass is then: voxel of the query ->
// one probe of the list index -> a flat walk of one contiguous run (lanes of a wave mostly share it) -> plane fit -> the shared
// residual / reduction tail.  No LDS staging, no cell grid, no barrier before the reduction; same candidates in the same order
// into the same best_offer: bit-identical planes (tests/test_gpu_neighbour_lists.py).
//
// The list index is an ordinary brick hash (voxel_hash.hip) built from one stand-in point per dilated voxel (its centre), so a
// TargetView serves as the view of the lists: bricks / bmask / bpref give the voxel's rank r, vox_start[r] .. vox_start[r + 1] is
// its run in pts -- which here holds the candidates (w = index of the point in the map's own array).
//
// Compiled with -ffp-contract=off (see kernels.hip).
#include "pcm_device.h"
#include "pcm_host.h"
#include "plane_fit.h"
#include "linearize_common.h"

#include <cstring>
#include <string>
#include <rocprim/rocprim.hpp>

namespace pcm {

namespace {

constexpr int kKeyBias = 1 << 20;

__device__ inline uint64_t nl_pack(int x, int y, int z) { return ((uint64_t)(uint32_t)(x + kKeyBias) << 42) | ((uint64_t)(uint32_t)(y + kKeyBias) << 21) | (uint64_t)(uint32_t)(z + kKeyBias); }

// one key per (occupied voxel, neighbour offset): the voxels a query can sit in and see this voxel.  grid = ceil(nvox / 256)
__global__ void k_nl_keys(const float4* __restrict__ pts, const uint32_t* __restrict__ vox_start, uint32_t nvox, float res, float inv_res, int mode, int nn, uint64_t* __restrict__ keys) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const float4 p = pts[vox_start[v]];
  const int cx = voxel_coord(p.x, res, inv_res, mode), cy = voxel_coord(p.y, res, inv_res, mode), cz = voxel_coord(p.z, res, inv_res, mode);   // the map's own convention (Pos2Grid  ivox3d.h:283-286; pclomp: floor(p * inverse_leaf_size))
  const int lim = kKeyBias - 64;
  for (int g = 0; g < nn; g++) {
    const int x = cx - c_nearby[g][0], y = cy - c_nearby[g][1], z = cz - c_nearby[g][2];   // u + nearby[g] = this voxel
    const bool ok = x > -lim && x < lim && y > -lim && y < lim && z > -lim && z < lim;
    keys[(size_t)v * nn + g] = ok ? nl_pack(x, y, z) : ~0ull;
  }
}

__global__ void k_nl_flags(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (keys[i] != ~0ull && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}

// the stand-in point of every distinct key: the centre of its voxel (Pos2Grid of it is the voxel again)
__global__ void k_nl_centres(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n, float res, float shift, float4* __restrict__ out,
                             uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1) *count = pos[i] + flag[i];
  if (!flag[i]) return;
  const uint64_t k = keys[i];
  const int x = (int)((k >> 42) & 0x1fffffu) - kKeyBias, y = (int)((k >> 21) & 0x1fffffu) - kKeyBias, z = (int)(k & 0x1fffffu) - kKeyBias;
  out[pos[i]] = make_float4(((float)x + shift) * res, ((float)y + shift) * res, ((float)z + shift) * res, 0.f);   // shift: 0 (round) or 0.5 (floor): the voxel's centre
}

// the run of voxel (vx, vy, vz) in the map's point array (brick probe re-used while consecutive cells stay in one brick)
struct BrickCursor { int bx = 0x7fffffff, by = 0, bz = 0; uint32_t slot = ~0u, base = 0; };
__device__ inline bool nl_voxel_run(const TargetView& tg, BrickCursor& c, int vx, int vy, int vz, uint32_t& s, uint32_t& e) {
  const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
  if (bx != c.bx || by != c.by || bz != c.bz) {
    uint32_t np = 0;
    c.slot = brick_find<false>(tg, bx, by, bz, c.base, np);
    c.bx = bx; c.by = by; c.bz = bz;
  }
  if (c.slot == ~0u) return false;
  const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
  const uint32_t m = gload_u(&tg.bmask[(size_t)c.slot * 16 + w]);
  if (!((m >> bit) & 1u)) return false;
  const uint32_t v = c.base + gload_u16(&tg.bpref[(size_t)c.slot * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u));
  s = gload_u(&tg.vox_start[v]);
  e = gload_u(&tg.vox_start[v + 1]);
  return true;
}

// FILL = false: len[r] = candidates of list voxel r; FILL = true: copy them.  One lane per list voxel.
template <bool FILL>
__global__ void k_nl_lists(const float4* __restrict__ centres, uint32_t nd, TargetView tg, int nn, uint32_t* __restrict__ len, const uint32_t* __restrict__ start, float4* __restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nd) return;
  const float4 c = centres[r];
  const int cx = (int)roundf(c.x * tg.inv_res), cy = (int)roundf(c.y * tg.inv_res), cz = (int)roundf(c.z * tg.inv_res);
  BrickCursor cur;
  uint32_t n = 0;
  const uint32_t o = FILL ? start[r] : 0u;
  for (int g = 0; g < nn; g++) {   // nearby_grids_ order  ivox3d.h:211-235
    uint32_t s, e;
    if (!nl_voxel_run(tg, cur, cx + c_nearby[g][0], cy + c_nearby[g][1], cz + c_nearby[g][2], s, e)) continue;
    if (FILL) {
      for (uint32_t k = s; k < e; k++) {   // the voxel's points in insertion order (points_)  ivox3d_node.hpp:158-166
        float4 p = gload4(tg.pts + k);
        p.w = __uint_as_float(k);
        out[o + n + (k - s)] = p;
      }
    }
    n += e - s;
  }
  if (!FILL) len[r] = n;
}

// pclomp NDT: the neighbour LEAVES of every list voxel in the order getNeighborhoodAtPoint{,7,1} / the radius search visit them
// (voxel_grid_covariance_omp_impl.hpp:373-442; pcl::getAllNeighborCellIndices order for the 27 cells), filtered as far as the filter
// does not depend on the query: DIRECT1/7/27 keep a leaf with >= 6 points; KDTREE (nn = 0) keeps a leaf of the centroid cloud and
// stores its float centroid for the radius test the pass applies per point.  Entry = (centroid xyz or 0, leaf index).
__device__ inline void nl_ndt_offset(int nO, int k, int& ox, int& oy, int& oz) {
  if (nO == 27) { ox = k / 9 - 1; oy = (k / 3) % 3 - 1; oz = k % 3 - 1; return; }
  ox = oy = oz = 0;
  if (k == 1) ox = 1; else if (k == 2) ox = -1; else if (k == 3) oy = 1; else if (k == 4) oy = -1; else if (k == 5) oz = 1; else if (k == 6) oz = -1;
}
__device__ inline int nl_voxel_rank(const TargetView& tg, BrickCursor& c, int vx, int vy, int vz) {
  const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
  if (bx != c.bx || by != c.by || bz != c.bz) {
    uint32_t np = 0;
    c.slot = brick_find<false>(tg, bx, by, bz, c.base, np);
    c.bx = bx; c.by = by; c.bz = bz;
  }
  if (c.slot == ~0u) return -1;
  const uint32_t li = local_index(vx, vy, vz), w = li >> 5, bit = li & 31;
  const uint32_t m = gload_u(&tg.bmask[(size_t)c.slot * 16 + w]);
  if (!((m >> bit) & 1u)) return -1;
  return (int)(c.base + gload_u16(&tg.bpref[(size_t)c.slot * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u)));
}
template <bool FILL>
__global__ void k_nl_leaf_lists(const float4* __restrict__ centres, uint32_t nd, TargetView tg, int mode, const PclLeaf* __restrict__ leaves, int nn, uint32_t* __restrict__ len,
                                const uint32_t* __restrict__ start, float4* __restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nd) return;
  const float4 c = centres[r];
  const int cx = voxel_coord(c.x, tg.res, tg.inv_res, mode), cy = voxel_coord(c.y, tg.res, tg.inv_res, mode), cz = voxel_coord(c.z, tg.res, tg.inv_res, mode);
  const int nk = nn == 0 ? 27 : nn;
  BrickCursor cur;
  uint32_t n = 0;
  const uint32_t o = FILL ? start[r] : 0u;
  for (int k = 0; k < nk; k++) {
    int ox, oy, oz;
    nl_ndt_offset(nk, k, ox, oy, oz);
    const int v = nl_voxel_rank(tg, cur, cx + ox, cy + oy, cz + oz);
    if (v < 0) continue;
    const PclLeaf* L = leaves + v;
    const bool keep = nn == 0 ? L->in_centroids != 0 : L->n >= 6;
    if (!keep) continue;
    if (FILL) out[o + n] = nn == 0 ? make_float4(L->centroid[0], L->centroid[1], L->centroid[2], __int_as_float(v)) : make_float4(0.f, 0.f, 0.f, __int_as_float(v));
    n++;
  }
  if (!FILL) len[r] = n;
}

// fast_gicp NDTCuda / VGICP of the CUDA core (ndt.hip k_ndt): the voxel of every neighbour offset of a list voxel, in the offset order
// of ndt_cuda.cu:35-88 (DIRECT1 / 7 / 27), -1 where the cell is empty -- a fixed row of nO indices per list voxel, so the pass reads
// one row behind one probe instead of probing nO cells one after the other.
__global__ void k_nl_voxel_slots(const float4* __restrict__ centres, uint32_t nd, TargetView tg, int mode, int nO, int32_t* __restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nd) return;
  const float4 c = centres[r];
  const int cx = voxel_coord(c.x, tg.res, tg.inv_res, mode), cy = voxel_coord(c.y, tg.res, tg.inv_res, mode), cz = voxel_coord(c.z, tg.res, tg.inv_res, mode);
  BrickCursor cur;
  for (int k = 0; k < nO; k++) {
    int ox, oy, oz;
    nl_ndt_offset(nO, k, ox, oy, oz);   // 7: {0, +x, -x, +y, -y, +z, -z}; 27: i, j, k loops -- the order of both references
    out[(size_t)r * nO + k] = nl_voxel_rank(tg, cur, cx + ox, cy + oy, cz + oz);
  }
}

TargetView view_of_map(const TargetMap& m) {
  TargetView v{};
  v.pts = m.pts; v.vox_start = m.vox_start; v.bricks = m.bricks; v.bmask = m.bmask; v.bpref = m.bpref; v.gvox = m.gvox;
  v.mask = m.cap - 1; v.num_points = m.num_points; v.inv_res = m.inv_res; v.res = m.res;
  return v;
}

}  // namespace

void NeighbourLists::release() {
  index.release();
  if (start) hipFree(start);
  if (pts) hipFree(pts);
  start = nullptr; pts = nullptr; start_cap = 0; pts_cap = 0; num_lists = 0; num_candidates = 0; num_neighbors = 0; valid = false; kind = 0;
}

TargetView view_of_lists(const NeighbourLists& l) {
  TargetView v = view_of_map(l.index);
  v.pts = l.pts;
  v.vox_start = l.start;
  v.num_points = (uint32_t)l.num_candidates;
  return v;
}

// Build the candidate lists of `map` for the neighbourhood `nn` (7 / 19 / 27 cells); with `ndt_leaves` (pclomp NDT: nn = 0 for the
// KDTREE search, 1 / 7 / 27) the lists hold neighbour LEAVES instead of points.  Host syncs: the number of dilated voxels, the index
// build's own, the total list length.
int build_neighbour_lists(hipStream_t stream, const TargetMap& map, int nn, NeighbourLists* out, std::string* err, const PclLeaf* ndt_leaves, bool voxel_slots) {
  out->valid = false;
  const int mode = map.coord_mode;
  if (!map.valid || (mode != COORD_ROUND && mode != COORD_FLOOR_MUL && mode != COORD_FLOOR_HALF)) { *err = "neighbour lists: unsupported voxel convention"; return PCM_ERR_UNSUPPORTED; }
  if (voxel_slots ? (nn != 1 && nn != 7 && nn != 27) : ndt_leaves ? (nn != 0 && nn != 1 && nn != 7 && nn != 27) : (nn != 1 && nn != 7 && nn != 19 && nn != 27)) { *err = "neighbour lists: unsupported neighbourhood"; return PCM_ERR_INVALID_ARGUMENT; }
  const int nset = nn == 0 ? 27 : nn;   // cells of the neighbourhood (the SET is the same in the iVox and the pcl order: c_nearby's prefixes)
  const uint32_t nvox = map.num_voxels;
  const size_t nk = (size_t)nvox * nset;
  if (nk >= (1ull << 31) || (size_t)map.num_points * nset >= (1ull << 32)) { *err = "neighbour lists: map too large"; return PCM_ERR_UNSUPPORTED; }
  {   // room for them?  27 x 16 B per map point for the lists, ~56 B per (voxel, offset) key while they are built; a quarter of the
      // free memory stays untouched (the caller's next targets, the scratch of the passes)
    size_t free_b = 0, total_b = 0;
    const size_t need = ((ndt_leaves || voxel_slots) ? (size_t)nvox : (size_t)map.num_points) * nset * sizeof(float4) + nk * 56 + ((size_t)64 << 20);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b - free_b / 4) {
      *err = "neighbour lists: " + std::to_string(need >> 20) + " MiB needed, " + std::to_string(free_b >> 20) + " MiB of device memory free";
      return PCM_ERR_HIP;
    }
  }
  uint64_t *keys = nullptr, *keys_s = nullptr;
  uint32_t *flag = nullptr, *pos = nullptr, *d_cnt = nullptr, *len = nullptr;
  float4* centres = nullptr;
  void *tmp = nullptr, *tmp2 = nullptr, *tmp3 = nullptr;
  size_t tmp_bytes = 0, tmp2_bytes = 0, tmp3_bytes = 0;
  uint32_t h_cnt[2] = {0, 0};
  int rc = PCM_OK;
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); rc = PCM_ERR_HIP; goto done; } \
  } while (0)
  {
    const unsigned nb = (unsigned)((nk + 255) / 256);
    CK(hipMallocAsync(reinterpret_cast<void**>(&keys), sizeof(uint64_t) * nk, stream));
    CK(hipMallocAsync(reinterpret_cast<void**>(&keys_s), sizeof(uint64_t) * nk, stream));
    CK(hipMallocAsync(reinterpret_cast<void**>(&flag), sizeof(uint32_t) * nk, stream));
    CK(hipMallocAsync(reinterpret_cast<void**>(&pos), sizeof(uint32_t) * nk, stream));
    CK(hipMallocAsync(reinterpret_cast<void**>(&d_cnt), sizeof(uint32_t) * 2, stream));
    k_nl_keys<<<(nvox + 255) / 256, 256, 0, stream>>>(map.pts, map.vox_start, nvox, map.res, map.inv_res, mode, nset, keys);
    CK(hipGetLastError());
    CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys_s, nk, 0, 64, stream));
    CK(hipMallocAsync(&tmp, tmp_bytes, stream));
    CK(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, keys_s, nk, 0, 64, stream));
    k_nl_flags<<<nb, 256, 0, stream>>>(keys_s, (uint32_t)nk, flag);
    CK(hipGetLastError());
    CK(rocprim::exclusive_scan(nullptr, tmp2_bytes, flag, pos, 0u, nk, rocprim::plus<uint32_t>(), stream));
    CK(hipMallocAsync(&tmp2, tmp2_bytes, stream));
    CK(rocprim::exclusive_scan(tmp2, tmp2_bytes, flag, pos, 0u, nk, rocprim::plus<uint32_t>(), stream));
    CK(hipMallocAsync(reinterpret_cast<void**>(&centres), sizeof(float4) * nk, stream));   // at most one per key
    k_nl_centres<<<nb, 256, 0, stream>>>(keys_s, flag, pos, (uint32_t)nk, map.res, mode == COORD_ROUND ? 0.f : mode == COORD_FLOOR_MUL ? 0.5f : 1.0f, centres, d_cnt);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(&h_cnt[0], d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    uint32_t nd = h_cnt[0];
    if (nd == 0) { *err = "neighbour lists: empty map"; rc = PCM_ERR_NO_INPUT; goto done; }
    // the list index: a brick hash over the stand-in points (one per dilated voxel)
    rc = build_target_map(stream, centres, &nd, map.res, mode, false, 0u, &out->index, err);
    if (rc != PCM_OK) goto done;
    if (out->index.num_voxels != nd || out->index.num_points != nd) { *err = "neighbour lists: index does not hold one voxel per list"; rc = PCM_ERR_INTERNAL; goto done; }
    if (voxel_slots) {   // a fixed row of nn voxel indices per list voxel, kept in the `pts` allocation
      const size_t words = (size_t)nd * nn, cells = (words + 3) / 4 + 4;
      if (out->pts_cap < cells) {
        if (out->pts) hipFree(out->pts);
        out->pts = nullptr; out->pts_cap = 0;
        CK(hipMalloc(reinterpret_cast<void**>(&out->pts), sizeof(float4) * cells));
        out->pts_cap = cells;
      }
      k_nl_voxel_slots<<<(nd + 127) / 128, 128, 0, stream>>>(out->index.pts, nd, view_of_map(map), mode, nn, reinterpret_cast<int32_t*>(out->pts));
      CK(hipGetLastError());
      CK(hipStreamSynchronize(stream));
      out->num_lists = nd;
      out->num_candidates = words;
      out->num_neighbors = nn;
      out->kind = 2;
      out->valid = true;
      goto done;
    }
    // list lengths -> starts -> candidates
    if (out->start_cap < (size_t)nd + 1) {
      if (out->start) hipFree(out->start);
      out->start = nullptr; out->start_cap = 0;
      CK(hipMalloc(reinterpret_cast<void**>(&out->start), sizeof(uint32_t) * ((size_t)nd + 1)));
      out->start_cap = (size_t)nd + 1;
    }
    CK(hipMallocAsync(reinterpret_cast<void**>(&len), sizeof(uint32_t) * ((size_t)nd + 1), stream));
    CK(hipMemsetAsync(len, 0, sizeof(uint32_t) * ((size_t)nd + 1), stream));
    if (ndt_leaves) k_nl_leaf_lists<false><<<(nd + 127) / 128, 128, 0, stream>>>(out->index.pts, nd, view_of_map(map), mode, ndt_leaves, nn, len, nullptr, nullptr);
    else k_nl_lists<false><<<(nd + 127) / 128, 128, 0, stream>>>(out->index.pts, nd, view_of_map(map), nn, len, nullptr, nullptr);
    CK(hipGetLastError());
    CK(rocprim::exclusive_scan(nullptr, tmp3_bytes, len, out->start, 0u, (size_t)nd + 1, rocprim::plus<uint32_t>(), stream));
    CK(hipMallocAsync(&tmp3, tmp3_bytes, stream));
    CK(rocprim::exclusive_scan(tmp3, tmp3_bytes, len, out->start, 0u, (size_t)nd + 1, rocprim::plus<uint32_t>(), stream));
    CK(hipMemcpyAsync(&h_cnt[1], out->start + nd, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    const size_t total = h_cnt[1];
    if (out->pts_cap < total + 4) {   // + 4: the walk of k_linearize_lists reads (never uses) up to three entries past a run
      if (out->pts) hipFree(out->pts);
      out->pts = nullptr; out->pts_cap = 0;
      CK(hipMalloc(reinterpret_cast<void**>(&out->pts), sizeof(float4) * (total + 4)));
      out->pts_cap = total + 4;
    }
    CK(hipMemsetAsync(out->pts + total, 0, sizeof(float4) * 4, stream));
    if (ndt_leaves) k_nl_leaf_lists<true><<<(nd + 127) / 128, 128, 0, stream>>>(out->index.pts, nd, view_of_map(map), mode, ndt_leaves, nn, nullptr, out->start, out->pts);
    else k_nl_lists<true><<<(nd + 127) / 128, 128, 0, stream>>>(out->index.pts, nd, view_of_map(map), nn, nullptr, out->start, out->pts);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(stream));
    out->num_lists = nd;
    out->num_candidates = total;
    out->num_neighbors = nn;
    out->kind = ndt_leaves ? 1 : 0;
    out->valid = true;
  }
done:
  if (keys) (void)hipFreeAsync(keys, stream);
  if (keys_s) (void)hipFreeAsync(keys_s, stream);
  if (flag) (void)hipFreeAsync(flag, stream);
  if (pos) (void)hipFreeAsync(pos, stream);
  if (d_cnt) (void)hipFreeAsync(d_cnt, stream);
  if (len) (void)hipFreeAsync(len, stream);
  if (centres) (void)hipFreeAsync(centres, stream);
  if (tmp) (void)hipFreeAsync(tmp, stream);
  if (tmp2) (void)hipFreeAsync(tmp2, stream);
  if (tmp3) (void)hipFreeAsync(tmp3, stream);
  if (rc != PCM_OK) out->valid = false;
  return rc;
#undef CK
}

// ---------------------------------------------------------------------------
// k_linearize_lists: grid (ceil(N / 256), pairs), one scan point per lane
// ---------------------------------------------------------------------------
template <bool WRITE_PLANES>
__global__ void __launch_bounds__(256) k_linearize_lists(const PairDesc* __restrict__ descs, const PairState* __restrict__ states, KernelParams kp) {
  // the tile placement of k_linearize (kernels.hip): neighbouring tiles of a pair share an XCD (they read the same lists)
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
  const TargetView nl = d.nl;
  __shared__ __align__(16) unsigned char s_mem[kReduceLdsBytes];

  float4 pl = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
  float q[3] = {0.f, 0.f, 0.f};
  float pn_body = 0.f;
  if (live) {
    const float4 p = gload4(d.src.pts + i);
    pn_body = pcm_sqrtf_rn(p.x * p.x + p.y * p.y + p.z * p.z);
    transform(P, p, q);
    const float fx = roundf(q[0] * nl.inv_res), fy = roundf(q[1] * nl.inv_res), fz = roundf(q[2] * nl.inv_res);  // Pos2Grid  ivox3d.h:283-286
    const float lim = (float)(kCoordBias - 32);
    if (fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim) {   // also false for NaN
      const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
      uint32_t vox_base = 0, n_probe = 0;
      const uint32_t slot = brick_find<false>(nl, cx >> kBrickShift, cy >> kBrickShift, cz >> kBrickShift, vox_base, n_probe);
      if (slot != ~0u) {
        const uint32_t li = local_index(cx, cy, cz), w = li >> 5, bit = li & 31;
        const uint32_t m = gload_u(&nl.bmask[(size_t)slot * 16 + w]);
        if ((m >> bit) & 1u) {
          const uint32_t r = vox_base + gload_u16(&nl.bpref[(size_t)slot * 16 + w]) + (uint32_t)__popc(m & ((1u << bit) - 1u));
          const uint32_t s = gload_u(&nl.vox_start[r]), e = gload_u(&nl.vox_start[r + 1]);
          Best best;
          best_init(best, kp.max_range_sq);
          // the voxel's candidates in the reference's visit order, four per trip (all four loads under way before the first offer;
          // the array is padded, a load past the run is never offered)
          for (uint32_t k = s; k < e; k += 4) {
            const float4 c0 = gload4(nl.pts + k), c1 = gload4(nl.pts + k + 1), c2 = gload4(nl.pts + k + 2), c3 = gload4(nl.pts + k + 3);
            best_offer(best, c0, q, k, kp.max_range_sq);
            if (k + 1 < e) best_offer(best, c1, q, k + 1, kp.max_range_sq);
            if (k + 2 < e) best_offer(best, c2, q, k + 2, kp.max_range_sq);
            if (k + 3 < e) best_offer(best, c3, q, k + 3, kp.max_range_sq);
          }
          best_finish(best);
          if (best.m >= KMIN) {   // laser_mapping.cc:619-623
            float px[K], py[K], pz[K];
#pragma unroll
            for (int j = 0; j < K; j++) {
              float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
              if (j < best.m) mp = gload4(nl.pts + best.i[j]);
              px[j] = mp.x; py[j] = mp.y; pz[j] = mp.z;
            }
            float4 fit;
            if (esti_plane(px, py, pz, best.m, kp.plane_threshold, &fit)) pl = fit;
          }
        }
      }
    }
  }
  residual_and_reduce<WRITE_PLANES>(d, i, tile_x, live, pl, q, pn_body, s_mem);
}

void launch_linearize_lists(hipStream_t stream, const PairDesc* d_descs, const PairState* d_states, const KernelParams& kp, int npairs, bool write_planes) {
  dim3 grid((unsigned)kp.tiles_per_pair, (unsigned)npairs);
  if (write_planes) k_linearize_lists<true><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
  else k_linearize_lists<false><<<grid, 256, 0, stream>>>(d_descs, d_states, kp);
}

}  // namespace pcm

"""Per-wave statistics of the k_linearize search on a bench pair (CPU, numpy): distinct query voxels per wave, loop trips of the
per-cell lock-step search (what the kernel runs), of a flat per-lane candidate stream, of a voxel-cooperative broadcast search."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
synth = importlib.import_module("pointcloud-slam_amd.synth")
NEARBY = np.array([(0,0,0),(-1,0,0),(1,0,0),(0,1,0),(0,-1,0),(0,0,-1),(0,0,1),(1,1,0),(-1,1,0),(1,-1,0),(-1,-1,0),(1,0,1),(-1,0,1),(1,0,-1),
                   (-1,0,-1),(0,1,1),(0,-1,1),(0,1,-1),(0,-1,-1),(1,1,1),(-1,1,1),(1,-1,1),(1,1,-1),(-1,-1,1),(-1,1,-1),(1,-1,-1),(-1,-1,-1)], np.int64)
def key(v):
    return ((v[:, 0] + (1 << 20)) << 42) | ((v[:, 1] + (1 << 20)) << 21) | (v[:, 2] + (1 << 20))
def spread10(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v
pid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_scan = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n_map = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
p = synth.make_pair(pid, n_scan, n_map)
res = 0.5
T = np.asarray(p.guess, np.float32)
q = (p.scan[:, :3].astype(np.float32) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
vq = np.round(q / res).astype(np.int64)
c = vq - np.round(T[:3, 3] / res).astype(np.int64)
c = np.clip(c, -512, 511) + 512
m = spread10(c[:, 0]) | (spread10(c[:, 1]) << 1) | (spread10(c[:, 2]) << 2)
order = np.argsort(m, kind="stable")
vq = vq[order]
vm = np.round(p.submap[:, :3].astype(np.float32) / res).astype(np.int64)
mk, mc = np.unique(key(vm), return_counts=True)
# per point, per cell: candidates
N = len(vq)
lens = np.zeros((N, 27), np.int32)
for g in range(27):
    k = key(vq + NEARBY[g])
    j = np.searchsorted(mk, k)
    j[j >= len(mk)] = 0
    hit = mk[j] == k
    lens[:, g] = np.where(hit, mc[j], 0)
tot = lens.sum(1)
print("points", N, "candidates/point mean %.2f max %d; occupied cells/point %.2f" % (tot.mean(), tot.max(), (lens > 0).mean(1).mean() * 27))
W = N // 64
L = lens[:W * 64].reshape(W, 64, 27)
qk = key(vq[:W * 64]).reshape(W, 64)
cur = L.max(1).sum(1)                       # sum_g max_lane len_g
cur_cells = (L.max(1) > 0).sum(1)           # cells entered by the wave
flat = L.sum(2).max(1)                      # max_lane sum_g
kd = np.array([len(np.unique(r)) for r in qk])
runs = (np.diff(qk, axis=1) != 0).sum(1) + 1
coop = np.zeros(W, np.int64); union = np.zeros(W, np.int64)
for w in range(W):
    u, idx = np.unique(qk[w], return_index=True)
    coop[w] = L[w, idx].sum()
    cells = set()
    for v in vq[w * 64 + idx]:
        for g in range(27):
            cells.add(tuple(v + NEARBY[g]))
    ck = key(np.array(list(cells), np.int64))
    j = np.searchsorted(mk, ck); j[j >= len(mk)] = 0
    union[w] = mc[j][mk[j] == ck].sum()
def s(name, a): print("%-34s mean %.1f  p50 %.0f  p90 %.0f  max %d" % (name, a.mean(), np.percentile(a, 50), np.percentile(a, 90), a.max()))
s("distinct query voxels / wave", kd); s("voxel runs / wave (consecutive)", runs)
s("trips now (sum_g max_lane len)", cur); s("cells entered now", cur_cells)
s("trips flat (max_lane sum_g len)", flat); s("trips voxel-coop (sum_vox cand)", coop); s("union brute force points", union)
T256 = N // 256
qk4 = key(vq[:T256 * 256]).reshape(T256, 256)
s("distinct query voxels / tile", np.array([len(np.unique(r)) for r in qk4]))

"""Small profiling driver: a few batched aligns of N pairs (no torch.distributed, no fork)."""
import sys, os, importlib, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--scan", type=int, default=100000)
ap.add_argument("--map", type=int, default=1000000)
ap.add_argument("--flags", type=int, default=0)
a = ap.parse_args()
synth = importlib.import_module("pointcloud-slam_amd.synth")
pairs = [synth.make_pair(i, a.scan, a.map) for i in range(a.pairs)]
import pointcloud_slam_amd as pcm
regs = []
for p in pairs:
    r = pcm.P2PlaneRegistration(0, optimizer="GN", voxel_resolution=0.5, num_neighbors=27, flags=a.flags)
    r.set_input_target(p.submap); r.set_input_source(p.scan); regs.append(r)
g = np.stack([p.guess for p in pairs])
for s in range(a.steps):
    res = pcm.align_batch(regs, g)
print([r.num_linearize for r in res])
regs[0].set_profiling(4)
res = pcm.align_batch(regs, g)
pc = regs[0].phase_cycles()
n = max(1, pc[7])
names = ['load+box', 'probe', 'scan', 'stage', 'search', 'fit', 'jobs']
print('tiles', pc[7]); print({k: round(v / n) for k, v in zip(names, pc[:7])}, 'ticks/tile')

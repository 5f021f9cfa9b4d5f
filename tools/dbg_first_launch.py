"""Diagnostic: one batched align of 4 pairs (for per-dispatch PMC comparisons of kernel variants)."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
synth = importlib.import_module("pointcloud-slam_amd.synth")
pairs = [synth.make_pair(i, 100000, 1000000) for i in range(4)]
import pointcloud_slam_amd as pcm
regs = []
for p in pairs:
    r = pcm.P2PlaneRegistration(0, optimizer="GN", voxel_resolution=0.5, num_neighbors=27, max_iterations=2)
    r.set_input_target(p.submap); r.set_input_source(p.scan); regs.append(r)
res = pcm.align_batch(regs, np.stack([p.guess for p in pairs]))
print([r.num_linearize for r in res])

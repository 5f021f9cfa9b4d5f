import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
synth = importlib.import_module("pointcloud-slam_amd.synth")
import pointcloud_slam_amd as pcm
from oracle import Oracle
p = synth.make_pair(3, 8000, 80000)
for reg in ("PLANE", "MIN_EIG"):
    g = pcm.VgicpCudaRegistration(0, optimizer="LM", regularization=reg)
    cfg = g.config
    o = Oracle("VGICP_CUDA", "LM", voxel_resolution=cfg.voxel_resolution, num_neighbors=cfg.num_neighbors, k_correspondences=cfg.k_correspondences, regularization=cfg.regularization)
    o.set_input_target(p.submap); o.set_input_source(p.scan); g.set_input_target(p.submap); g.set_input_source(p.scan)
    for target in (False, True):
        c0, c1 = o.covariances(target), g.get_covariances(target)
        d = np.abs(c1 - c0).reshape(len(c0), -1).max(axis=1)
        print(reg, target, "n", len(c0), "exact-equal points", int((d == 0).sum()), "max", d.max(), "quantiles", np.quantile(d, [0.5, 0.9, 0.99, 0.999]))
        i = int(np.argmax(d)); print("  worst", i, c0[i].ravel(), c1[i].ravel())

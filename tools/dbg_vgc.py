import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
synth = importlib.import_module("pointcloud-slam_amd.synth")
import pointcloud_slam_amd as pcm
from oracle import Oracle
p = synth.make_pair(3, 8000, 80000)
g = pcm.VgicpCudaRegistration(0); cfg = g.config
o = Oracle("VGICP_CUDA", "LM", voxel_resolution=cfg.voxel_resolution, num_neighbors=cfg.num_neighbors, k_correspondences=cfg.k_correspondences, regularization=cfg.regularization)
o.set_input_target(p.submap); o.set_input_source(p.scan); g.set_input_target(p.submap); g.set_input_source(p.scan)
for tgt in (False, True):
    c0, c1 = o.covariances(tgt), g.get_covariances(tgt)
    d = np.abs(c1 - c0).reshape(len(c0), -1).max(axis=1)
    print("target" if tgt else "source", "points differing at all:", int((d > 0).sum()), "of", len(d), "max", d.max())
    i = int(np.argmax(d)); print(" worst", i, c0[i].ravel(), c1[i].ravel())
T = p.guess.astype(np.float64)
c0, H0, b0 = o.linearize(T); c1, H1, b1, inl = g.evaluate_cost(T)
print("H rel", np.abs(H1-H0).max()/np.abs(H0).max(), "b rel", np.abs(b1-b0).max()/np.abs(b0).max(), "c rel", abs(c1-c0)/abs(c0), inl, o.num_inliers)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import pose_error
from oracle.loader import result_T
print("lower-triangle H diff rel", np.abs(np.tril(H1 - H0)).max() / np.abs(H0).max())
for opt, nn in (("LM", 1), ("GN", 7), ("GN", 1)):
    g = pcm.VgicpCudaRegistration(0, optimizer=opt, num_neighbors=nn); cfg = g.config
    o = Oracle("VGICP_CUDA", opt, voxel_resolution=cfg.voxel_resolution, num_neighbors=nn, k_correspondences=cfg.k_correspondences, regularization=cfg.regularization)
    o.set_input_target(p.submap); o.set_input_source(p.scan); g.set_input_target(p.submap); g.set_input_source(p.scan)
    o.enable_trace(256)
    ro, rg = o.align(p.guess), g.align(p.guess)
    print(opt, nn, "iters", rg.iterations, ro.iterations, "pose diff", pose_error(result_T(ro), rg.T64), "evals", rg.num_linearize, ro.num_linearize, rg.num_compute_error, ro.num_compute_error)
    # replay: compare linearize at the oracle's final pose
    Tf = result_T(ro)
    c0, H0, b0 = o.linearize(Tf); c1, H1, b1, inl = g.evaluate_cost(Tf)
    print("   at final pose: tril H rel", np.abs(np.tril(H1 - H0)).max() / np.abs(H0).max(), "b rel", np.abs(b1 - b0).max() / np.abs(b0).max(), "inl", inl, o.num_inliers)
print("---- trial cost")
g = pcm.VgicpCudaRegistration(0, optimizer="LM", num_neighbors=1); cfg = g.config
o = Oracle("VGICP_CUDA", "LM", voxel_resolution=cfg.voxel_resolution, num_neighbors=1, k_correspondences=cfg.k_correspondences, regularization=cfg.regularization, max_iterations=1)
o.set_input_target(p.submap); o.set_input_source(p.scan); g.set_input_target(p.submap); g.set_input_source(p.scan)
T = p.guess.astype(np.float64)
c0, H0, b0 = o.linearize(T); c1, H1, b1, inl = g.evaluate_cost(T)
for d in ([0.02, -0.01, 0.01], [0.001, 0.0, 0.0], [0.2, 0.1, -0.1]):
    T2 = T.copy(); T2[:3, 3] += d
    e0, e1 = o.compute_error(T2), g.compute_error(T2)
    print(d, e0, e1, abs(e1 - e0) / abs(e0))
o.enable_trace(64)
ro = o.align(p.guess)
print("oracle trace", o.trace()[:6] if hasattr(o, "trace") else None)
g2 = pcm.VgicpCudaRegistration(0, optimizer="LM", num_neighbors=1, max_iterations=1); g2.set_input_target(p.submap); g2.set_input_source(p.scan)
rg = g2.align(p.guess)
print("gpu", rg.num_linearize, rg.num_compute_error, rg.cost, "oracle", ro.num_linearize, ro.num_compute_error, ro.cost)
print(np.abs(rg.T64 - result_T(ro)).max())

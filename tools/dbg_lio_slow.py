"""Diagnostic of the config-5 frame: the first re-matching ObsModel call of a frame (= map update + search) against a second one on the
up-to-date map (= search alone), with and without the LRU capacity, raw 100k-point frames and frames down-sampled at 0.5 m."""
import sys, os, time, importlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial.transform import Rotation as R
synth = importlib.import_module("pointcloud-slam_amd.synth")
M, N = 5_000_000, 100_000
scene = synth.scene_for_points(1234, M, 8.0)
submap = synth.sample_submap(scene, M, 4321)
T0 = synth.sensor_pose(scene, 77)
import torch
import pointcloud_slam_amd as pcm
for cap, leaf in ((1_000_000, 0.0), (0, 0.0), (1_000_000, 0.5)):
    g = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27, map_capacity=cap)
    g.set_input_target(torch.from_numpy(submap).cuda())
    for f in range(5):
        T = T0.copy(); T[:3, 3] += T[:3, 0] * 0.25 * f
        sc, ex = synth.livox_scan(scene, T, N, 555 + f, point_filter_num=1)
        st = (R.from_matrix(T[:3, :3]).as_quat(), T[:3, 3].copy(), np.array([0, 0, 0, 1.0]), np.zeros(3))
        if leaf > 0:
            sc = g.voxel_downsample(np.ascontiguousarray(sc), leaf)
        g.set_input_source(torch.from_numpy(np.ascontiguousarray(sc)).cuda())
        torch.cuda.synchronize(); t = time.perf_counter(); g.obs_model(*st, False, True); torch.cuda.synchronize(); t_match = time.perf_counter() - t
        t = time.perf_counter(); g.obs_model(*st, False, True); torch.cuda.synchronize(); t_match2 = time.perf_counter() - t
        s = g.stats()
        added = g.map_incremental(*st, 0.5, True)
        print("cap %d leaf %.1f frame %d: scan %d pts, first match %.2f ms = update %.2f + search %.2f ms | voxels %d added %d hazards %d" % (
            cap, leaf, f, len(sc), 1e3 * t_match, 1e3 * (t_match - t_match2), 1e3 * t_match2, s["target_voxels"], added, s["lru_batch_hazards"]), flush=True)

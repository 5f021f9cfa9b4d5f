"""BASELINE configs[4]: jueying_lio loop -- 20 Hz scan stream against a sliding submap (5M points,
1M-voxel LRU capacity) with incremental voxel-hash rebuild, 1x MI355X.
Per frame: new scan -> 4 x ObsModel (re-match, then 3 updates as the IEKF would: max_iteration 3-4,
config/livox.yaml:41) -> MapIncremental -> (lazy) map rebuild at the next frame's first match.
The EKF algebra itself (23x23, host) is outside the path; the state is the ground-truth trajectory."""
import sys, os, time, importlib, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial.transform import Rotation as R
ap = argparse.ArgumentParser()
ap.add_argument("--map", type=int, default=5_000_000)
ap.add_argument("--scan", type=int, default=100_000)
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--capacity", type=int, default=1_000_000)
a = ap.parse_args()
synth = importlib.import_module("pointcloud-slam_amd.synth")
scene = synth.scene_for_points(1234, a.map, 8.0)
submap = synth.sample_submap(scene, a.map, 4321)
T0 = synth.sensor_pose(scene, 77)
scans, states = [], []
for f in range(a.frames):
    T = T0.copy(); T[:3, 3] += T[:3, 0] * 0.25 * f        # 5 m/s at 20 Hz
    sc, _ = synth.livox_scan(scene, T, a.scan, 555 + f)
    scans.append(sc)
    states.append((R.from_matrix(T[:3, :3]).as_quat(), T[:3, 3].copy(), np.array([0, 0, 0, 1.0]), np.zeros(3)))
import torch
import pointcloud_slam_amd as pcm
g = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27, map_capacity=a.capacity)
t0 = time.perf_counter(); g.set_input_target(torch.from_numpy(submap).cuda()); g.set_input_source(torch.from_numpy(scans[0]).cuda())
g.obs_model(*states[0], False, True); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
d_scans = [torch.from_numpy(s).cuda() for s in scans]
per = {"match": [], "update": [], "map_incremental": [], "frame": [], "added": [], "n_eff": [], "undistort": [], "downsample": [], "downsampled_points": []}
# scan pre-processing of the raw frame (timed on its own: host buffers in and out, as the reference hands them over)
K = 11
poses = np.zeros((K, 22)); vel = T0[:3, 0] * 5.0
for k in range(K):
    poses[k, 0] = 0.005 * k; poses[k, 7:10] = vel; poses[k, 10:13] = vel * 0.005 * k; poses[k, 13:22] = np.eye(3).ravel()
raw = np.zeros((a.scan, 12), np.float32)
for f in range(1, a.frames):
    tf = time.perf_counter()
    g.set_input_source(d_scans[f])
    t = time.perf_counter(); H, h, n, s2, ok = g.obs_model(*states[f], False, True); per["match"].append(time.perf_counter() - t)   # includes the map rebuild
    t = time.perf_counter()
    for _ in range(3):
        g.obs_model(*states[f], False, False)
    per["update"].append(time.perf_counter() - t)
    t = time.perf_counter(); added = g.map_incremental(*states[f], 0.5, True); per["map_incremental"].append(time.perf_counter() - t)
    per["frame"].append(time.perf_counter() - tf); per["added"].append(added); per["n_eff"].append(n)
    raw[:, :4] = scans[f]; raw[:, 10] = np.linspace(0.0, 50.0, a.scan, dtype=np.float32)
    t = time.perf_counter(); g.undistort(raw, 10, poses, [0, 0, 0, 1.0], poses[-1, 10:13], [0, 0, 0, 1.0], [0, 0, 0]); per["undistort"].append(time.perf_counter() - t)
    t = time.perf_counter(); ds = g.voxel_downsample(raw, 0.5); per["downsample"].append(time.perf_counter() - t); per["downsampled_points"].append(len(ds))
out = {k: float(np.median(v)) for k, v in per.items()}
out.update({"first_frame_s": t_first, "frames": a.frames - 1, "hz_sustained": 1.0 / float(np.mean(per["frame"])), "map_points_end": int(len(g.get_target())),
            "target_voxels": g.stats()["target_voxels"], "worst_frame_ms": 1e3 * float(np.max(per["frame"]))})
print(json.dumps(out, indent=1))

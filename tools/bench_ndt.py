"""BASELINE configs[3] (models: NDT_OMP = pclomp NDT as jueying_slam calls it, NDT_D2D / NDT_P2D = fast_gicp NDTCuda): NDT variant -- 100k-pt scan vs 0.5 m voxel NDT grid of a 10M-pt map, 1x MI355X.
Reports build time, registrations/s (target map reused), and the oracle on the host cores."""
import sys, os, time, importlib, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--map", type=int, default=10_000_000)
ap.add_argument("--scan", type=int, default=100_000)
ap.add_argument("--scans", type=int, default=8, help="scans registered against the one map (one batch)")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cpu", type=int, default=1)
ap.add_argument("--lib", default="", help="A/B: load this build of the library")
ap.add_argument("--eps", type=float, default=0.01, help="pclomp NDT transformation epsilon: 0.01 at the call sites (jueying_slam/src/localization.cpp:170-175), 0.1 the class default")
ap.add_argument("--models", default="NDT_OMP,NDT_OMP_KDTREE,NDT_D2D,NDT_P2D")
ap.add_argument("--flags", type=int, default=0, help="pcm_config.flags (64 = PCM_FLAG_NO_NEIGHBOUR_LISTS: every cell looked up, the A/B partner)")
a = ap.parse_args()
synth = importlib.import_module("pointcloud-slam_amd.synth")
scene = synth.scene_for_points(1234, a.map, 60.0)          # >6 points per 0.5 m voxel on surfaces
submap = synth.sample_submap(scene, a.map, 4321)
scans, guesses, gts = [], [], []
for i in range(a.scans):
    T = synth.sensor_pose(scene, 77 + i)
    sc, _ = synth.livox_scan(scene, T, a.scan, 555 + i)
    scans.append(sc); gts.append(T); guesses.append(synth.perturb_pose(T, 99 + i).astype(np.float32))
import torch
import pointcloud_slam_amd as pcm
if a.lib:
    pcm.capi.library_path = lambda: os.path.abspath(a.lib)
d_map = torch.from_numpy(submap).cuda()
out = {}
for mname in a.models.split(","):
    model = "NDT_OMP" if mname.startswith("NDT_OMP") else mname
    nn = 0 if mname == "NDT_OMP_KDTREE" else 7
    regs = []
    t0 = time.perf_counter()
    for sc in scans:
        r = pcm.PclNdtRegistration(0, voxel_resolution=0.5, num_neighbors=nn, translation_eps=a.eps, flags=a.flags) if model == "NDT_OMP" else pcm.NdtRegistration(0, model=model, voxel_resolution=0.5, num_neighbors=7, flags=a.flags)
        r.set_input_target(d_map); regs.append(r)
    d_scans = [torch.from_numpy(s).cuda() for s in scans]
    for r, s in zip(regs, d_scans):
        r.set_input_source(s)
    res = pcm.align_batch(regs, np.stack(guesses))      # cold: builds every object's target grid (same map, 8 objects)
    torch.cuda.synchronize(); t_cold = time.perf_counter() - t0
    for r, s in zip(regs, d_scans):                      # second registration against every grid: its neighbour-leaf lists are built here (untimed)
        r.set_input_source(s)
    pcm.align_batch(regs, np.stack(guesses))
    torch.cuda.synchronize(); t_lists = time.perf_counter() - t0 - t_cold
    t0 = time.perf_counter()
    for _ in range(a.reps):
        for r, s in zip(regs, d_scans):
            r.set_input_source(s)
        res = pcm.align_batch(regs, np.stack(guesses))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
    errs = [float(np.linalg.norm((np.linalg.inv(g) @ x.T64)[:3, 3])) for g, x in zip(gts, res)]
    out[mname] = {"registrations_per_s": a.scans / dt, "ms_per_batch": 1e3 * dt, "cold_s": t_cold, "second_batch_with_list_build_s": t_lists, "iterations": [x.iterations for x in res], "evaluations": [x.num_linearize for x in res],
                  "converged": [int(x.converged) for x in res], "transformation_epsilon": a.eps if model == "NDT_OMP" else None, "err_vs_gt_m": [round(e, 3) for e in errs], "target_voxels": regs[0].stats()["target_voxels"]}
    if a.cpu:
        from oracle import Oracle
        from oracle.loader import result_T
        okw = dict(translation_eps=a.eps, max_iterations=35) if model == "NDT_OMP" else {}
        o = Oracle(model, "LM", voxel_resolution=0.5, num_neighbors=nn, num_threads=min(16, len(os.sched_getaffinity(0))), **okw)
        t0 = time.perf_counter(); o.set_input_target(submap); o.set_input_source(scans[0])
        if model == "NDT_OMP":
            o.ndt_leaf(submap[0, :3])
        else:
            o.linearize(guesses[0].astype(np.float64))
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter(); ro = o.align(guesses[0]); t_al = time.perf_counter() - t0
        D = np.linalg.inv(result_T(ro)) @ res[0].T64
        out[mname]["cpu_oracle"] = {"build_s": t_build, "align_s": t_al, "registrations_per_s": 1.0 / t_al, "pose_diff_m": float(np.linalg.norm(D[:3, 3])),
                                    "pose_diff_rad": float(np.linalg.norm(D[:3, :3] - np.eye(3)))}
    del regs
print(json.dumps(out, indent=1))

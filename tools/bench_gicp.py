"""GICP / VGICP at the headline size: 100k-pt scans vs a 1M-pt submap, 1x MI355X.
Reports the covariance build time, registrations/s (fresh scan per registration: source covariances
are recomputed, target covariances reused like the reference's cached target_covs_), and the oracle."""
import sys, os, time, importlib, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--map", type=int, default=1_000_000)
ap.add_argument("--scan", type=int, default=100_000)
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cpu", type=int, default=1)
ap.add_argument("--models", default="GICP,GICP_unbounded,VGICP_direct1,VGICP_direct7")
ap.add_argument("--lib", default="", help="A/B: load this build of the library")
a = ap.parse_args()
synth = importlib.import_module("pointcloud-slam_amd.synth")
pairs = [synth.make_pair(i, a.scan, a.map) for i in range(a.pairs)]
import torch
import pointcloud_slam_amd as pcm
if a.lib:
    pcm.capi.library_path = lambda: os.path.abspath(a.lib)
out = {}
for name, cls, kw in (("GICP", pcm.GicpRegistration, {"max_corr_dist": 1.0}), ("GICP_unbounded", pcm.GicpRegistration, {}),
                      ("VGICP_direct1", pcm.VgicpRegistration, {}), ("VGICP_direct7", pcm.VgicpRegistration, {"num_neighbors": 7})):
    if name not in a.models.split(","):
        continue
    regs = []
    d_scans = [torch.from_numpy(p.scan).cuda() for p in pairs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for p in pairs:
        r = cls(0, **kw); r.set_input_target(p.submap); regs.append(r)
    for r, s in zip(regs, d_scans):
        r.set_input_source(s)
    t1 = time.perf_counter()
    n_cov = regs[0].get_covariances(True).shape[0]           # builds map + target covariances of object 0
    torch.cuda.synchronize(); t_cov = time.perf_counter() - t1
    guesses = np.stack([p.guess for p in pairs])
    res = pcm.align_batch(regs, guesses)
    torch.cuda.synchronize(); t_cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(a.reps):
        for r, s in zip(regs, d_scans):
            r.set_input_source(s)
        res = pcm.align_batch(regs, guesses)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
    errs = [float(np.linalg.norm((np.linalg.inv(p.T_gt) @ x.T64)[:3, 3])) for p, x in zip(pairs, res)]
    out[name] = {"registrations_per_s": a.pairs / dt, "ms_per_batch": 1e3 * dt, "cold_s": t_cold, "target_cov_build_s_incl_map": t_cov, "target_points": n_cov,
                 "iterations": [x.iterations for x in res], "converged": [int(x.converged) for x in res], "err_vs_gt_m": [round(e, 3) for e in errs]}
    if a.cpu:
        from oracle import Oracle
        from oracle.loader import result_T
        cfg = regs[0].config
        o = Oracle(cfg.model == 1 and "GICP" or "VGICP", "LM", voxel_resolution=cfg.voxel_resolution, num_neighbors=cfg.num_neighbors, max_corr_dist=float(cfg.max_corr_dist),
                   num_threads=min(16, len(os.sched_getaffinity(0))))
        p = pairs[0]
        t0 = time.perf_counter(); o.set_input_target(p.submap); o.set_input_source(p.scan); o.linearize(p.guess.astype(np.float64)); t_build = time.perf_counter() - t0
        t0 = time.perf_counter(); ro = o.align(p.guess); t_al = time.perf_counter() - t0
        D = np.linalg.inv(result_T(ro)) @ res[0].T64
        out[name]["cpu_oracle"] = {"build_s": t_build, "align_s": t_al, "registrations_per_s": 1.0 / t_al, "iterations": ro.iterations, "pose_diff_m": float(np.linalg.norm(D[:3, 3])),
                                   "pose_diff_rad": float(np.linalg.norm(D[:3, :3] - np.eye(3)))}
    del regs
print(json.dumps(out, indent=1))

"""Randomised parity sweep (GPU vs oracle) beyond the fixed test cases: many pairs of varying size, density,
neighbourhood and optimizer; reports per-point plane mismatches and pose differences."""
import sys, os, importlib, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import pose_error
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=24)
a = ap.parse_args()
synth = importlib.import_module("pointcloud-slam_amd.synth")
import pointcloud_slam_amd as pcm
from oracle import Oracle
from oracle.loader import result_T
rng = np.random.default_rng(2026)
worst = {"plane_mismatch_points": 0, "dt": 0.0, "dr": 0.0, "iter_mismatch": 0, "H_rel": 0.0}
rows = []
for t in range(a.pairs):
    n_scan = int(rng.integers(3000, 40000)); m_map = int(rng.integers(30000, 400000))
    dens = float(rng.choice([4.0, 8.0, 20.0, 60.0])); nn = int(rng.choice([1, 7, 19, 27])); opt = str(rng.choice(["GN", "LM"]))
    res = float(rng.choice([0.3, 0.5, 1.0]))
    p = synth.make_pair(1000 + t, n_scan, m_map, density=dens)
    o = Oracle("P2PLANE", opt, voxel_resolution=res, num_neighbors=nn)
    o.set_input_target(p.submap); o.set_input_source(p.scan)
    g = pcm.P2PlaneRegistration(0, optimizer=opt, voxel_resolution=res, num_neighbors=nn)
    g.set_input_target(p.submap); g.set_input_source(p.scan)
    T = p.guess.astype(np.float64)
    c0, H0, b0 = o.linearize(T)
    inl0 = int(o.num_inliers)   # at the guess, like the GPU figure beside it
    c1, H1, b1, inl = g.evaluate_cost(T)
    pl0, sel0 = o.get_planes(len(p.scan))
    pl1 = g.get_planes(len(p.scan))
    sel1 = ~np.isnan(pl1[:, 0])
    both = sel0 & sel1
    bad = int((sel0 != sel1).sum() + (pl0[both] != pl1[both]).any(axis=1).sum())
    ro, rg = o.align(p.guess), g.align(p.guess)
    dt, dr = pose_error(result_T(ro), rg.T64)
    hrel = float(np.abs(H1 - H0).max() / max(np.abs(H0).max(), 1e-300))
    rows.append(dict(t=t, n=n_scan, m=m_map, dens=dens, nn=nn, opt=opt, res=res, bad_planes=bad, inl=(int(inl), inl0), dt=dt, dr=dr, it=(rg.iterations, ro.iterations), H_rel=hrel))
    worst["plane_mismatch_points"] += bad; worst["dt"] = max(worst["dt"], dt); worst["dr"] = max(worst["dr"], dr)
    worst["iter_mismatch"] += int(rg.iterations != ro.iterations); worst["H_rel"] = max(worst["H_rel"], hrel)
    worst["inlier_mismatch"] = worst.get("inlier_mismatch", 0) + int(int(inl) != inl0)
    print("pair %d done" % t, file=sys.stderr, flush=True)
print(json.dumps({"worst": worst, "rows": rows}, indent=1))

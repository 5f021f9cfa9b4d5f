"""pclomp GICP-BFGS functor: evaluations/s of fdf over one packed correspondence set, 1x MI355X,
with the serial oracle timed beside it.  Algorithmic bytes: one 64-byte record per correspondence per evaluation."""
import sys, os, time, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=1_000_000)
ap.add_argument("--evals", type=int, default=300)
a = ap.parse_args()
from test_gicp_bfgs import _problem
import pointcloud_slam_amd as pcm
from oracle import loader as L
src, tgt, isrc, itgt, maha, base, x = _problem(11, n=a.m + a.m // 4, m=a.m)
g = pcm.GicpRegistration(0)
t0 = time.perf_counter(); g.gicp_bfgs_set_correspondences(src, tgt, isrc, itgt, maha); t_set = time.perf_counter() - t0
t0 = time.perf_counter(); g.gicp_bfgs_set_correspondences(src, tgt, isrc, itgt, maha); t_set = min(t_set, time.perf_counter() - t0)
for _ in range(10):
    g.gicp_bfgs_fdf(base, x, 2)
t0 = time.perf_counter()
for k in range(a.evals):
    f, gr = g.gicp_bfgs_fdf(base, x * (1.0 + 1e-3 * k), 2)
dt = (time.perf_counter() - t0) / a.evals
t0 = time.perf_counter(); fo, go = L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, x * (1.0 + 1e-3 * (a.evals - 1)), 2); t_cpu = time.perf_counter() - t0
print(json.dumps({"correspondences": a.m, "gpu_us_per_evaluation_incl_host_round_trip": 1e6 * dt, "gpu_GBps_algorithmic": 64.0 * a.m / dt / 1e9,
                  "set_correspondences_ms_host_buffers": 1e3 * t_set, "oracle_serial_ms_per_evaluation": 1e3 * t_cpu, "speedup": t_cpu / dt,
                  "f_rel_diff": abs(f - fo) / abs(fo), "g_rel_diff": float(np.abs(gr - go).max() / np.abs(go).max())}, indent=1))

# ---- the correspondence step of the outer loop on the device (gicp_omp_impl.hpp:405-472), 100k-pt scan vs 1M-pt map ----
import importlib
synth = importlib.import_module("pointcloud-slam_amd.synth")
from oracle import Oracle
p = synth.make_pair(0, 100000, 1000000)
gc = pcm.GicpRegistration(0, regularization="PCLOMP", max_corr_dist=1.0)
gc.set_input_target(p.submap); gc.set_input_source(p.scan)
I4 = np.eye(4, dtype=np.float32)
t0 = time.perf_counter(); m = gc.gicp_bfgs_update_correspondences(I4, p.guess); t_first = time.perf_counter() - t0     # builds maps + covariances
ts = []
for _ in range(10):
    t0 = time.perf_counter(); m = gc.gicp_bfgs_update_correspondences(I4, p.guess); ts.append(time.perf_counter() - t0)
o = Oracle("GICP", "LM", regularization="PCLOMP", max_corr_dist=1.0, num_threads=1)
o.set_input_target(p.submap); o.set_input_source(p.scan)
o.gicp_bfgs_correspondences(I4, p.guess)                      # covariances + grid (excluded, like the GPU's first call)
t0 = time.perf_counter(); isrc0, itgt0, M0 = o.gicp_bfgs_correspondences(I4, p.guess); t_orc = time.perf_counter() - t0
isrc1, itgt1, M1 = gc.gicp_bfgs_get_correspondences()
same = itgt1 == itgt0 if len(itgt1) == len(itgt0) else np.zeros(1, bool)
print(json.dumps({"correspondence_step": {"scan_points": len(p.scan), "map_points": len(p.submap), "pairs": int(m), "gpu_ms": 1e3 * float(np.median(ts)),
                  "gpu_first_call_s_incl_maps_and_covariances": t_first, "oracle_serial_ms": 1e3 * t_orc, "pairs_equal_to_oracle": bool(len(isrc1) == len(isrc0) and np.array_equal(isrc1, isrc0)),
                  "same_target_fraction": float(same.mean())}}, indent=1))

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
